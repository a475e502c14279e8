"""f1 (BASELINE config 5): tiled inference throughput on one MI355X -- 1024x1024 image -> 25 overlapping 256x256 tiles
(overlap 0.9, infer.py:162) -> generator forward (eval) -> overlap-averaged mask.  Prints tiles/s, images/s, peak VRAM."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import patchgan_amd as pg
from patchgan_amd.infer import n_crop, build_mask

torch.manual_seed(0)
g = pg.UNet(3, 1, 64, activation='leakyrelu', final_act='sigmoid').cuda().eval()
img = torch.rand(3, 1024, 1024).cuda()
reps = int(os.environ.get('REPS', '10'))
def once():
    crops = n_crop(img, 256, 0.9)
    with torch.no_grad():
        masks = g(crops)
    return crops.shape[0], masks
for _ in range(2):
    once()
torch.cuda.synchronize(); torch.cuda.reset_peak_memory_stats()
t0 = time.perf_counter()
for _ in range(reps):
    n, masks = once()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
t1 = time.perf_counter()
m = build_mask(masks, 256, (1024, 1024), 0.5, 0.9)
dtm = time.perf_counter() - t1
print(json.dumps({'metric': 'tiled inference, 1024x1024 -> 25 tiles of 256x256, UNet nf=64 fp32', 'tiles_per_s': round(n / dt, 1),
                  'images_per_s': round(1 / dt, 2), 'ms_per_image_forward': round(dt * 1e3, 2), 'build_mask_ms': round(dtm * 1e3, 2),
                  'peak_vram_GiB': round(torch.cuda.max_memory_allocated() / 2**30, 2)}))
