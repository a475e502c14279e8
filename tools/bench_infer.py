"""f1 (BASELINE config 5): tiled inference throughput on one MI355X -- 1024x1024 image -> 25 overlapping 256x256 tiles
(overlap 0.9, infer.py:162) -> generator forward (eval) -> overlap-averaged mask.  Prints tiles/s, images/s, peak VRAM."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import patchgan_amd as pg
from patchgan_amd.infer import predict_image
from patchgan_amd import engine as E

torch.manual_seed(0)
precision = sys.argv[1] if len(sys.argv) > 1 else 'fp32'            # tools/bench_infer.py bf16: bf16 kernels + bf16 activation storage
g = pg.UNet(3, 1, 64, activation='leakyrelu', final_act='sigmoid').cuda().eval().set_precision(precision)
img = torch.rand(3, 1024, 1024).cuda()
reps = int(os.environ.get('REPS', '10'))
n = E.tiles_gather(img, 256, 0.9).N
for _ in range(2):
    predict_image(g, img, 256, 0.9, 0.5)
torch.cuda.synchronize(); torch.cuda.reset_peak_memory_stats()
t0 = time.perf_counter()
for _ in range(reps):
    m = predict_image(g, img, 256, 0.9, 0.5)          # includes the D2H copy of the 8 MB float64 mask
dt = (time.perf_counter() - t0) / reps
# device-side stages alone
ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
ev[0].record()
tiles = E.tiles_gather(img, 256, 0.9)
ev[1].record()
pred = E.View.alloc(tiles.N, 256, 256, 1, 'cuda')
g.engine.forward(g.flat, tiles, pred, False, 0)
ev[2].record()
mask = E.tiles_blend(pred, (1024, 1024), 0.5, 0.9)
ev[3].record()
torch.cuda.synchronize()
print(json.dumps({'metric': f'tiled inference, 1024x1024 -> 25 tiles of 256x256, UNet nf=64 {precision}', 'tiles_per_s': round(n / dt, 1),
                  'images_per_s': round(1 / dt, 2), 'ms_per_image': round(dt * 1e3, 2),
                  'gather_ms': round(ev[0].elapsed_time(ev[1]), 3), 'forward_ms': round(ev[1].elapsed_time(ev[2]), 3),
                  'blend_ms': round(ev[2].elapsed_time(ev[3]), 3),
                  'peak_vram_GiB': round(torch.cuda.max_memory_allocated() / 2**30, 2)}))
