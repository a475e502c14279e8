"""Soak run: N steps of the cfg2 workload with fresh random batches; reports throughput per block of 100 steps, peak memory
and that every loss stays finite (leaks / slowdowns / NaNs show up here, not in a 20-step bench)."""
import json, os, sys, tempfile, time, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import patchgan_amd as pg

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
precision = sys.argv[2] if len(sys.argv) > 2 else 'fp32'        # tools/soak.py 1000 bf16: bf16 kernels + bf16 activation storage
torch.manual_seed(0)
g = pg.UNet(3, 1, 64, use_dropout=os.environ.get('SOAK_DROPOUT', '1') != '0', activation='leakyrelu', final_act='sigmoid').cuda()
d = pg.Discriminator(4, 64, n_layers=3).cuda()
g.set_precision(precision)
d.set_precision(precision)
t = pg.Trainer(g, d, tempfile.mkdtemp())
t.gc_freeze = True
t.graph = os.environ.get('SOAK_GRAPH', 'auto')          # 'auto' (dropout off: SOAK_DROPOUT=0) / '' = launch by launch on one stream
if t.graph == '':
    t.graph = False
t.setup_optimizers(2e-4, 2e-4)
g.train(); d.train()
gen = torch.Generator(device='cuda').manual_seed(1)
blocks, t0, ok = [], time.perf_counter(), True
for s in range(1, steps + 1):
    x = torch.rand(16, 3, 256, 256, device='cuda', generator=gen)
    y = (x.mean(1, keepdim=True) > 0.5).float()          # a learnable target
    cur = t.batch(x, y, train=True)
    if s > 1:
        ok = ok and all(math.isfinite(v) for v in l.values())     # step s - 1, read once step s is enqueued (as Trainer.train does)
    l = cur
    if s % 100 == 0:
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        blocks.append({'step': s, 'img_per_s': round(1600 / dt, 1), 'gen': round(l['gen'], 4), 'disc': round(l['disc'], 4),
                       'mem_GiB': round(torch.cuda.memory_allocated() / 2**30, 3), 'reserved_GiB': round(torch.cuda.memory_reserved() / 2**30, 3), 'peak_GiB': round(torch.cuda.max_memory_allocated() / 2**30, 3)})
        print(blocks[-1], flush=True)
        t0 = time.perf_counter()
print(json.dumps({'finite': ok, 'blocks': len(blocks)}))
