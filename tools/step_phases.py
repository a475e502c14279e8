"""Where the step's wall time goes, stream by stream: HIP events at the phase boundaries of Trainer._enqueue_step (Trainer.marks), on
whichever stream the phase runs on, over a few steady-state steps.  Unlike a rocprofv3 kernel trace this does not serialise the streams.
    python tools/step_phases.py [cfg2] [f32|bf16] [steps=8]          PATCHGAN_DP_FORCE=1 for the data-parallel path (one-rank RCCL group)"""
import os
import sys
import tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import patchgan_amd as pg

cfg = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else 'cfg2']
dtype = sys.argv[2] if len(sys.argv) > 2 else 'f32'
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 8
dev = torch.device('cuda', 0)
torch.cuda.set_device(0)
if os.environ.get('PATCHGAN_DP_FORCE') == '1':
    import torch.distributed as dist
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29577')
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
torch.manual_seed(1234)
G = pg.UNet(3, cfg['out_nc'], cfg['nf'], use_dropout=False, activation=cfg['activation'], final_act=cfg['final_act']).to(dev)
D = pg.Discriminator(3 + cfg['out_nc'], cfg['ndf'], n_layers=cfg['n_layers'], norm=False).to(dev)
if dtype == 'bf16':
    G.set_precision('bf16')
    D.set_precision('bf16')
t = pg.Trainer(G, D, tempfile.mkdtemp(prefix='pgph_'))
t.loss_type, t.seg_alpha = cfg['loss_type'], 200
t.two_streams = True if os.environ.get('ONE_STREAM') != '1' else False
t.setup_optimizers(1e-3, 1e-3)
G.train()
D.train()
x, y = bench.make_inputs(cfg['batch'], 0, cfg)
x, y = x.to(dev), y.to(dev)
for _ in range(24):
    t.batch(x, y, train=True)
torch.cuda.synchronize()
t.marks = []
for _ in range(steps):
    t.batch(x, y, train=True)
t.flush()
torch.cuda.synchronize()
marks, t.marks = t.marks, None
# split into steps at 'start'
runs, cur = [], None
for name, ev, th in marks:
    if name == 'start':
        cur = []
        runs.append(cur)
    cur.append((name, ev, th))
names = [n for n, _, _ in runs[1]]
print(f'{cfg["desc"][:40]} {dtype}  DP={os.environ.get("PATCHGAN_DP_FORCE", "0")}  two_streams={t.two_streams}: ms since the step\'s start (median over {len(runs) - 2} steps)')
for i, n in enumerate(names):
    vals = sorted(r[0][1].elapsed_time(r[i][1]) for r in runs[1:-1] if len(r) == len(names))
    host = sorted((r[i][2] - r[0][2]) * 1e3 for r in runs[1:-1] if len(r) == len(names))
    print(f'  {n:32s} device {vals[len(vals) // 2]:8.3f}   host enqueue {host[len(host) // 2]:8.3f}')
starts = [r[0][1] for r in runs]
per = sorted(starts[i].elapsed_time(starts[i + 1]) for i in range(1, len(starts) - 1))
print(f'  step to step (start -> next start)   {per[len(per) // 2]:8.3f}')
