"""One way of launching the training step, by decree: python tools/launch_mode_bench.py <cfgN> <f32|bf16> <eager1|eager2|graph|graph2>
(Trainer.AUTO_FORCE; graph2 = the two-stream fork / join schedule inside the captured step).  Prints ms per step over 40 steps, a hash of both
networks' weights (the modes are the same computation bit for bit) and the last generator loss."""
import os, sys, time, tempfile
sys.path.insert(0, os.getcwd())
import torch, bench
import patchgan_amd as pg
cfgname, dtype, mode = sys.argv[1], sys.argv[2], sys.argv[3]
cfg = bench.CONFIGS[cfgname]
dev = torch.device('cuda')
torch.manual_seed(1234)
G = pg.UNet(3, cfg['out_nc'], cfg['nf'], use_dropout=False, activation=cfg['activation'], final_act=cfg['final_act']).to(dev)
D = pg.Discriminator(3 + cfg['out_nc'], cfg['ndf'], n_layers=cfg['n_layers'], norm=False).to(dev)
if dtype == 'bf16':
    G.set_precision('bf16'); D.set_precision('bf16')
t = pg.Trainer(G, D, tempfile.mkdtemp())
t.loss_type, t.seg_alpha = cfg['loss_type'], 200
t.graph = 'auto'
pg.Trainer.AUTO_FORCE = mode
t.setup_optimizers(1e-3, 1e-3)
G.train(); D.train()
x, y = bench.make_inputs(cfg['batch'], 0, cfg)
x, y = x.to(dev), y.to(dev)
for _ in range(12):
    l = t.batch(x, y, train=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 40
for _ in range(n):
    l = t.batch(x, y, train=True)
t.flush(); torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / n * 1e3
import hashlib
h = hashlib.sha256(G.flat.cpu().numpy().tobytes() + D.flat.cpu().numpy().tobytes()).hexdigest()[:12]
print(cfgname, dtype, mode, t.launch_mode, f'{ms:.3f} ms/step', h, float(l['gen']))
