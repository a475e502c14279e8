import os, sys, tempfile, cProfile, pstats, io
sys.path.insert(0, os.getcwd())
import torch, bench
import patchgan_amd as pg
cfg = bench.CONFIGS[sys.argv[1]]
dev = torch.device('cuda')
torch.manual_seed(1234)
G = pg.UNet(3, cfg['out_nc'], cfg['nf'], use_dropout=False, activation=cfg['activation'], final_act=cfg['final_act']).to(dev)
D = pg.Discriminator(3 + cfg['out_nc'], cfg['ndf'], n_layers=cfg['n_layers'], norm=False).to(dev)
t = pg.Trainer(G, D, tempfile.mkdtemp())
t.loss_type, t.seg_alpha = cfg['loss_type'], 200
t.two_streams = True
t.setup_optimizers(1e-3, 1e-3)
G.train(); D.train()
x, y = bench.make_inputs(cfg['batch'], 0, cfg)
x, y = x.to(dev), y.to(dev)
for _ in range(20):
    t.batch(x, y, train=True)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(50):
    t.batch(x, y, train=True)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print(s.getvalue()[:6000])
