"""Print the 10-step loss curve of the cfg2 workload (bs 16, 256x256, nf = ndf = 64) on the GPU as JSON: used to compare
kernel-selection variants (e.g. PATCHGAN_NO_WINOGRAD=1, PATCHGAN_ALGO=direct) on identical inputs and weights."""
import json, os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import patchgan_amd as pg

B = int(os.environ.get('LC_BATCH', '16'))
steps = int(os.environ.get('LC_STEPS', '10'))
torch.manual_seed(1234)
g = pg.UNet(3, 1, 64, use_dropout=False, activation='leakyrelu', final_act='sigmoid').cuda()
d = pg.Discriminator(4, 64, n_layers=3).cuda()
gen = torch.Generator().manual_seed(7)
x = torch.rand(B, 3, 256, 256, generator=gen)
y = (torch.rand(B, 1, 256, 256, generator=gen) > 0.7).float()
t = pg.Trainer(g, d, tempfile.mkdtemp())
t.setup_optimizers(1e-3, 1e-3)
g.train(); d.train()
out = [t.batch(x, y, train=True) for _ in range(steps)]
print(json.dumps([[o[k] for k in ('gen', 'gen_loss', 'gdisc', 'discr', 'discf', 'disc')] for o in out]))
