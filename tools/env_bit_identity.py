"""Are two experiment settings the same computation bit for bit?  Three cfg2-shaped G+D steps (nf = ndf = 64, batch 4, 256 x 256) in a child
process per setting; prints a hash of both networks' weights and the losses.   python tools/env_bit_identity.py "VAR=a" "VAR=b" ...
(each argument: space-separated VAR=value pairs, or "-" for none; PATCHGAN_EXPERIMENT=1 is added)"""
import hashlib
import os
import subprocess
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if sys.argv[1:2] == ['--child']:
    import tempfile
    import torch
    import patchgan_amd as pg
    torch.manual_seed(11)
    g = pg.UNet(3, 1, 64, use_dropout=False, activation='leakyrelu', final_act='sigmoid').cuda()
    d = pg.Discriminator(4, 64, n_layers=3).cuda()
    t = pg.Trainer(g, d, tempfile.mkdtemp())
    t.two_streams = False
    t.setup_optimizers(1e-3, 1e-3)
    g.train(), d.train()
    gen = torch.Generator().manual_seed(5)
    x = torch.rand(4, 3, 256, 256, generator=gen)
    y = (torch.rand(4, 1, 256, 256, generator=gen) > 0.7).float()
    for _ in range(3):
        l = t.batch(x, y, train=True)
    t.flush()
    torch.cuda.synchronize()
    h = hashlib.sha256(g.flat.cpu().numpy().tobytes() + d.flat.cpu().numpy().tobytes()).hexdigest()[:16]
    print('RESULT', h, {k: float(v) for k, v in l.items()})
else:
    for setting in sys.argv[1:]:
        env = dict(os.environ, PATCHGAN_EXPERIMENT='1')
        if setting != '-':
            env.update(dict(kv.split('=', 1) for kv in setting.split()))
        out = subprocess.run([sys.executable, os.path.abspath(__file__), '--child'], env=env, capture_output=True, text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith('RESULT')]
        print(f'[{setting}]', line[0] if line else out.stderr[-400:])
