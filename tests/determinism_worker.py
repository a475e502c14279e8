"""Back-to-back repetitions of K training steps (bf16 network of the determinism hunt (tools/forensics/), N = 1), device clones of the step's end-of-step
tensors after every step (no host synchronisation inside a repetition beyond the step's own loss read-back), compared with repetition 0
at the end.  Run two of these at once on one GPU (tests/test_determinism_gpu.py does).
usage: python tests/determinism_worker.py [precision] [reps] [steps] [clone: 0|1]"""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import patchgan_amd as pg

prec = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
two_streams = prec == 'fp32-two-streams'          # fp32 with the weight gradients of each backward pass on a second stream (Trainer 'auto')
if two_streams:
    prec = 'fp32'
    pg.Trainer.AUTO_FORCE = 'eager2'              # by decree: two streams from the 4th step of a kind on
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
do_clone = (sys.argv[4] if len(sys.argv) > 4 else '1') == '1'
N = 1
torch.manual_seed(77)
g0 = pg.UNet(3, 4, 64, activation='leakyrelu', final_act='softmax', use_dropout=False)
d0 = pg.Discriminator(7, 64, n_layers=3)
gw = {k: v.clone() for k, v in g0.state_dict().items()}
dw = {k: v.clone() for k, v in d0.state_dict().items()}
gen = torch.Generator().manual_seed(8)
x = torch.rand(N, 3, 256, 256, generator=gen)
y = (torch.rand(N, 4, 256, 256, generator=gen) > 0.7).float()
runs = []
for r in range(reps):
    g = pg.UNet(3, 4, 64, activation='leakyrelu', final_act='softmax', use_dropout=False)
    d = pg.Discriminator(7, 64, n_layers=3)
    g.load_state_dict(gw); d.load_state_dict(dw)
    g.cuda().set_precision(prec); d.cuda().set_precision(prec)
    t = pg.Trainer(g, d, tempfile.mkdtemp())
    t.loss_type = 'weighted_bce'
    if two_streams:
        t.graph = 'auto'
    t.setup_optimizers(1e-3, 1e-3)
    g.train(); d.train()
    rec = []
    for s in range(steps):
        l = t.batch(x, y, train=True)
        if do_clone:
            t.flush()          # (a two-stream step leaves the discriminator's update running on the second stream: flat buffers are raw)
            rec.append((f'step {s + 1} gen image', t._last_gen.t.clone()))
            rec.append((f'step {s + 1} G grad', g.grad_flat.clone()))
            rec.append((f'step {s + 1} D grad', d.grad_flat.clone()))
            rec.append((f'step {s + 1} G weights', g.flat.clone()))
            rec.append((f'step {s + 1} D weights', d.flat.clone()))
        rec.append((f'step {s + 1} losses', torch.tensor([l[k] for k in ('gen', 'gen_loss', 'gdisc', 'discr', 'discf', 'disc')], dtype=torch.float64)))
    t.flush()
    rec.append(('final G weights', g.flat.clone()))
    rec.append(('final D weights', d.flat.clone()))
    runs.append(rec)
torch.cuda.synchronize()
for r in range(1, reps):
    msg = []
    for (k, a), (_, b) in zip(runs[0], runs[r]):
        if not torch.equal(a.contiguous().view(torch.uint8), b.contiguous().view(torch.uint8)):
            af, bf = a.double(), b.double()
            m = ~(torch.isnan(af) | torch.isnan(bf))
            msg.append(f'{k}: rel {float((af[m] - bf[m]).abs().max() / af[m].abs().max()):.1e} ({int(((af != bf) & m).sum())} of {a.numel()})')
    print(f'pid {os.getpid()} rep {r}:', 'all equal' if not msg else 'FIRST: ' + msg[0] + f'   (+{len(msg) - 1} more)', flush=True)
