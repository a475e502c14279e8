"""Forensics for the discriminator head's data gradient (Ca == 1) under GPU sharing: as tools/debug_cc4.py (dy, t, weights and the OUTPUT of
every such call cloned on the device), but when two repetitions differ the differing elements are CLASSIFIED: which repetition is the
wrong one (against a float64 restatement from the cloned inputs), and what the wrong values equal -- stale memory (the same buffer's
content from an earlier call), a neighbour's value, the un-multiplied accumulator, acc * f'(t of a neighbour), ...
With PATCHGAN_EXPERIMENT=1 PATCHGAN_CA1S1_BF16=1 the LDS-staged kernel runs on bf16 outputs.  Two of these at once.
usage: python tools/debug_cc5.py [reps] [steps]"""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import patchgan_amd as pg
from patchgan_amd import engine as E, _lib as L

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 18
N = 1
rec = []
o_s2b = E.ConvOp.small2big


def s2b(self, small, P, p_off, bias, b_off, big, act=L.ACT_NONE, **kw):
    if self.Ca != 1:
        return o_s2b(self, small, P, p_off, bias, b_off, big, act, **kw)
    mul = kw.get('mul')
    r = dict(dy=small.t.clone(), t=mul[0].t.clone() if mul else None, w=P[p_off:p_off + 16 * self.Cb].clone(),
             before=big.t.clone(), ptr=big.t.data_ptr(), geom=(small.N, small.H, small.W, big.H, big.W, self.Cb),
             mact=mul[1] if mul else None)
    ret = o_s2b(self, small, P, p_off, bias, b_off, big, act, **kw)
    r['out'] = big.t.clone()
    rec.append(r)
    return ret


E.ConvOp.small2big = s2b


def acc64(r):
    """out[n, h, w, b] = sum over taps of dy[n, h + 1 - kh, w + 1 - kw] * P[kh * 4 + kw][b] in float64 (stride 1, pad 1)."""
    n, hs, ws, hb, wb, cb = r['geom']
    dy = r['dy'].double().view(n, hs, ws)
    w = r['w'].double().view(16, cb)
    pad = torch.zeros(n, hs + 6, ws + 6, dtype=torch.float64, device=dy.device)
    pad[:, 3:3 + hs, 3:3 + ws] = dy
    out = torch.zeros(n, hb, wb, cb, dtype=torch.float64, device=dy.device)
    for kh in range(4):
        for kw in range(4):
            out += pad[:, 4 - kh:4 - kh + hb, 4 - kw:4 - kw + wb].unsqueeze(-1) * w[kh * 4 + kw]
    return out.flatten()


def deriv(t, act):
    t = t.double().flatten()
    if act == L.ACT_TANH:
        return 1 - t * t
    if act == L.ACT_SIGMOID:
        return t * (1 - t)
    if act == L.ACT_LEAKY:
        return torch.where(t > 0, torch.ones_like(t), torch.full_like(t, 0.2))
    if act == L.ACT_RELU:
        return (t > 0).double()
    return torch.ones_like(t)


def classify(ri, run_a, run_b, i):
    a, b = run_a[i], run_b[i]
    oa, ob = a['out'].double().flatten(), b['out'].double().flatten()
    idx = torch.nonzero(oa != ob).flatten()
    acc = acc64(a)
    d = deriv(a['t'], a['mact']) if a['t'] is not None else torch.ones_like(acc)
    ref = acc * d
    ea, eb = (oa[idx] - ref[idx]).abs().sum().item(), (ob[idx] - ref[idx]).abs().sum().item()
    bad, good, who = (ob, oa, f'rep {ri}') if eb > ea else (oa, ob, 'rep 0')
    cb = a['geom'][5]
    print(f'  call {i} geom {a["geom"]}, {idx.numel()} elements differ; wrong run: {who} (|err| vs float64 {max(ea, eb):.3e} against {min(ea, eb):.3e})')
    print(f'  index % 4 histogram {torch.bincount(idx % 4, minlength=4).tolist()}; channels {int((idx % cb).min())}..{int((idx % cb).max())}; pixels {sorted(set((idx // cb).tolist()))[:24]}')
    for pix in sorted(set((idx // cb).tolist()))[:16]:
        sel = idx[(idx // cb) == pix] % cb
        q0, q2 = sorted((sel[sel % 4 == 0] // 4).tolist()), sorted((sel[sel % 4 == 2] // 4).tolist())
        print(f'    pixel {pix}: quads with a wrong e0: {q0}')
        print(f'    pixel {pix}: quads with a wrong e2: {q2}')
    # which taps explain the error: implied accumulator = bad / f(t); per element the 16 products x_tap * w_tap[b]
    n, hs, ws, hb, wb, _ = a['geom']
    dyp = torch.zeros(n, hs + 6, ws + 6, dtype=torch.float64, device=acc.device)
    dyp[:, 3:3 + hs, 3:3 + ws] = a['dy'].double().view(n, hs, ws)
    w64 = a['w'].double().view(16, cb)
    for j in idx[:24].tolist():
        pix, ch = j // cb, j % cb
        nn, rem = pix // (hb * wb), pix % (hb * wb)
        h, wq = rem // wb, rem % wb
        xs_ = [float(dyp[nn, h + 4 - kh, wq + 4 - kw]) for kh in range(4) for kw in range(4)]
        terms = [xs_[t_] * float(w64[t_, ch]) for t_ in range(16)]
        if abs(float(d[j])) < 1e-3:
            continue
        imp = float(bad[j] / d[j])
        delta = imp - float(acc[j])
        # the other x of each ds_read2 pair, and neighbouring channels' weights
        best = []
        for t_ in range(16):
            for t2 in range(16):
                if t2 != t_:
                    alt = xs_[t2] * float(w64[t_, ch]) - terms[t_]
                    best.append((abs(alt - delta), f'tap {t_} multiplied by the x of tap {t2}'))
            for dc in (-2, -1, 1, 2):
                if 0 <= ch + dc < cb:
                    alt = xs_[t_] * float(w64[t_, ch + dc]) - terms[t_]
                    best.append((abs(alt - delta), f'tap {t_} with the weight of channel {dc:+d}'))
            best.append((abs(-terms[t_] - delta), f'tap {t_} missing'))
            best.append((abs(terms[t_] - delta), f'tap {t_} twice'))
        best.sort()
        print(f'    idx {j} (pix {pix} = n {nn} h {h} w {wq}, ch {ch}): acc {float(acc[j]):.6e} implied {imp:.6e} delta {delta:.3e}; nearest single-term explanations: '
              + '; '.join(f'{nm} (residual {r_:.1e})' for r_, nm in best[:3]))
    tol = 2.0 ** -7
    bv = bad[idx]

    def frac(cand):
        c = cand[idx] if cand.numel() == bad.numel() else None
        if c is None:
            return None
        return float(((bv - c).abs() <= tol * c.abs().clamp_min(1e-30)).double().mean())

    cands = {'float64 acc * f(t) [i.e. the right value]': ref, 'acc alone (multiplier not applied)': acc, 'zero': torch.zeros_like(ref),
             'buffer content before the call (store did not land)': (b if who != 'rep 0' else a)['before'].double().flatten()}
    src = run_b if who != 'rep 0' else run_a
    for k in range(1, 7):
        if i - k >= 0 and src[i - k]['out'].numel() == bad.numel():
            cands[f'output of call {i - k} (same run)'] = src[i - k]['out'].double().flatten()
            if src[i - k]['t'] is not None and src[i - k]['t'].numel() == bad.numel():
                cands[f'acc * f(t of call {i - k})'] = acc * deriv(src[i - k]['t'], a['mact'])
    for s in (1, -1, 2, -2, 3, -3, 4, -4, cb, -cb, 64 * 4, -64 * 4):
        cands[f'right value of element index {s:+d}'] = torch.roll(ref, -s)
        cands[f'acc * f(t[index {s:+d}])'] = acc * torch.roll(d, -s)
        cands[f'acc[index {s:+d}] * f(t)'] = torch.roll(acc, -s) * d
    for name, c in cands.items():
        f = frac(c)
        if f is not None and f > 0.2:
            print(f'    {f * 100:5.1f} % of the wrong elements equal (to 2^-7): {name}')
    tt = a['t'].double().flatten() if a['t'] is not None else torch.zeros_like(acc)
    for j in idx[:4].tolist():
        implied = bad[j] / acc[j] if acc[j] != 0 else float('nan')
        print(f'    idx {j} (pixel {j // cb}, ch {j % cb}): good {good[j]:.6e} bad {bad[j]:.6e} ref {ref[j]:.6e} | acc {acc[j]:.6e} t {tt[j]:.6f} f {d[j]:.6e} implied f {float(implied):.6e}'
              f' -> implied |t| {float((1 - implied).clamp_min(0).sqrt()) if a["mact"] == L.ACT_TANH else float("nan"):.6f}; t neighbours {[round(float(tt[j + o]), 4) for o in (-2, -1, 1, 2) if 0 <= j + o < tt.numel()]}')


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
    g.cuda().set_precision('bf16'); d.cuda().set_precision('bf16')
    t = pg.Trainer(g, d, tempfile.mkdtemp())
    t.loss_type = 'weighted_bce'
    t.setup_optimizers(1e-3, 1e-3)
    g.train(); d.train()
    rec = []
    for s in range(steps):
        l = t.batch(x, y, train=True)
        l['gen']
    runs.append(rec)
    print(f'pid {os.getpid()} rep {r} done ({len(rec)} calls recorded)', flush=True)
torch.cuda.synchronize()
KEYS = ('dy', 't', 'w', 'out')


def same(a, b):
    if a is None or b is None:
        return a is b
    return a.shape == b.shape and torch.equal(a.contiguous().view(torch.uint8), b.contiguous().view(torch.uint8))


for r in range(1, reps):
    first = None
    for i, (a, b) in enumerate(zip(runs[0], runs[r])):
        bad = [k for k in KEYS if not same(a[k], b[k])]
        if bad:
            first = (i, bad)
            break
    if first is None:
        print(f'pid {os.getpid()} rep {r}: all equal', flush=True)
        continue
    i, bad = first
    print(f'pid {os.getpid()} rep {r}: FIRST difference at call {i}: {bad} (pointers of the output buffer: rep 0 {runs[0][i]["ptr"]:#x}, rep {r} {runs[r][i]["ptr"]:#x})', flush=True)
    if bad == ['out']:
        classify(r, runs[0], runs[r], i)
    sys.stdout.flush()
