"""K4 (north star: "wgrad/dgrad fused into the backward kernel") priced by measurement on the implicit-GEMM layers the review names
(enc4, dec2: 512-channel layers on 16 x 16 / 8 x 8 maps, batch 16) and on the smallest ones (enc6, dec0).

For each layer: time of the weight-gradient call, of the data-gradient call, of the two back to back on one stream (what the step does),
of the two on two streams (concurrent), and of pg_conv4x4_bwd_big where the layer is a ConvTranspose2d -- plus the bytes each GEMM
must move (dy, the second operand, the output; algorithmic) so that the share a fused kernel could save (dy read once instead of
twice + one launch boundary) is a number.  Writes one JSON line; tools/collect... copies it to profiles/r03_k4_ab.json."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from patchgan_amd import engine as E

dev = torch.device('cuda')
B, REPS = 16, 30
LAYERS = [('enc4', 'conv', (B, 16, 16, 512, 512, 2)), ('enc6', 'conv', (B, 4, 4, 512, 512, 2)),
          ('dec0', 'convT', (B, 4, 4, 512, 512, 2)), ('dec2', 'convT', (B, 16, 16, 1024, 512, 2))]


def timed(fn, streams=None):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / REPS * 1e3


out = {}
side = torch.cuda.Stream()
for name, kind, geom in LAYERS:
    N, Hb, Wb, Ca, Cb, s = geom
    op = E.ConvOp(*geom, E.DEFAULT_ALGO)
    big = E.View.alloc(N, Hb, Wb, Cb, dev); big.t.normal_()
    small = E.View.alloc(N, op.Hs, op.Ws, Ca, dev); small.t.normal_()
    P = torch.randn(16 * Ca * Cb, device=dev) * 0.05
    dP = torch.empty_like(P)
    dbig, dsmall = E.View.alloc(N, Hb, Wb, Cb, dev), E.View.alloc(N, op.Hs, op.Ws, Ca, dev)
    if kind == 'conv':          # dy = small: dW = wgrad(dy, x), dx = small2big(dy, W)
        wg = lambda: op.wgrad(small, big, dP, 0)
        dg = lambda: op.small2big(small, P, 0, None, 0, dbig)
        dy_bytes, other = small.npix * Ca * 4, {'wgrad_second_operand_x': big.npix * Cb * 4, 'dgrad_weights': P.numel() * 4}
        outs = {'dW': P.numel() * 4, 'dx': big.npix * Cb * 4}
    else:                       # dy = big: dW = wgrad(x, dy), dx = big2small(dy, W)
        wg = lambda: op.wgrad(small, big, dP, 0)
        dg = lambda: op.big2small(big, P, 0, None, 0, dsmall)
        dy_bytes, other = big.npix * Cb * 4, {'wgrad_second_operand_x': small.npix * Ca * 4, 'dgrad_weights': P.numel() * 4}
        outs = {'dW': P.numel() * 4, 'dx': small.npix * Ca * 4}

    def both():
        wg(); dg()

    def two_streams():
        ev = torch.cuda.Event(); ev.record()
        with torch.cuda.stream(side):
            side.wait_event(ev)
            # (own workspace on the side stream is not needed for a timing probe of these two kernels: the weight gradient's
            #  split-K slabs and the data gradient's live in disjoint halves only by luck -- results are not checked here)
            dg()
        wg()
        torch.cuda.current_stream().wait_stream(side)
    r = {'geom': geom, 'kind': kind, 'wgrad_us': round(timed(wg), 1), 'dgrad_us': round(timed(dg), 1), 'back_to_back_us': round(timed(both), 1),
         'two_streams_us': round(timed(two_streams), 1), 'kernels': [op.describe(2), op.describe(1 if kind == 'conv' else 0)],
         'dy_MB': round(dy_bytes / 1e6, 2), 'other_operands_MB': {k: round(v / 1e6, 2) for k, v in other.items()},
         'outputs_MB': {k: round(v / 1e6, 2) for k, v in outs.items()}}
    if kind == 'convT':
        r['bwd_big_one_call_us'] = round(timed(lambda: op.bwd_big(small, big, P, dP, 0, dsmall)), 1)
    total = 2 * dy_bytes + sum(other.values()) + sum(outs.values())
    # what ONE fused kernel could save at best: the second read of dy at the achievable memory rate + one dependent launch boundary
    r['algorithmic_MB_two_kernels'] = round(total / 1e6, 2)
    r['dy_share_of_bytes'] = round(dy_bytes / total, 4)
    r['best_case_saving_us'] = round(dy_bytes / 6.3e12 * 1e6 + 1.5, 2)
    r['best_case_saving_share'] = round(r['best_case_saving_us'] / r['back_to_back_us'], 4)
    out[name] = r
print(json.dumps(out))
