"""Is the two-rank (gloo, one shared GPU) bf16 run reproducible from launch to launch?  Runs the workers of
tests/test_dp_gpu.py::test_two_rank_bf16_storage_tracks_single_process REPS times and compares rank 0's losses and weights bitwise.
usage: python tools/debug_dp_repro.py [REPS]"""
import os, sys, socket
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.multiprocessing as mp

if __name__ == '__main__':
    from tests.test_dp_gpu import _worker_bf16, _collect
    import patchgan_amd as pg
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    torch.manual_seed(77)
    g = pg.UNet(3, 4, 64, activation='leakyrelu', final_act='softmax', use_dropout=False)
    d = pg.Discriminator(7, 64, n_layers=3)
    gw = {k: v.clone() for k, v in g.state_dict().items()}
    dw = {k: v.clone() for k, v in d.state_dict().items()}
    gen = torch.Generator().manual_seed(8)
    x = torch.rand(2, 3, 256, 256, generator=gen)
    y = (torch.rand(2, 4, 256, 256, generator=gen) > 0.7).float()
    runs = []
    for r in range(reps):
        s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
        ctx = mp.get_context('spawn')
        q = ctx.Queue()
        procs = [ctx.Process(target=_worker_bf16, args=(k, 2, port, q, gw, dw, x, y, 3)) for k in range(2)]
        for p in procs:
            p.start()
        res = sorted(_collect(q, procs), key=lambda t: t[0])
        for p in procs:
            p.join(timeout=120)
        runs.append(res[0])
        print(f'launch {r}: rank 0 step losses\n{res[0][1]}', flush=True)
        print('  ranks agree:', np.array_equal(res[0][2], res[1][2]) and np.array_equal(res[0][3], res[1][3]))
    for r in range(1, reps):
        same = np.array_equal(runs[0][1], runs[r][1]) and np.array_equal(runs[0][2], runs[r][2]) and np.array_equal(runs[0][3], runs[r][3])
        rel = np.abs(runs[r][1] - runs[0][1]) / np.maximum(np.abs(runs[0][1]), 1e-3)
        print(f'launch {r} vs 0: bitwise {"EQUAL" if same else "DIFFERENT"}; max rel loss difference per step {rel.max(axis=1)}')
        print('   step-1 per key (gen, gen_loss, gdisc, discr, discf, disc):', rel[0])
