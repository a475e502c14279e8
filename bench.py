#!/usr/bin/env python3
"""bench.py -- train images/sec of one full patchGAN G+D step (Trainer.batch(train=True) semantics) on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1], "cfg2" of SURVEY.md 8(d)): synthetic 256x256x3 -> 1-channel masks, batch 16
PER GPU (weak scaling; cfg3 = 8 x 16), UNet nf=64 leakyrelu/sigmoid, Discriminator ndf=64 n_layers=3 norm=False,
focal-Tversky x 200 + BCE, Adam lr 1e-3, fp32, random-init weights, inputs resident in HBM before timing.

Prints ONE JSON line on rank 0.  `roofline` is measured live with HIP events around every launch of the conv
kernels in the timed region; `cpu_baseline` times the CPU oracle (oracle/, the restatement pinned to the
reference's goldens) on this box's host cores for a bounded sample of the same workload (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3            # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
GFLOP_PER_IMAGE = 85.98                  # useful conv FLOPs of one G+D step per image (cfg2)
BATCH_PER_GPU = 16
SIZE = 256


def make_inputs(batch, rank):
    import torch
    g = torch.Generator().manual_seed(7 + rank)
    x = torch.rand(batch, 3, SIZE, SIZE, generator=g)
    y = (torch.rand(batch, 1, SIZE, SIZE, generator=g) > 0.7).float()
    return x, y


def pmc_traffic(symbol):
    """HBM bytes per launch of `symbol` from the committed rocprofv3 --pmc passes (profiles/*_pmc_traffic.json: FETCH_SIZE
    and WRITE_SIZE collected in separate passes, gfx950 corrections applied); None if that kernel was not profiled."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_pmc_traffic.json'))):
        try:
            k = json.load(open(f))['kernels'].get(symbol.replace(' ', ''))
        except Exception:
            k = None
        if k:
            best = k['hbm_bytes_per_launch']
    return best


def usable_cpus():
    """Host threads this process may really use: min(affinity mask, cgroup CPU quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def cpu_baseline(max_seconds=25.0):
    """The CPU oracle's G+D step on the same cfg2 workload, all host cores available to this process."""
    import torch
    from oracle import patchgan_oracle as O
    cores = usable_cpus()
    torch.set_num_threads(cores)
    torch.manual_seed(1234)
    gw = O.default_init(O.unet_weight_shapes(3, 1, 64))
    dw = O.default_init(O.disc_weight_shapes(4, 64, 3, False))
    ot = O.OracleTrainer(gw, dw, activation='leakyrelu', final_act='sigmoid', n_layers=3, norm=False,
                         loss_type='tversky', seg_alpha=200)
    x, y = make_inputs(BATCH_PER_GPU, 0)
    ot.batch(x, y, train=True)                       # warm-up (oneDNN primitive creation)
    t0 = time.perf_counter()
    n = 0
    while True:
        ot.batch(x, y, train=True)
        n += 1
        if time.perf_counter() - t0 > max_seconds * 0.6 or n >= 5:
            break
    dt = time.perf_counter() - t0
    return {'value': round(BATCH_PER_GPU * n / dt, 3), 'unit': 'images/sec', 'cores': cores, 'kind': 'port',
            'sample': f'{n} G+D steps of cfg2 (bs 16, 256x256, nf=ndf=64) after 1 warm-up step; {dt / n:.2f} s/step'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--dropout', action='store_true', help='nn.Dropout(0.2) in the generator (CLI default of the reference)')
    ap.add_argument('--events', choices=['dominant', 'all', 'none'], default='dominant',
                    help='which conv launches get HIP events in the timed region (roofline leg)')
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import patchgan_amd as pg
    from patchgan_amd import engine as E

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs {args.gpus} ranks (launch with torch.distributed.run); WORLD_SIZE={world}")
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if world > 1:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        dist.init_process_group('nccl', device_id=dev)

    torch.manual_seed(1234)
    G = pg.UNet(3, 1, 64, use_dropout=args.dropout, activation='leakyrelu', final_act='sigmoid').to(dev)
    D = pg.Discriminator(4, 64, n_layers=3, norm=False).to(dev)
    t = pg.Trainer(G, D, tempfile.mkdtemp(prefix='pgbench_'))
    t.loss_type, t.seg_alpha = 'tversky', 200
    t.setup_optimizers(1e-3, 1e-3)
    G.train()
    D.train()
    x, y = make_inputs(BATCH_PER_GPU, rank)
    x, y = x.to(dev), y.to(dev)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # warm-up; the last warm-up step is profiled launch by launch to find the dominant conv kernel symbol, so that the
    # timed region only carries events around THAT kernel's launches (2 event records per launch are not free)
    wprof = E.LaunchProfiler()
    for i in range(max(args.warmup, 1)):
        E.PROFILER = wprof if i == max(args.warmup, 1) - 1 else None
        t.batch(x, y, train=True)
    E.PROFILER = None
    torch.cuda.synchronize()
    wsum = wprof.summary()
    dominant = max(wsum.items(), key=lambda kv: kv[1]['ms'])[0]
    prof = E.LaunchProfiler(only=None if args.events == 'all' else dominant)
    E.PROFILER = prof if args.events != 'none' else None
    sync()
    t0 = time.perf_counter()
    last = None
    for _ in range(args.steps):
        last = t.batch(x, y, train=True)
    sync()
    elapsed = time.perf_counter() - t0
    E.PROFILER = None
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = tt.item()

    if rank == 0:
        images = BATCH_PER_GPU * world * args.steps
        value = images / elapsed
        summ = prof.summary() if args.events != 'none' else wsum
        # dominant kernel = the conv kernel symbol with the most device time (found in the profiled warm-up step)
        sym, d = max(summ.items(), key=lambda kv: kv[1]['ms'])
        if args.events == 'none':
            d = dict(d, launches=d['launches'] * args.steps, ms=d['ms'] * args.steps, flops=d['flops'] * args.steps)
        per_step_all = wsum                      # every conv kernel, from the profiled warm-up step
        conv_ms = sum(v['ms'] for v in per_step_all.values()) * args.steps
        achieved = d['flops'] / (d['ms'] * 1e-3) / 1e12
        roofline = {'bound': 'mfma', 'kernel': sym,
                    'achieved': round(achieved, 2), 'peak': FP32_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                    'frac': round(achieved / FP32_MFMA_PEAK_TFLOPS, 4), 'traffic': pmc_traffic(sym),
                    'launches_per_step': d['launches'] / args.steps,
                    'avg_launch_ms': round(d['ms'] / d['launches'], 4),
                    'kernel_share_of_step': round(d['ms'] / args.steps / (elapsed / args.steps * 1e3), 4),
                    'all_conv_kernels_TFLOPs': round(sum(v['flops'] for v in per_step_all.values()) * args.steps / (conv_ms * 1e-3) / 1e12, 2),
                    'all_conv_share_of_step': round(conv_ms / args.steps / (elapsed / args.steps * 1e3), 4),
                    'step_frac_of_fp32_roofline': round(value / world * GFLOP_PER_IMAGE / 1e3 / FP32_MFMA_PEAK_TFLOPS, 4)}
        out = {
            'metric': 'train images/sec (G+D step) at 256x256 bs=16 per GPU', 'value': round(value, 2),
            'unit': 'images/sec', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(elapsed / args.steps * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'cfg2: 256x256x3->1 masks, bs 16/GPU, UNet nf=64 leakyrelu/sigmoid + PatchGAN ndf=64 '
                                   'n_layers=3, focal-Tversky*200 + BCE, Adam 1e-3, dropout ' + ('on' if args.dropout else 'off'),
                       'global_batch': BATCH_PER_GPU * world, 'parallelism': f'dp{world}'},
            'last_losses': {k: round(v, 5) for k, v in last.items()},
            'roofline': roofline,
        }
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline()
        elif world == 1:
            out['cpu_baseline'] = None
        out['conv_kernels_note'] = 'per-kernel table from one fully instrumented warm-up step; roofline from the timed region'
        kernels = {k: {'launches_per_step': v['launches'], 'ms_per_step': round(v['ms'], 4),
                                            'TFLOPs': round(v['flops'] / (v['ms'] * 1e-3) / 1e12, 2)}
                   for k, v in sorted(per_step_all.items(), key=lambda kv: -kv[1]['ms'])}
        out['conv_kernels'] = kernels
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
