#!/usr/bin/env python3
"""bench.py -- train images/sec of one full patchGAN G+D step (Trainer.batch(train=True) semantics) on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1: starts its own N ranks, one per GPU, over RCCL)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W              (the driver's launch: RANK / LOCAL_RANK / WORLD_SIZE from the env)

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
# workloads.  cfg2 is the one BASELINE.json's metric is quoted on and the default; the others are the
# parity-test / "next" configurations, runnable here for reference only.
CONFIGS = {
    'cfg2': dict(size=256, batch=16, out_nc=1, nf=64, ndf=64, n_layers=3, activation='leakyrelu', final_act='sigmoid',
                 loss_type='tversky', gflop_per_image=85.98,
                 desc='cfg2: 256x256x3->1 masks, bs 16/GPU, UNet nf=64 leakyrelu/sigmoid + PatchGAN ndf=64 n_layers=3, '
                      'focal-Tversky*200 + BCE, Adam 1e-3'),
    'cfg4': dict(size=512, batch=8, out_nc=4, nf=64, ndf=64, n_layers=3, activation='leakyrelu', final_act='softmax',
                 loss_type='weighted_bce', gflop_per_image=352.99,
                 desc='cfg4: 512x512x3->4-class masks, bs 8/GPU, nf=ndf=64'),
    'cfg1': dict(size=256, batch=4, out_nc=7, nf=32, ndf=16, n_layers=5, activation='relu', final_act='sigmoid',
                 loss_type='weighted_bce', gflop_per_image=12.2,
                 desc='cfg1: COCO-yaml hyper-parameters (nf=32, ndf=16, n_layers=5, relu, 7 classes, weighted BCE), 256x256 bs 4'),
}
CFG = CONFIGS['cfg2']
GFLOP_PER_IMAGE = CFG['gflop_per_image']
BATCH_PER_GPU = CFG['batch']
SIZE = CFG['size']


def make_inputs(batch, rank, cfg=None):
    import torch
    cfg = cfg or CFG
    g = torch.Generator().manual_seed(7 + rank)
    x = torch.rand(batch, 3, cfg['size'], cfg['size'], generator=g)
    y = (torch.rand(batch, cfg['out_nc'], cfg['size'], cfg['size'], generator=g) > 0.7).float()
    return x, y


def pmc_traffic(symbol):
    """(HBM bytes per launch of `symbol`, the file it came from) from the committed rocprofv3 --pmc passes
    (profiles/*_pmc_traffic.json: FETCH_SIZE and WRITE_SIZE collected in separate passes, gfx950 corrections applied; the
    newest file that has the kernel); (None, None) if that kernel was not profiled.  NOT measured in this run: PMC counters
    need the profiler."""
    import glob
    best, src = None, None
    for f in sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_pmc_traffic.json'))):
        try:
            ks = json.load(open(f))['kernels']
        except Exception:
            continue
        sym = symbol.replace(' ', '')
        # the C ABI reports the kernel family of a call (k_conv_bf16x<2,2,2,2,0,64>); the profiler sees its instantiations, which
        # carry further template flags (...,64,false,false>: without / with the epilogue multiplier, the statistics): take the
        # instantiation with the most launches
        cand = [v for n, v in ks.items() if n == sym or n.startswith(sym[:-1] + ',')]
        if cand:
            best = max(cand, key=lambda v: v.get('launches', 0))['hbm_bytes_per_launch']
            src = 'profiles/' + os.path.basename(f)
    return best, src


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


def cpu_baseline(max_seconds=25.0, cfg=None, max_steps=5):
    """The CPU oracle's G+D step on the same workload (cfg2 unless `cfg`), all host cores available to this process."""
    import torch
    from oracle import patchgan_oracle as O
    cfg = cfg or CFG
    cores = usable_cpus()
    torch.set_num_threads(cores)
    torch.manual_seed(1234)
    gw = O.default_init(O.unet_weight_shapes(3, cfg['out_nc'], cfg['nf']))
    dw = O.default_init(O.disc_weight_shapes(3 + cfg['out_nc'], cfg['ndf'], cfg['n_layers'], False))
    ot = O.OracleTrainer(gw, dw, activation=cfg['activation'], final_act=cfg['final_act'], n_layers=cfg['n_layers'],
                         norm=False, loss_type=cfg['loss_type'], seg_alpha=200)
    x, y = make_inputs(cfg['batch'], 0, cfg)
    ot.batch(x, y, train=True)                       # warm-up (oneDNN primitive creation)
    t0 = time.perf_counter()
    n = 0
    while True:
        ot.batch(x, y, train=True)
        n += 1
        if time.perf_counter() - t0 > max_seconds * 0.6 or n >= max_steps:
            break
    dt = time.perf_counter() - t0
    return {'value': round(cfg['batch'] * n / dt, 3), 'unit': 'images/sec', 'cores': cores, 'kind': 'port',
            'sample': f'{n} G+D steps of the same workload (bs {cfg["batch"]}, {cfg["size"]}x{cfg["size"]}) after 1 warm-up step; {dt / n:.2f} s/step'}


def cpu_baseline_tiles(tiles=25, reps=3):
    """The CPU oracle's generator forward over the 25 tiles of one 1024x1024 image (cfg5's dominant work; n_crop / build_mask are
    not in it), all host cores."""
    import torch
    from oracle import patchgan_oracle as O
    cores = usable_cpus()
    torch.set_num_threads(cores)
    torch.manual_seed(1234)
    gw = O.default_init(O.unet_weight_shapes(3, 1, 64))
    x = torch.rand(tiles, 3, 256, 256, generator=torch.Generator().manual_seed(5))
    with torch.no_grad():
        O.unet_forward(gw, x[:5], 'leakyrelu', 'sigmoid')            # warm-up
        t0 = time.perf_counter()
        for _ in range(reps):
            O.unet_forward(gw, x, 'leakyrelu', 'sigmoid')
        dt = (time.perf_counter() - t0) / reps
    return {'value': round(tiles / dt, 2), 'unit': 'tiles/sec', 'cores': cores, 'kind': 'port',
            'sample': f'{reps} generator forwards over {tiles} tiles of 256x256 (one 1024x1024 image) after a 5-tile warm-up; {dt:.2f} s per image'}


def _write_coco_like(folder, n, size, seed=0):
    """n synthetic COCO-stuff-like files: JPEG photographs (smooth content, quality 90) + PNG label maps."""
    import numpy as np
    from PIL import Image
    rng = np.random.default_rng(seed)
    os.makedirs(os.path.join(folder, 'img'))
    os.makedirs(os.path.join(folder, 'mask'))
    base = rng.integers(0, 256, (n, size // 8, size // 8, 3), dtype=np.uint8)
    lab = rng.integers(0, 3, (n, size // 16, size // 16), dtype=np.uint8)
    for i in range(n):
        Image.fromarray(base[i]).resize((size, size), Image.BICUBIC).save(os.path.join(folder, 'img', f'{i:012d}.jpg'), quality=90)
        Image.fromarray(lab[i]).resize((size, size), Image.NEAREST).save(os.path.join(folder, 'mask', f'{i:012d}.png'))


class _Stamped:
    """A DataLoader handed to Trainer.train that notes when each batch left it (and restarts like the loader itself)."""

    def __init__(self, loader):
        self.loader, self.stamps = loader, []
        self.sampler = getattr(loader, 'sampler', None)

    def __len__(self):
        return len(self.loader)

    def __iter__(self):
        self.stamps.append([])
        for item in self.loader:
            self.stamps[-1].append(time.perf_counter())
            yield item


def e2e_training(dev, batches=48, files=192):
    """End-to-end rate of the reference's training loop (train.py:77-83 -> trainer.py:215-218) at cfg2: JPEG / PNG files ->
    COCOStuffDataset -> DataLoader(shuffle, pin_memory, workers, persistent) -> host-to-device copy -> Trainer.train's step, for the
    reference's float items and for decoded bytes (device_pipeline=True: `/ 255.` and the one-hot mask on the GPU).  Two epochs of
    `batches` steps each through Trainer.train; the rate is the second epoch's, from its 6th batch to the drained GPU."""
    import shutil
    import torch
    import patchgan_amd as pg
    from torch.utils.data import ConcatDataset, DataLoader
    from patchgan_amd.io import COCOStuffDataset
    cfg = CONFIGS['cfg2']
    cores = usable_cpus()
    workers = max(2, min(12, cores - 3))
    torch.set_num_threads(max(1, cores - workers))
    folder = tempfile.mkdtemp(prefix='pg_e2e_')
    out = {'workload': f'cfg2 fed by DataLoader: {files} synthetic 256x256 JPEG + PNG files (repeated to {batches} batches of 16 per epoch), '
                       f'{workers} workers, pin_memory, shuffle; Trainer.train, 2 epochs, the second one timed from its 6th batch',
           'unit': 'images/sec', 'host_cpus': cores, 'workers': workers}
    import contextlib
    try:
      with contextlib.redirect_stdout(sys.stderr):        # the dataset and Trainer.train print their progress: stdout carries the ONE JSON line only
          _write_coco_like(folder, files, cfg['size'])
          for fmt in ('float', 'u8_device_pipeline'):
              torch.manual_seed(1234)
              G = pg.UNet(3, 1, cfg['nf'], use_dropout=False, activation=cfg['activation'], final_act=cfg['final_act']).to(dev)
              D = pg.Discriminator(4, cfg['ndf'], n_layers=cfg['n_layers']).to(dev)
              tr = pg.Trainer(G, D, os.path.join(folder, 'ckpt_' + fmt))
              tr.loss_type, tr.seg_alpha = cfg['loss_type'], 200
              tr.graph, tr.gc_freeze = 'auto', True          # as the patchgan_train entry point sets them
              if fmt != 'float':
                  tr.label_values = [1]
              ds = COCOStuffDataset(os.path.join(folder, 'img'), os.path.join(folder, 'mask'), labels=[1], size=cfg['size'],
                                    augmentation='resize', device_pipeline=(fmt != 'float'))
              ds = ConcatDataset([ds] * ((batches * cfg['batch'] + files - 1) // files))
              dl = _Stamped(DataLoader(ds, batch_size=cfg['batch'], shuffle=True, pin_memory=True, drop_last=True, num_workers=workers,
                                       persistent_workers=True, prefetch_factor=4))
              tr.train(dl, [], 2, dsc_learning_rate=1e-3, gen_learning_rate=1e-3, save_freq=1000)
              torch.cuda.synchronize()
              t_end = time.perf_counter()
              st = dl.stamps[1]              # second epoch of the training loader (stamps[0] = first)
              n = len(st) - 5
              out[fmt] = {'value': round(n * cfg['batch'] / (t_end - st[5]), 1), 'steps_timed': n,
                          'h2d_bytes_per_image': (3 * 4 + 1 * 4) * cfg['size'] ** 2 if fmt == 'float' else (3 + 1) * cfg['size'] ** 2}
              del dl, tr, G, D
    finally:
        shutil.rmtree(folder, ignore_errors=True)
    return out


def spawn_ranks(n, argv, script=None, timeout=None, grace=5.0):
    """`python bench.py --gpus N` without a launcher: start N fresh worker processes (one rank per GPU) from THIS process,
    which has not touched the GPU and does not import torch (a process that has initialised HIP must never exec or be
    replaced), and supervise them: every child is polled; the FIRST rank that exits non-zero (or the overall `timeout`,
    default PATCHGAN_SPAWN_TIMEOUT_S / 900 s) ends the job at once -- its siblings, which would otherwise sit in a collective
    or in the rendezvous until the RCCL watchdog fires while holding their GPUs, are terminated (SIGTERM, SIGKILL after
    `grace` seconds: exactly the PIDs started here), the failed rank is named and the exit status is non-zero.  Rank 0's
    stdout is drained by a reader thread (so polling cannot deadlock on a full pipe) and forwarded.  `script` (tests): the
    program the ranks run instead of this file."""
    import socket
    import subprocess
    import threading
    if timeout is None:
        timeout = float(os.environ.get('PATCHGAN_SPAWN_TIMEOUT_S', '900'))
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        env.setdefault('GPU_MAX_HW_QUEUES', '8')          # (patchgan_amd/__init__.py: one hardware queue per stream of the step)
        env.setdefault('OMP_NUM_THREADS', str(max(1, usable_cpus() // n)))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(script or __file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    deadline = time.monotonic() + timeout
    failure = None
    while True:
        codes = [p.poll() for p in procs]
        bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            failure = f"rank {bad[0][0]} exited with status {bad[0][1]} (all failed so far, (rank, status): {bad})"
            break
        if all(c == 0 for c in codes):
            break
        if time.monotonic() > deadline:
            failure = f"still running after {timeout:.0f} s (ranks alive: {[r for r, c in enumerate(codes) if c is None]})"
            break
        time.sleep(0.05)
    if failure is not None:
        alive = [p for p in procs if p.poll() is None]
        for p in alive:
            p.terminate()
        t_end = time.monotonic() + grace
        for p in alive:
            try:
                p.wait(timeout=max(0.0, t_end - time.monotonic()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
    reader.join(timeout=10.0)
    sys.stdout.write(b''.join(c for c in chunks if c).decode(errors='replace'))
    sys.stdout.flush()
    if failure is not None:
        raise SystemExit(f"bench.py: {failure}; the other ranks were terminated")


def main():
    # stdout carries the ONE JSON line and nothing else: whatever libraries print there (RCCL's version banner at communicator
    # creation, the dataset and Trainer.train in the loader-fed leg) goes to stderr -- file descriptor 1 is pointed at stderr for the
    # whole run and the line is written to the saved descriptor at the end
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extra', action='store_true',
                    help='skip the extra_configs leg (cfg4 bf16 training, cfg5 tiled inference) measured after the headline')
    ap.add_argument('--dropout', action='store_true', help='nn.Dropout(0.2) in the generator (CLI default of the reference)')
    ap.add_argument('--no-graph', action='store_true',
                    help='enqueue every step launch by launch (default on one GPU with dropout off: the steady-state step is replayed from a captured hipGraph)')
    ap.add_argument('--config', choices=sorted(CONFIGS), default='cfg2', help='workload (default: the BASELINE metric config)')
    ap.add_argument('--dtype', choices=['f32', 'bf16'], default='f32',
                    help="f32 = the parity path the metric is quoted on; bf16 = bf16-multiply / fp32-accumulate conv kernels ('next' row f2)")
    ap.add_argument('--fp32-activations', action='store_true',
                    help='with --dtype bf16: keep the activations fp32 in HBM (round-1 behaviour) instead of bf16 activation storage')
    ap.add_argument('--events', choices=['dominant', 'all', 'none'], default='dominant',
                    help='which conv launches get HIP events in the timed region (roofline leg)')
    args = ap.parse_args()
    global CFG, GFLOP_PER_IMAGE, BATCH_PER_GPU, SIZE
    CFG = CONFIGS[args.config]
    GFLOP_PER_IMAGE, BATCH_PER_GPU, SIZE = CFG['gflop_per_image'], CFG['batch'], CFG['size']

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        return spawn_ranks(args.gpus, sys.argv[1:])        # (forwards rank 0's stdout: that rank's one line)
    sys.stdout.flush()
    line_fd = os.dup(1)
    os.dup2(2, 1)
    os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')      # before the HIP runtime initialises (patchgan_amd/__init__.py says why)

    import torch
    import torch.distributed as dist
    import patchgan_amd as pg
    from patchgan_amd import engine as E
    from patchgan_amd import parallel

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    # PATCHGAN_DIST_BACKEND=gloo + PATCHGAN_SHARE_GPU=1 rehearse the multi-rank path on a single-GPU box (all ranks on
    # device 0, collectives through gloo); the real run is one rank per GPU over RCCL ("nccl")
    backend = os.environ.get('PATCHGAN_DIST_BACKEND', 'nccl')
    if os.environ.get('PATCHGAN_SHARE_GPU') == '1':
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    # PATCHGAN_DP_FORCE=1: run the data-parallel code path (RCCL collectives on the comm stream) even with one rank
    use_dist = world > 1 or os.environ.get('PATCHGAN_DP_FORCE') == '1'
    if use_dist:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        # bounded rendezvous / collective timeout: a rank whose peer never arrives (or died) gives up instead of waiting out
        # the 10-30 min default while holding its GPU (spawn_ranks ends the job sooner when it is the launcher)
        import datetime
        tmo = datetime.timedelta(seconds=float(os.environ.get('PATCHGAN_DIST_TIMEOUT_S', '300')))
        if backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev, timeout=tmo)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=tmo)
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"process group has {dist.get_world_size()} ranks, expected {args.gpus}")

    m = measure_training(CFG, args.dtype, args.steps, args.warmup, dev, rank, world, use_dist, events=args.events,
                         dropout=args.dropout, fp32_activations=args.fp32_activations, graph=not args.no_graph)
    if rank == 0:
        out = {
            'metric': f'train images/sec (G+D step) at {SIZE}x{SIZE} bs={BATCH_PER_GPU} per GPU', 'value': round(m['value'], 2),
            'unit': 'images/sec', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(m['ms_per_step'], 3), 'host_enqueue_ms_per_step': round(m['host_ms_per_step'], 3),
            'step_launch': ('hipGraph replay of the captured step' if m['graph'] else
                            'launch by launch, the weight gradients of each backward pass and the discriminator step\'s forward on a second stream'
                            if m['two_streams'] == 'fp32' else
                            'launch by launch, the discriminator step\'s forward on a second stream' if m['two_streams'] == 'bf16' else 'launch by launch')
                           + (' (auto, measured ms per step: ' + ', '.join(f'{k} {v:.2f}' if v is not None else f'{k} not tried' for k, v in m['step_times'].items() if k != 'host_enqueue') + ')' if m.get('step_times') else ''),
            'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None,
            'dtype': args.dtype + (' (fp32 tensors and accumulation; polyphase Winograd GEMMs in bf16x3 split form: six bf16 MFMA products per '
                                   'fp32 product)' if (args.dtype == 'f32' and m.get('split_bf16')) else ''),
            'data': 'synthetic',
            'activation_storage': m['activation_storage'], 'peak_vram_GiB': m['peak_vram_GiB'],
            'config': {'workload': CFG['desc'] + ', dropout ' + ('on' if args.dropout else 'off'),
                       'global_batch': BATCH_PER_GPU * world, 'parallelism': f'dp{world}'},
            'last_losses': m['last_losses'],
            'roofline': m['roofline'],
        }
        if m['comm'] is not None:
            out['comm'] = m['comm']
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline()
        elif world == 1:
            out['cpu_baseline'] = None
        out['conv_kernels_note'] = 'per-kernel table from one fully instrumented warm-up step; roofline from the timed region'
        out['conv_kernels'] = m['conv_kernels']
        # the other single-GPU configurations of BASELINE.json, measured AFTER the headline's timed region and JSON assembly (the
        # headline metric / config / dtype above stay cfg2 fp32): cfg4 in bf16 (the "next" row f2) and cfg5 tiled inference (f1)
        if world == 1 and not use_dist and not args.no_extra and args.config == 'cfg2' and args.dtype == 'f32':
            del m
            out['extra_configs'] = extra_configs(dev, cpu=not args.no_cpu_baseline)
        sys.stdout.flush()
        os.write(line_fd, (json.dumps(out) + '\n').encode())
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def measure_training(cfg, dtype, steps, warmup, dev, rank=0, world=1, use_dist=False, events='dominant', dropout=False,
                     fp32_activations=False, graph=True):
    """W untimed + K timed G+D steps of workload `cfg` on this rank's device (inputs resident in HBM; barrier + synchronize on
    both sides; MAX over ranks).  Returns the pieces of the JSON line: value (whole-job images/s), ms_per_step, roofline of the
    dominant conv kernel (HIP events on the launch stream inside the timed region), the per-kernel table of one fully
    instrumented warm-up step, comm statistics under data parallelism."""
    import torch
    import torch.distributed as dist
    import patchgan_amd as pg
    from patchgan_amd import engine as E
    from patchgan_amd import parallel
    batch, size = cfg['batch'], cfg['size']
    torch.cuda.reset_peak_memory_stats()
    torch.manual_seed(1234)
    G = pg.UNet(3, cfg['out_nc'], cfg['nf'], use_dropout=dropout, activation=cfg['activation'],
                final_act=cfg['final_act']).to(dev)
    D = pg.Discriminator(3 + cfg['out_nc'], cfg['ndf'], n_layers=cfg['n_layers'], norm=False).to(dev)
    if dtype == 'bf16':
        G.set_precision('bf16', bf16_storage=not fp32_activations)
        D.set_precision('bf16', bf16_storage=not fp32_activations)
    t = pg.Trainer(G, D, tempfile.mkdtemp(prefix='pgbench_'))
    t.loss_type, t.seg_alpha = cfg['loss_type'], 200
    t.gc_freeze = True            # as the patchgan_train entry point does (trainer._settle_gc): opt-in, process-global
    # as the patchgan_train entry point: a launch-bound steady-state step is replayed from a captured hipGraph (one GPU, dropout off), a
    # device-bound one stays launch by launch with the fp32 weight gradients on a second stream -- also with dropout and under data
    # parallelism (Trainer.graph = 'auto' measures and decides per kind of step)
    t.graph = 'auto' if graph else False
    t.setup_optimizers(1e-3, 1e-3)
    G.train()
    D.train()
    x, y = make_inputs(batch, rank, cfg)
    x, y = x.to(dev), y.to(dev)

    def sync():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # warm-up; the last warm-up step is profiled launch by launch to find the dominant conv kernel symbol, so that the
    # timed region only carries events around THAT kernel's launches (2 event records per launch are not free)
    wprof = E.LaunchProfiler()
    for i in range(warmup):
        E.PROFILER = wprof if i == warmup - 1 else None
        t.batch(x, y, train=True)
    E.PROFILER = None
    torch.cuda.synchronize()
    wsum = wprof.summary()
    if not wsum:                       # --warmup 0: nothing to pick the dominant kernel from: instrument every launch
        events = 'all'
    dominant = max(wsum.items(), key=lambda kv: kv[1]['ms'])[0] if wsum else None
    prof = E.LaunchProfiler(only=None if events == 'all' else dominant)
    if events != 'none' and wsum:
        per_step = sum(v['launches'] for k, v in wsum.items() if events == 'all' or k == dominant)
        if events == 'dominant':
            # a bounded sample: every launch of the dominant kernel in the first steps of the timed region, at most ~96 launches
            # (event records are stream operations: instrumenting 20+ steps costs up to 1 ms per step on a busy box)
            prof.limit = int(per_step * max(1, min(steps, 96 // max(1, int(per_step)))))
        n_ev = prof.limit if prof.limit is not None else int(per_step * steps)
        prof.reserve(2 * n_ev + 8)        # the events exist (and have been recorded once) before the timed region starts
    E.PROFILER = prof if events != 'none' else None
    graphed = after = False
    if t.graph:
        # untimed: more steps until the trainer has decided how the step is launched (after Trainer.GRAPH_WARM_STEPS launch-by-launch
        # ones): replayed from a captured graph, or left launch by launch on two streams.  Either way the timed steps carry no
        # per-launch events (a capture cannot hold them; the launch profiler turns the second stream off); the dominant kernel's
        # launches are timed on one-stream launch-by-launch steps of the same trainer right after the timed region (below)
        E.PROFILER = None
        # (under data parallelism every rank must run the SAME number of steps -- each step holds collectives -- so nothing below may
        #  depend on a rank's own clock: the tournament has a fixed length, the settling loop a fixed count, and the flags that steer
        #  the rest of this function are agreed on across ranks)
        n_decide = 3 * (2 * t.TRIAL_STEPS + 2) + 4
        for _ in range(n_decide):
            if t.graph_decided() and not use_dist:
                break
            t.batch(x, y, train=True)
        graphed, after = t.graph_captured(), t.graph_decided()
        two = 'eager2' in t.decided_modes()
        if use_dist:
            flags = torch.tensor([int(two), int(graphed)], dtype=torch.int32, device=dev)
            dist.all_reduce(flags, op=dist.ReduceOp.MAX)
            two, graphed = bool(flags[0].item()), bool(flags[1].item())
        after = graphed or two          # (decided 'eager1': nothing to settle, the timed steps carry the launch profiler as before)
        if two:
            # the first two-stream steps hold every backward operand until the streams join: the caching allocator grows for some
            # steps (device allocations synchronise: 12-15 ms per step instead of 8.7) before the pattern settles -- outside the timed
            # region, as any warm-up: groups of four steps until two consecutive groups take the same time (2 %), at most 48 steps
            # (data parallel: a fixed 16 steps)
            prev = None
            for grp in range(12):
                torch.cuda.synchronize()
                tg = time.perf_counter()
                for _ in range(4):
                    t.batch(x, y, train=True)
                torch.cuda.synchronize()
                tg = time.perf_counter() - tg
                if use_dist:
                    if grp == 3:
                        break
                    continue
                if prev is not None and abs(tg - prev) <= 0.02 * prev:
                    break
                prev = tg
        E.PROFILER = None if after else (prof if events != 'none' else None)
    pd = parallel.current()
    trace = bool(os.environ.get('PATCHGAN_BENCH_TRACE'))
    if trace:
        import gc
        _gc_t = [0.0]

        def _gc_cb(phase, info):
            if phase == 'start':
                _gc_t[0] = time.perf_counter()
            elif info.get('generation') == 2:
                sys.stderr.write(f'gc gen2 {1e3 * (time.perf_counter() - _gc_t[0]):.2f} ms collected {info.get("collected")}\n')
        gc.callbacks.append(_gc_cb)
    sync()
    t0 = time.perf_counter()
    last = None
    host_ms = 0.0
    for i in range(steps):
        cur = t.batch(x, y, train=True)
        host_ms += t.host_ms
        if trace:
            sys.stderr.write(f'step host_ms {t.host_ms:.3f} reserved_MiB {torch.cuda.memory_reserved() >> 20} segments {torch.cuda.memory_stats().get("segment.all.allocated", -1)}\n')
        if last is not None:
            last['gen']            # as Trainer.train's epoch loop: step i's losses are read once step i + 1 is enqueued
        last = cur
    t.flush()                      # the last step's (deferred, data-parallel) discriminator update belongs to the timed work
    sync()
    elapsed = time.perf_counter() - t0
    sample_note = 'every launch of the kernel in the first steps of the timed region (<= 96 launches), HIP events on the launch stream'
    if after and events != 'none' and wsum:
        # the timed steps were hipGraph replays (or two-stream steps) (no per-launch events: the HIP runtime torch brings along refuses external event-record
        # nodes inside a capture, tools/graph_event_probe.py): the sample is taken on launch-by-launch steps of the same trainer, same
        # buffers, right after the timed region -- the same kernels with the same arguments, HIP events on the launch stream
        E.PROFILER = prof
        nsample = max(1, prof.limit // max(1, int(per_step))) if prof.limit else min(steps, 4)
        if use_dist:
            nsample = min(steps, 4)          # (rank-independent: every step holds collectives)
        for _ in range(nsample):
            t.batch(x, y, train=True)
        torch.cuda.synchronize()
        sample_note = (f'{nsample} one-stream launch-by-launch steps right after the timed region (the timed steps are '
                       + ('hipGraph replays, which cannot carry per-launch events on this runtime' if graphed else
                          'two-stream steps; the launch profiler keeps everything on one stream')
                       + '): every launch of the kernel, HIP events on the launch stream')
    E.PROFILER = None
    comm = None
    if pd.on:
        # communication statistics from `csteps` EXTRA steps after the timed region (every rank runs them: they hold collectives):
        # HIP events around each collective (comm stream) and around each wait for one (compute stream).  Not inside the timed
        # region: an event record or a stream wait is a barrier packet on MI355X (5-10 us of an otherwise back-to-back kernel
        # queue, tools/dp_sync_probe.py), three of them per collective on the compute stream, and with them every bucket is
        # waited for one by one instead of once for all (parallel.GradReducer.finish)
        csteps = max(1, min(steps, 8))
        pd.timing, pd.exposed = [], []
        for _ in range(csteps):
            t.batch(x, y, train=True)
        t.flush()
        sync()
        recs, pd.timing = pd.timing, None
        waits, pd.exposed = pd.exposed, None
        tsteps, steps = steps, csteps             # (the statistics below are per sampled step)
        per = max(1, len(recs) // steps)          # collectives of one step, in issue order (the same every step)
        by_slot = [[r for r in recs[i::per]] for i in range(per)] if len(recs) == per * steps else []
        comm = {'backend': dist.get_backend(), 'ranks_in_group': dist.get_world_size(),
                'sampled_steps': csteps,
                'collectives_per_step': len(recs) / steps,
                'allreduce_ms_per_step': round(sum(e0.elapsed_time(e1) for e0, e1, _ in recs) / steps, 4),
                'allreduce_MB_per_step': round(sum(b for _, _, b in recs) / steps / 1e6, 2),
                # per collective of a step, in issue order: payload and mean duration on the comm stream
                'per_collective': [{'MB': round(rs[0][2] / 1e6, 3),
                                    'ms': round(sum(e0.elapsed_time(e1) for e0, e1, _ in rs) / len(rs), 4)} for rs in by_slot],
                # EXPOSED communication: time the consuming (compute) stream stood still waiting on a collective's `done`
                # event -- everything else of allreduce_ms_per_step ran under compute kernels
                'exposed_ms_per_step': round(sum(a.elapsed_time(b) for a, b, _ in waits) / steps, 4),
                'waits_per_step': len(waits) / steps,
                'note': 'sampled on extra steps AFTER the timed region (the timed steps carry no communication events): HIP '
                        'events on the second (comm) stream around each gradient / loss-term all-reduce; they run '
                        'under the backward and discriminator kernels of the compute stream; exposed_ms_per_step = event '
                        'pairs on the compute stream around each wait for a collective'}
        steps = tsteps
    if use_dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = tt.item()
    res = {'comm': comm, 'elapsed': elapsed, 'graph': graphed, 'two_streams': (after and not graphed) and ('fp32' if not G.engine.act_bf else 'bf16'),
           'step_times': t.step_times, 'peak_vram_GiB': round(torch.cuda.max_memory_allocated() / 2 ** 30, 3)}
    last_vals = {k: round(v, 5) for k, v in last.items()} if last is not None else None
    t.release()                    # this trainer's captured steps, workspaces and second stream go back to the allocator
    if rank != 0:
        return res
    value = batch * world * steps / elapsed
    summ = prof.summary() if events != 'none' else wsum
    # dominant kernel = the conv kernel symbol with the most device time (found in the profiled warm-up step)
    sym, d = max(summ.items(), key=lambda kv: kv[1]['ms'])
    if events == 'none':
        d = dict(d, launches=d['launches'] * steps, ms=d['ms'] * steps, flops=d['flops'] * steps, kflops=d['kflops'] * steps,
                 uflops=d['uflops'] * steps)
    timed_steps = steps if (events != 'dominant' or prof.limit is None) else max(1, d['launches'] // max(1, int(wsum[sym]['launches'])))
    if after and events != 'none' and wsum:
        timed_steps = nsample
    per_step_all = wsum                      # every conv kernel, from the profiled warm-up step
    if not per_step_all:
        per_step_all = {k: dict(launches=v['launches'] / steps, ms=v['ms'] / steps, flops=v['flops'] / steps,
                                kflops=v['kflops'] / steps, uflops=v['uflops'] / steps) for k, v in summ.items()}
    conv_ms = sum(v['ms'] for v in per_step_all.values()) * steps
    # achieved = FLOPs the kernel EXECUTED / its time (a Winograd kernel executes 2.25-2.56x fewer than the layer's
    # direct-convolution count; crediting it with those would "exceed" the peak)
    achieved = d['kflops'] / (d['ms'] * 1e-3) / 1e12
    useful = d['uflops'] / (d['ms'] * 1e-3) / 1e12      # ... without the padding of ragged edge tiles to whole tiles
    peak = FP32_MFMA_PEAK_TFLOPS if dtype == 'f32' else 2500.0     # dense MFMA peak of the multiply dtype
    # a split-bf16 kernel (k_*_s3: every fp32 product as six bf16 products, fp32 accumulate) runs on the bf16 matrix pipe: it is priced
    # against THAT pipe's dense peak with the bf16 FLOPs it issues (6 x its fp32-equivalent count); the fp32-equivalent rate is quoted too
    s3_dom = '_s3<' in sym
    fp32_equiv = achieved
    if s3_dom:
        achieved, useful, peak = 6.0 * achieved, 6.0 * useful, 2500.0
    s3_any = any('_s3<' in k for k in per_step_all)
    traffic, traffic_src = pmc_traffic(sym)
    res.update({
        'value': value, 'ms_per_step': elapsed / steps * 1e3, 'host_ms_per_step': host_ms / steps,
        'activation_storage': 'bf16' if (dtype == 'bf16' and G.engine.act_bf) else 'f32',
        'last_losses': last_vals, 'split_bf16': s3_any,
        'roofline': {'bound': 'mfma', 'kernel': sym,
                     'achieved': round(achieved, 2), 'peak': peak, 'unit': 'TFLOP/s',
                     'frac': round(achieved / peak, 4), 'achieved_useful': round(useful, 2), 'frac_useful': round(useful / peak, 4),
                     'frac_note': 'frac = FLOPs the kernel executes on the MFMA pipe (ragged Winograd edge tiles count as whole tiles) / time / '
                                  'peak; frac_useful = the same algorithm on the exact extents (no tile padding)'
                                  + ('; split-bf16 kernel: six bf16 MFMA products per fp32 product, priced in issued bf16 FLOPs against the '
                                     'dense bf16 MFMA peak' if s3_dom else '')
                                  + ('; this kernel is sampled ALONE on the chip (one-stream steps), where its K split -- sized for the timed '
                                     'two-stream steps, in which it runs beside the data-gradient chain: 128 workgroups, half a round of the chip '
                                     '-- leaves CUs idle (conv_wino.hip wgrad_s3_fill: step 6.94 ms at this split against 7.18 at the split '
                                     'that is fastest alone); the runner-up k_wino_bgemm_s3<2,2,2,2,2> is in conv_kernels'
                                     if sym.startswith('k_wino_wgrad_gemm_s3') else ''),
                     'achieved_fp32_equivalent': round(fp32_equiv, 2),
                     'fp32_equivalent_over_fp32_mfma_peak': round(fp32_equiv / FP32_MFMA_PEAK_TFLOPS, 4),
                     'traffic': traffic,
                     'traffic_source': (f'{traffic_src} (committed rocprofv3 --pmc pass of the same command; not measured '
                                        'in this run)') if traffic_src else None,
                     'launches_per_step': d['launches'] / timed_steps, 'launches_timed': d['launches'], 'sample': sample_note,
                     'avg_launch_ms': round(d['ms'] / d['launches'], 4),
                     'achieved_in_direct_conv_flops': round(d['flops'] / (d['ms'] * 1e-3) / 1e12, 2),
                     'kernel_share_of_step': round(d['ms'] / timed_steps / (elapsed / steps * 1e3), 4),
                     'all_conv_kernels_TFLOPs': round(sum(v['kflops'] for v in per_step_all.values()) * steps / (conv_ms * 1e-3) / 1e12, 2),
                     'all_conv_kernels_direct_conv_TFLOPs': round(sum(v['flops'] for v in per_step_all.values()) * steps / (conv_ms * 1e-3) / 1e12, 2),
                     'all_conv_share_of_step': round(conv_ms / steps / (elapsed / steps * 1e3), 4),
                     'step_frac_of_mfma_roofline': round(value / world * cfg['gflop_per_image'] / 1e3
                                                         / (FP32_MFMA_PEAK_TFLOPS if dtype == 'f32' else 2500.0), 4)},
        'conv_kernels': {k: {'launches_per_step': v['launches'], 'ms_per_step': round(v['ms'], 4),
                             'TFLOPs': round(v['kflops'] / (v['ms'] * 1e-3) / 1e12, 2),
                             'useful_TFLOPs': round(v['uflops'] / (v['ms'] * 1e-3) / 1e12, 2),
                             'direct_conv_TFLOPs': round(v['flops'] / (v['ms'] * 1e-3) / 1e12, 2)}
                         for k, v in sorted(per_step_all.items(), key=lambda kv: -kv[1]['ms'])},
    })
    return res


def extra_configs(dev, cpu=True):
    """BASELINE.json's remaining single-GPU configurations, for the driver's record (never the headline `value`):
      cfg4_bf16  512x512x3 -> 4-class masks, bs 8, bf16 MFMA path with bf16 activation storage: 5 warm-up + 10 timed G+D steps
      cfg5       1024x1024 image -> 25 tiles of 256x256 (overlap 0.9) through predict_image, nf = 64 fp32: 10 images after 2
                 warm-up ones, device -> host copy of the mask included; tiles/s, peak device memory, fraction of the fp32 MFMA
                 roofline, the CPU oracle's forward over the same 25 tiles.
      e2e_cfg2   the reference's training LOOP at cfg2: files -> DataLoader -> H2D -> Trainer.train's step (e2e_training)."""
    import gc
    import torch
    out = {}
    try:
        gc.collect()
        torch.cuda.empty_cache()
        m = measure_training(CONFIGS['cfg4'], 'bf16', 10, 5, dev)
        r = m['roofline']
        out['cfg4_bf16'] = {
            'workload': CONFIGS['cfg4']['desc'] + ', bf16 MFMA kernels, '
                        + m['activation_storage'] + ' activation storage, dropout off',
            'metric': 'train images/sec (G+D step) at 512x512 bs=8 per GPU', 'value': round(m['value'], 2), 'unit': 'images/sec',
            'steps': 10, 'warmup': 5, 'ms_per_step': round(m['ms_per_step'], 3), 'dtype': 'bf16', 'peak_vram_GiB': m['peak_vram_GiB'],
            'step_launch': 'two streams' if m['two_streams'] else ('graph' if m['graph'] else 'one stream'),
            'roofline': {k: r[k] for k in ('bound', 'kernel', 'achieved', 'peak', 'unit', 'frac', 'achieved_useful', 'frac_useful', 'traffic',
                                           'traffic_source', 'launches_per_step', 'avg_launch_ms', 'kernel_share_of_step', 'all_conv_kernels_TFLOPs',
                                           'step_frac_of_mfma_roofline')},
            'conv_kernels': {k: v for k, v in list(m['conv_kernels'].items())[:8]}, 'last_losses': m['last_losses']}
        del m
        out['cfg4_bf16']['cpu_baseline'] = cpu_baseline(max_seconds=20.0, cfg=CONFIGS['cfg4'], max_steps=2) if cpu else None
    except Exception as e:          # the headline line must not be lost to an extra
        out['cfg4_bf16'] = {'error': f'{type(e).__name__}: {e}'}
    try:
        # the reference CLI's DEFAULT generator (train.py:92: use_dropout=True, unet.py:26-28,63-65) at cfg2 in fp32: per-layer dropout
        # seeds are launch arguments, so the step cannot be captured -- it runs launch by launch on two streams like the headline
        gc.collect()
        torch.cuda.empty_cache()
        m = measure_training(CONFIGS['cfg2'], 'f32', 10, 5, dev, dropout=True, events='none')
        out['cfg2_dropout_on'] = {
            'workload': CONFIGS['cfg2']['desc'] + ', dropout ON (nn.Dropout(0.2) in 7 encoder + 5 decoder blocks: the patchgan_train default)',
            'metric': 'train images/sec (G+D step) at 256x256 bs=16 per GPU', 'value': round(m['value'], 2), 'unit': 'images/sec',
            'steps': 10, 'warmup': 5, 'ms_per_step': round(m['ms_per_step'], 3), 'dtype': 'f32', 'peak_vram_GiB': m['peak_vram_GiB'],
            'step_launch': 'two streams' if m['two_streams'] else ('graph' if m['graph'] else 'one stream'), 'last_losses': m['last_losses']}
        del m
    except Exception as e:
        out['cfg2_dropout_on'] = {'error': f'{type(e).__name__}: {e}'}
    try:
        # BASELINE config 1 (examples/train_coco.yaml hyper-parameters at 256 x 256, bs 4): a launch-bound step -- the trainer's decision
        # for it is the captured hipGraph (dropout off here, as in the capture's precondition)
        gc.collect()
        torch.cuda.empty_cache()
        m = measure_training(CONFIGS['cfg1'], 'f32', 20, 5, dev, events='none')
        out['cfg1'] = {
            'workload': CONFIGS['cfg1']['desc'] + ', dropout off', 'metric': 'train images/sec (G+D step) at 256x256 bs=4 per GPU',
            'value': round(m['value'], 2), 'unit': 'images/sec', 'steps': 20, 'warmup': 5, 'ms_per_step': round(m['ms_per_step'], 3),
            'host_enqueue_ms_per_step': round(m['host_ms_per_step'], 3), 'dtype': 'f32', 'peak_vram_GiB': m['peak_vram_GiB'],
            'step_launch': 'two streams' if m['two_streams'] else ('graph' if m['graph'] else 'one stream'),
            'auto_measured_ms_per_step': {k: (round(v, 3) if isinstance(v, float) else v) for k, v in m['step_times'].items() if k != 'host_enqueue'} if m.get('step_times') else None}
        del m
    except Exception as e:
        out['cfg1'] = {'error': f'{type(e).__name__}: {e}'}
    try:
        import patchgan_amd as pg
        from patchgan_amd import engine as E
        from patchgan_amd.infer import predict_image
        gc.collect()
        E.release_workspaces()
        torch.cuda.empty_cache()
        base_GiB = torch.cuda.memory_allocated() / 2 ** 30          # what earlier legs still hold (0 when they released everything)
        torch.manual_seed(1234)
        G = pg.UNet(3, 1, 64, use_dropout=False, activation='leakyrelu', final_act='sigmoid').to(dev).eval()
        img = torch.rand(3, 1024, 1024, generator=torch.Generator().manual_seed(5)).to(dev)
        for _ in range(2):
            mask = predict_image(G, img, 256, 0.9, 0.5)
        torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats()
        n = 10
        t0 = time.perf_counter()
        for _ in range(n):
            mask = predict_image(G, img, 256, 0.9, 0.5)       # returns the host mask: each call ends with its D2H copy
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        out['cfg5'] = {'workload': 'cfg5: 1024x1024x3 image -> 25 tiles of 256x256 (overlap 0.9) -> UNet nf=64 forward (fp32) -> '
                                   'overlap-averaged, thresholded mask on the host (patchgan_infer predict_image)',
                       'metric': 'tiles/sec', 'value': round(25 / dt, 1), 'unit': 'tiles/sec', 'images': n,
                       'ms_per_image': round(dt * 1e3, 3), 'dtype': 'f32',
                       'peak_vram_GiB': round(torch.cuda.max_memory_allocated() / 2 ** 30, 3),
                       'vram_held_by_earlier_legs_GiB': round(base_GiB, 3),
                       'mask_positive_fraction': round(float(mask.mean()), 4)}
        # the generator forward's direct-convolution work (190.589 GFLOP per 16 images of 256x256 at nf = 64) over the
        # whole predict_image call (gather, 25 forwards, blend, device -> host copy of the mask), against the fp32 MFMA peak
        gflop = 190.589 / 16 * 25
        out['cfg5']['roofline'] = {'bound': 'mfma', 'achieved': round(gflop / dt / 1e3, 2), 'peak': FP32_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                                   'frac': round(gflop / dt / 1e3 / FP32_MFMA_PEAK_TFLOPS, 4), 'traffic': None,
                                   'note': 'direct-convolution FLOPs of the 25 generator forwards / the whole predict_image time '
                                           '(Winograd kernels execute 2.25-4x fewer; includes tile gather, blend and the mask copy)'}
        del G, img
        out['cfg5']['cpu_baseline'] = cpu_baseline_tiles() if cpu else None
    except Exception as e:
        out['cfg5'] = dict(out.get('cfg5', {}), error=f'{type(e).__name__}: {e}')
    try:
        gc.collect()
        torch.cuda.empty_cache()
        out['e2e_cfg2'] = e2e_training(dev)
    except Exception as e:
        out['e2e_cfg2'] = {'error': f'{type(e).__name__}: {e}'}
    return out


if __name__ == '__main__':
    main()
