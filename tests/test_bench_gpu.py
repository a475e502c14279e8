"""bench.py end to end on the GPU box: the plain N = 1 line, the self-spawned multi-rank launch (`--gpus 2` starts its own
ranks; rehearsed on one GPU over gloo), and the RCCL code path itself (backend "nccl": a one-rank group with the
data-parallel path forced on, and a real 2-rank run when two GPUs are visible).  Needs an MI355X."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(extra_args, extra_env, timeout=900, extras=False, cpu=False):
    env = dict(os.environ, **extra_env)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT'):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '3', '--warmup', '2']
                       + ([] if cpu else ['--no-cpu-baseline']) + ([] if extras else ['--no-extra']) + extra_args,
                       env=env, capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith('{'), p.stdout      # stdout carries the ONE JSON line and nothing else
    return json.loads(lines[0])


def test_bench_single_gpu_line():
    out = _bench([], {})
    assert out['n_gpus'] == 1 and out['unit'] == 'images/sec' and out['value'] > 100
    # the 'auto' policy is a tournament: each candidate's measured ms per step is quoted in the line, and the step is launched the way
    # that measured fastest (cfg2 fp32: two streams, 8.3 against 8.9 ms on one; cfg1: two streams as well, 2.5 against 2.9 for one stream
    # or the graph -- the graph is only tried where the one-stream step is nearly host-bound)
    import re

    def decided(o):
        m = re.search(r'auto, measured ms per step: ([^)]*)\)', o['step_launch'])
        assert m, o['step_launch']
        ms = {kv.rsplit(' ', 1)[0]: float(kv.rsplit(' ', 1)[1]) for kv in m.group(1).split(', ') if not kv.endswith('not tried')}
        best = min(ms, key=ms.get)
        want = {'eager1': 'launch by launch (', 'eager2': 'launch by launch, the weight gradients of each backward pass and the discriminator',
                'graph': 'hipGraph'}[best]
        assert o['step_launch'].startswith(want), o['step_launch']
        return best
    assert out['roofline']['launches_timed'] >= 8
    assert decided(out) in ('eager2', 'eager1')          # (never the graph: the cfg2 step is device-bound four times over)
    small = _bench(['--config', 'cfg1'], {})
    if decided(small) == 'graph':
        assert small['host_enqueue_ms_per_step'] < 1.0, small['step_launch']
    r = out['roofline']
    assert r['bound'] == 'mfma' and 0.05 < r['frac'] < 1.0 and r['kernel'].startswith('k_')
    assert 'comm' not in out
    # every conv kernel the bench times is pinned per layer at this geometry and batch size (tests/test_bench_layers_gpu.py)
    from tests import bench_layers as BL
    planned = {s for k, v in BL.committed_plan().items() if k.startswith('cfg2-fp32-') for s in v}
    assert set(out['conv_kernels']) <= planned, set(out['conv_kernels']) - planned


def test_bench_extra_configs_ride_on_the_headline_line():
    """The default N = 1 run also measures BASELINE.json's other single-GPU configurations after the headline's timed region -- cfg4
    in bf16 (images/s, dominant-kernel fraction of the bf16 MFMA peak) and cfg5 tiled inference (tiles/s, peak VRAM) -- under
    `extra_configs` of the same single line; the headline metric / config / dtype stay cfg2 fp32.  Each carries its CPU-oracle leg
    (bounded), cfg5 its fraction of the fp32 MFMA roofline, and `e2e_cfg2` the loader-fed rate of Trainer.train for both item formats."""
    out = _bench([], {}, extras=True, cpu=True)
    assert out['dtype'].startswith('f32') and out['config']['workload'].startswith('cfg2') and '256x256 bs=16' in out['metric']
    x4, x5 = out['extra_configs']['cfg4_bf16'], out['extra_configs']['cfg5']
    assert 'error' not in x4 and 'error' not in x5, out['extra_configs']
    assert x4['dtype'] == 'bf16' and x4['value'] > 100 and x4['roofline']['peak'] == 2500.0 and 0.02 < x4['roofline']['frac'] < 1.0
    assert 'bf16' in x4['roofline']['kernel'] and 'bf16 activation storage' in x4['workload']
    assert x5['unit'] == 'tiles/sec' and x5['value'] > 500 and 0.5 < x5['peak_vram_GiB'] < 40
    assert out['cpu_baseline']['kind'] == 'port' and 0 < out['cpu_baseline']['value'] < out['value']
    assert x4['cpu_baseline']['unit'] == 'images/sec' and 0 < x4['cpu_baseline']['value'] < x4['value']
    assert x5['cpu_baseline']['unit'] == 'tiles/sec' and 0 < x5['cpu_baseline']['value'] < x5['value']
    assert x5['roofline']['peak'] == 157.3 and 0.05 < x5['roofline']['frac'] < 1.0
    r = out['roofline']
    assert 0 < r['frac_useful'] <= r['frac'] < 1.0
    xd, x1 = out['extra_configs']['cfg2_dropout_on'], out['extra_configs']['cfg1']
    assert 'error' not in xd and 'error' not in x1, (xd, x1)
    # the reference CLI's default generator (dropout on) rides the same schedule as the headline: within 10 % of it in this short run
    assert xd['value'] > 0.9 * out['value'] and xd['step_launch'] in ('two streams', 'one stream')
    assert x1['value'] > 100 and x1['step_launch'] in ('graph', 'two streams', 'one stream')
    e = out['extra_configs']['e2e_cfg2']
    assert 'error' not in e, e
    for fmt in ('float', 'u8_device_pipeline'):
        assert e[fmt]['steps_timed'] >= 30 and 50 < e[fmt]['value'] <= out['value'] * 1.05, e
    assert e['u8_device_pipeline']['h2d_bytes_per_image'] * 4 == e['float']['h2d_bytes_per_image']


def test_bench_rccl_path_one_rank():
    """backend "nccl" (= RCCL) really executes: a one-rank process group with the data-parallel path forced on runs every
    collective of the step (bucketed gradient all-reduces on the comm stream, loss terms, the deferred discriminator update)
    through RCCL and must reproduce the single-process losses."""
    plain = _bench(['--no-graph'], {})      # launch by launch on one stream: the same number of steps before the last timed one
    out = _bench(['--no-graph'], {'PATCHGAN_DP_FORCE': '1', 'MASTER_PORT': '29631'})
    assert plain['step_launch'] == out['step_launch'] == 'launch by launch'
    assert out['comm']['backend'] == 'nccl' and out['comm']['ranks_in_group'] == 1
    assert out['comm']['collectives_per_step'] >= 7            # 6 x 32 MiB G buckets + D gradient + loss terms
    assert out['comm']['allreduce_MB_per_step'] > 170          # 167 MB + 11 MB of gradients
    for k, v in plain['last_losses'].items():
        assert abs(out['last_losses'][k] - v) <= 1e-5 * max(abs(v), 1e-3), (k, out['last_losses'], plain['last_losses'])
    # the default launch policy under data parallelism: never a captured graph (collectives stay outside), and a device-bound step
    # runs on two streams like the single-process one (the bucket all-reduce waits for the second stream's weight gradients)
    import re
    two = _bench([], {'PATCHGAN_DP_FORCE': '1', 'MASTER_PORT': '29632'})
    assert not two['step_launch'].startswith('hipGraph') and two['comm']['collectives_per_step'] >= 7
    assert 'auto, measured ms per step: eager1' in two['step_launch'] and 'graph' not in two['step_launch'].split('measured')[1], two['step_launch']


def test_bench_spawns_its_own_ranks_gloo_rehearsal():
    """`python bench.py --gpus 2` with no launcher: the parent (which never touches the GPU) starts two ranks and forwards rank
    0's line.  On a one-GPU box both ranks share device 0 and the collectives go through gloo."""
    out = _bench(['--gpus', '2'], {'PATCHGAN_SHARE_GPU': '1', 'PATCHGAN_DIST_BACKEND': 'gloo'})
    assert out['n_gpus'] == 2 and out['config']['global_batch'] == 32 and out['scaling'] == 'weak'
    assert out['comm']['ranks_in_group'] == 2 and out['comm']['backend'] == 'gloo'
    assert 'cpu_baseline' not in out


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason='needs two GPUs: one rank per device over RCCL')
def test_bench_two_ranks_over_rccl():
    out = _bench(['--gpus', '2'], {})
    assert out['n_gpus'] == 2 and out['comm']['backend'] == 'nccl' and out['comm']['ranks_in_group'] == 2
    assert out['value'] > 100
