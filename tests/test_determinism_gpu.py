"""Run-to-run determinism next to a SECOND process on the same GPU (what the two-rank tests' ranks, and RCCL's kernels in production, are
to each other): two processes run the same training steps back to back at the same time; inside each, every repetition must be
bit-identical to the first -- losses of every step and both weight buffers (tests/determinism_worker.py).  One process alone was always
reproducible; this situation is where round 3's LDS-staged head-gradient kernel was not (a packed multiply reading an LDS return too
early: EXPERIMENTS.md, "k_s2b_ca1_s1: cause"; the operand form itself is barred by tests/test_codeobj_cpu.py).
Detection power: with that kernel 50 of 92 repetitions of 18 bf16 steps showed an event in this very situation, so 2 x 11 compared
repetitions miss it with probability ~1e-7; for an effect ten times rarer the test would still catch it two times in three.
'fp32-two-streams': the step with the weight gradients of each backward pass on a second stream (from its 5th step on): a missing
dependency between the two streams would show as run-to-run differences exactly here.  Needs an MI355X."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize('precision', ['bf16', 'fp32', 'fp32-two-streams'])
def test_two_concurrent_processes_are_each_bitwise_reproducible(precision):
    reps, steps = 12, 18
    cmd = [sys.executable, os.path.join(ROOT, 'tests', 'determinism_worker.py'), precision, str(reps), str(steps), '0']
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT')}
    procs = [subprocess.Popen(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for _ in range(2)]
    outs = []
    try:
        for p in procs:
            out, err = p.communicate(timeout=600)
            assert p.returncode == 0, err[-2000:]
            outs.append(out)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for out in outs:
        lines = [l for l in out.splitlines() if l.startswith('pid ')]
        assert len(lines) == reps - 1, out
        assert all(l.endswith('all equal') for l in lines), out
