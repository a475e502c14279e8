import os
import sys

# before anything initialises the HIP runtime (the collection hook below asks torch for a GPU): the setting the package ships with
# (patchgan_amd/__init__.py: one hardware queue per stream of the step)
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def usable_cpus():
    """CPU threads this process may really use: min(affinity, cgroup quota).  The GPU box exposes 256 logical CPUs but
    a 16-CPU quota; 256 oneDNN threads on 16 CPUs run ~20x slower than 16."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    try:
        import torch
        torch.set_num_threads(min(usable_cpus(), 16))
    except Exception:
        pass


def pytest_collection_modifyitems(config, items):
    """GPU tests are skipped (not failed) when no GPU is visible, so `-m gpu` here is a no-op."""
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
