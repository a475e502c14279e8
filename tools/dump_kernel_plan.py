"""Writes tests/golden/bench_kernel_plan.json: the kernel symbol(s) the planner of the CURRENT library picks for every conv call
of the benchmark configurations (tests/bench_layers.py).  Host-only (pg_conv_kernel queries).  Re-run after a deliberate planner
or kernel change; tests/test_cabi_cpu.py fails until the committed plan and the live planner agree, so that no kernel the bench
times can silently leave the per-layer parity test (tests/test_bench_layers_gpu.py)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import bench_layers as BL

plan = BL.live_plan()
with open(BL.PLAN_FILE, 'w') as f:
    json.dump(plan, f, indent=0, sort_keys=True)
    f.write('\n')
syms = sorted({s for v in plan.values() for s in v})
print(f'{len(plan)} calls, {len(syms)} kernel symbols -> {BL.PLAN_FILE}')
for s in syms:
    print('  ', s)
