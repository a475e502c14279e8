#!/bin/bash
# in-step A/B of fp32 experiment knobs at cfg2:  tools/fp32_env_sweep.sh "<env assignments>" ...
for v in "$@"; do
  r=$(env PATCHGAN_EXPERIMENT=1 $v python bench.py --steps 30 --warmup 5 --no-extra --no-cpu-baseline --events none 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")
  echo "cfg2 fp32 [$v] $r ms/step"
done
