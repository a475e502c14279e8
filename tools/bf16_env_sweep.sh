#!/bin/bash
# in-step A/B of the bf16 planner's knobs (the back-to-back layer benchmark ranks tiles differently from the training step):
#   tools/bf16_env_sweep.sh "<env assignments>" ...      each argument is one variant; cfg4 and cfg2 bf16 ms per step
for v in "$@"; do
  for c in cfg4 cfg2; do
    r=$(env PATCHGAN_EXPERIMENT=1 $v python bench.py --config $c --dtype bf16 --steps 20 --warmup 5 --no-extra --no-cpu-baseline --events none 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")
    echo "$c [$v] $r ms/step"
  done
done
