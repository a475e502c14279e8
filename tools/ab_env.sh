#!/bin/bash
# Same-device A/B of two environment settings on ONE tree:  tools/ab_env.sh "<ENV_A>" "<ENV_B>" <rounds> [bench.py args]
# (each setting "VAR=x VAR2=y" or "-" for none; PATCHGAN_EXPERIMENT=1 is added; prints ms_per_step per run, alternating)
A=$1; B=$2; R=$3; shift 3
for i in $(seq $R); do
  for E in "$A" "$B"; do
    if [ "$E" = "-" ]; then EE=""; else EE="$E"; fi
    env PATCHGAN_EXPERIMENT=1 $EE python3 bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-extra --events none "$@" | python3 -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('[$E]', d['ms_per_step'], d['value'], d['step_launch'][:40])"
  done
done
