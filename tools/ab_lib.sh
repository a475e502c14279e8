#!/bin/bash
# Same-box A/B of two BUILT libraries over the same tree (PATCHGAN_LIB):  tools/ab_lib.sh <libA.so> <libB.so> <rounds> [bench.py args]
A=$1; B=$2; R=$3; shift 3
for i in $(seq $R); do
  for T in $A $B; do
    PATCHGAN_LIB=$T python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra --events none "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('$T', d['dtype'], d['ms_per_step'], d['value'])"
  done
done
