#!/bin/bash
# Same-device A/B of two trees:  tools/ab.sh <treeA> <treeB> <rounds> [bench.py args]   (prints ms_per_step per run)
A=$1; B=$2; R=$3; shift 3
for i in $(seq $R); do
  for T in $A $B; do
    python3 $T/bench.py --steps 30 --warmup 8 --no-cpu-baseline --events none "$@" | python3 -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('$T', d['dtype'], d['ms_per_step'], d['value'])"
  done
done
