#!/bin/bash
# as tools/ab.sh with the default event sampling (the driver's mode):  tools/ab_dom.sh <treeA> <treeB> <rounds> [bench.py args]
A=$1; B=$2; R=$3; shift 3
for i in $(seq $R); do
  for T in $A $B; do
    python3 $T/bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" | python3 -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']; print('$T', d['dtype'], d['ms_per_step'], d.get('host_ms_per_step'), r['kernel'], r.get('launches_timed'), r['avg_launch_ms'])"
  done
done
