#!/bin/bash
# two concurrent tools/debug_cc2.py:  tools/cc2.sh <precision> <reps> <steps> <clone> [env assignments...]
P=$1; R=$2; S=$3; C=$4; shift; shift; shift; shift
for kv in "$@"; do export "$kv"; done
(timeout -k 5 500 python tools/debug_cc2.py $P $R $S $C > gpurun_out/cc2_a.log 2>&1 &)
timeout -k 10 500 python tools/debug_cc2.py $P $R $S $C > gpurun_out/cc2_b.log 2>&1
sleep 12
echo "== $P reps $R steps $S clone $C $*"
grep "pid\|rror" gpurun_out/cc2_a.log | cut -c1-200; grep "pid\|rror" gpurun_out/cc2_b.log | cut -c1-200
