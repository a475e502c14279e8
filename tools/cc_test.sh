#!/bin/bash
# two cold processes of tools/debug_repro.py at the same time on one GPU:  tools/cc_test.sh <precision> <N> [env assignments...]
P=$1; N=$2; shift; shift
for kv in "$@"; do export "$kv"; done
(python tools/debug_repro.py $P $N 3 > gpurun_out/cc_a.log 2>&1 &)
python tools/debug_repro.py $P $N 3 > gpurun_out/cc_b.log 2>&1
sleep 6
echo "== $P N=$N $*"
grep "run 0" -A1 gpurun_out/cc_a.log | tr '\n' ' '; echo; grep "run . vs" gpurun_out/cc_a.log
grep "run 0" -A1 gpurun_out/cc_b.log | tr '\n' ' '; echo; grep "run . vs" gpurun_out/cc_b.log
