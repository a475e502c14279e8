#!/bin/bash
# two concurrent tools/forensics/debug_cc5.py:  tools/cc5.sh <reps> <steps> [env assignments...]
R=$1; S=$2; shift; shift
for kv in "$@"; do export "$kv"; done
(timeout -k 5 540 python tools/forensics/debug_cc5.py $R $S > gpurun_out/cc5_a.log 2>&1 &)
timeout -k 10 540 python tools/forensics/debug_cc5.py $R $S > gpurun_out/cc5_b.log 2>&1
sleep 15
echo "== reps $R steps $S $*"
grep -v "done (" gpurun_out/cc5_a.log | cut -c1-420; grep -v "done (" gpurun_out/cc5_b.log | cut -c1-420
