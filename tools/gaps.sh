#!/bin/bash
# kernel trace of a short bench run, then the idle gaps of the busiest queue:  tools/gaps.sh <tag> [env assignments]
TAG=$1; shift
R=$PWD
export TMPDIR=/tmp
cd /tmp
env PATCHGAN_EXPERIMENT=1 "$@" rocprofv3 --kernel-trace -d $R/gpurun_out/gaps_$TAG -o g --output-format csv -- python3 $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-extra --events none > $R/gpurun_out/gaps_$TAG.json 2>/dev/null
cd $R
python3 tools/trace_gaps.py $(find gpurun_out/gaps_$TAG -name "*kernel_trace.csv" | head -1) 3.0 | tail -3
python3 -c "import json; d=json.load(open('gpurun_out/gaps_$TAG.json')); print(d['ms_per_step'], d['step_launch'][:50])"
