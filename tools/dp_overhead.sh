#!/bin/bash
# What the data-parallel code path costs on ONE rank (RCCL collectives on the second stream, staged losses), and what the
# roofline events of the timed region cost: bench lines  ->  gpurun_out/dpo_*.json ; with "prof" as $1 also a kernel trace of the DP run
R=$PWD
export TMPDIR=/tmp
python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra > gpurun_out/dpo_plain.json 2>gpurun_out/dpo_plain.err
python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra --events none > gpurun_out/dpo_plain_noev.json 2>gpurun_out/dpo_plain.err
PATCHGAN_DP_FORCE=1 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra > gpurun_out/dpo_dp.json 2>gpurun_out/dpo_dp.err
PATCHGAN_DP_FORCE=1 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra --events none > gpurun_out/dpo_dp_noev.json 2>gpurun_out/dpo_dp.err
if [ "$1" = prof ]; then
cd /tmp
PATCHGAN_DP_FORCE=1 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/dpo_prof -o s --output-format csv -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra --events none > $R/gpurun_out/dpo_prof.json 2>/dev/null
cd $R
cp $(find gpurun_out/dpo_prof -name "*kernel_stats.csv" | head -1) gpurun_out/dpo_dp_stats.csv
cp $(find gpurun_out/dpo_prof -name "*kernel_trace.csv" | head -1) gpurun_out/dpo_dp_trace.csv
rm -rf gpurun_out/dpo_prof
fi
