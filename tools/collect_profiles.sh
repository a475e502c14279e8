#!/bin/bash
# Collect the rocprofv3 evidence for one round on the GPU box:  tools/collect_profiles.sh <tag> [bench.py args]
#   gpurun_out/<tag>/stats   --kernel-trace --stats of `bench.py --steps 10 --warmup 3 --events none`
#   gpurun_out/<tag>/fetch   --pmc FETCH_SIZE                     (separate passes: the TCC counters do not fit together)
#   gpurun_out/<tag>/write   --pmc WRITE_SIZE
#   gpurun_out/<tag>/mfma    --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
# then tools/summarize_profiles.py <tag> turns them into profiles/<tag>_kernel_stats.csv and profiles/<tag>_pmc_traffic.json.
set -e
TAG=${1:?tag}
shift
EXTRA="$@"          # extra bench.py arguments, e.g. --config cfg4 --dtype bf16
R=$PWD
export TMPDIR=/tmp
cd /tmp
# the kernel trace is of the DRIVER's command shape (20 timed steps behind 5 warm-up ones, hipGraph replays, the eager sampling steps after them)
ARGS="$R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra $EXTRA"
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$TAG/stats -o s --output-format csv -- python3 $ARGS > $R/gpurun_out/$TAG.stats_bench.json 2>/dev/null
# counters: launch by launch (a counter pass serialises the kernels anyway), two steps
ARGS="$R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra --events none --no-graph $EXTRA"
rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/$TAG/fetch -o f --output-format csv -- python3 $ARGS > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/$TAG/write -o w --output-format csv -- python3 $ARGS > /dev/null 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $R/gpurun_out/$TAG/mfma -o m --output-format csv -- python3 $ARGS > /dev/null 2>&1
cd $R
python3 bench.py --steps 20 --warmup 5 --no-extra $EXTRA > gpurun_out/$TAG.bench.json
ls -R gpurun_out/$TAG | head -40
