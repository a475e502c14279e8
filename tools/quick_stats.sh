#!/bin/bash
# rocprofv3 kernel stats of a short bench run:  tools/quick_stats.sh <tag> [bench.py args]  ->  gpurun_out/<tag>_stats.csv
TAG=${1:?tag}; shift
R=$PWD
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/qs_$TAG -o s --output-format csv -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra --events none "$@" > $R/gpurun_out/qs_$TAG.json 2>/dev/null
cd $R
cp $(find gpurun_out/qs_$TAG -name "*kernel_stats.csv" | head -1) gpurun_out/${TAG}_stats.csv
python3 tools/stats_table.py gpurun_out/${TAG}_stats.csv 13 200 | grep -i "total\|pack\|prep\|wino.*_u" 
python3 tools/stats_table.py gpurun_out/${TAG}_stats.csv 13 200 | grep -i "loss" 
