#!/bin/bash
# rocprofv3 PMC passes over two launch-by-launch steps of bench.py: what the waves of each kernel wait for IN THE STEP.
#   tools/pmc_step.sh <tag> [bench.py args]     -> gpurun_out/<tag>/{a,b,c}/...counter_collection.csv + gpurun_out/<tag>.pmc.txt
set -e
TAG=${1:?tag}; shift
R=$PWD
export TMPDIR=/tmp
cd /tmp
ARGS="$R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra --events none --no-graph $@"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVES GRBM_GUI_ACTIVE -d $R/gpurun_out/$TAG/a -o a --output-format csv -- python3 $ARGS > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS -d $R/gpurun_out/$TAG/b -o b --output-format csv -- python3 $ARGS > /dev/null 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM -d $R/gpurun_out/$TAG/c -o c --output-format csv -- python3 $ARGS > /dev/null 2>&1
cd $R
python3 tools/pmc_kernels.py gpurun_out/$TAG k_conv_bf16 k_wgrad_bf16 > gpurun_out/$TAG.pmc.txt
cat gpurun_out/$TAG.pmc.txt
