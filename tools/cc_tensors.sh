#!/bin/bash
# solo baseline, then two concurrent checkers:  tools/cc_tensors.sh <precision> [env assignments...]
P=$1; shift
for kv in "$@"; do export "$kv"; done
python tools/debug_cc_tensors.py save /tmp/base.pt $P > /dev/null 2>&1
(python tools/debug_cc_tensors.py check /tmp/base.pt $P 4 > gpurun_out/cct_a.log 2>&1 &)
python tools/debug_cc_tensors.py check /tmp/base.pt $P 4 > gpurun_out/cct_b.log 2>&1
sleep 8
echo "== $P $*"
grep pid gpurun_out/cct_a.log | cut -c1-400; grep pid gpurun_out/cct_b.log | cut -c1-400
