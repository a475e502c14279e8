#!/bin/bash
# Whole-step sweep of experiment switches (same device, one after the other):  tools/env_sweep.sh [bench.py args]  ->  ms per step per setting
run() { env PATCHGAN_EXPERIMENT=1 "$@" python3 bench.py --steps 30 --warmup 8 --no-cpu-baseline --events none $ARGS | python3 -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('$*', d['ms_per_step'])"; }
ARGS="$@"
run X=0
for v in PATCHGAN_SPLIT_TARGET=256 PATCHGAN_SPLIT_TARGET=768 PATCHGAN_SPLIT_TARGET=1024 PATCHGAN_TILE_REFINE=1 PATCHGAN_BGEMM_MZ=0 PATCHGAN_WINO2W_TILE=64 PATCHGAN_WINO2W_TILE=128 PATCHGAN_IN_CHUNK_MIN=256 PATCHGAN_IN_CHUNK_MIN=2048 PATCHGAN_TAPN_SPLIT=512 PATCHGAN_TAPN_SPLIT=2048 PATCHGAN_WINOW_TILE=128 PATCHGAN_WINO_TILE=1 PATCHGAN_WINO_TILE=2; do run $v; done
run X=0
