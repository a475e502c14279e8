#!/bin/bash
# ms_per_step and host enqueue ms of one bench run:  tools/hostms.sh <label> [bench.py args]
L=$1; shift
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" | python3 -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('$L', d['dtype'], d['ms_per_step'], 'host', d.get('host_enqueue_ms_per_step'))"
