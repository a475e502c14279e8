#!/bin/bash
# HBM-side traffic (L2 misses: FETCH_SIZE x 2 on gfx950, WRITE_SIZE) of every dispatch of the LAST launch-by-launch step of bench.py:
#   tools/pmc_traffic_step.sh <tag> <kernel-name-substring> [bench.py args]     -> gpurun_out/<tag>.traffic.txt
set -e
TAG=${1:?tag}; PAT=${2:?pattern}; shift 2
R=$PWD
export TMPDIR=/tmp
cd /tmp
ARGS="$R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra --events none --no-graph $@"
rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/$TAG/f -o f --output-format csv -- python3 $ARGS > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/$TAG/w -o w --output-format csv -- python3 $ARGS > /dev/null 2>&1
cd $R
python3 - "$TAG" "$PAT" > gpurun_out/$TAG.traffic.txt <<'PY'
import csv, glob, sys, re
tag, pat = sys.argv[1], sys.argv[2]
def load(sub, ctr):
    f = glob.glob(f'gpurun_out/{tag}/{sub}/**/*counter_collection.csv', recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if r['Counter_Name'] == ctr]
    rows.sort(key=lambda r: int(r['Dispatch_Id']))
    return rows
F, W = load('f', 'FETCH_SIZE'), load('w', 'WRITE_SIZE')
assert len(F) == len(W), (len(F), len(W))
n = len(F)
# the last step = the last third of the dispatches (warm-up 1 + 2 steps, all launch by launch)
names = [r['Kernel_Name'] for r in F]
adam = [i for i, nm in enumerate(names) if 'k_adam' in nm]
start = adam[len(adam) * 2 // 3 - 1] + 1 if adam else 0
for i in range(start, n):
    nm = re.sub(r'\(.*$', '', names[i].replace('void ', '').replace('(anonymous namespace)::', ''))
    if pat not in nm:
        continue
    fk, wk = float(F[i]['Counter_Value']) * 2, float(W[i]['Counter_Value'])      # KiB (FETCH_SIZE halves wide reads on gfx950)
    print(f"{i - start:4d} grid {F[i]['Grid_Size']:>9s} fetch {fk / 1024:8.1f} MiB  write {wk / 1024:8.1f} MiB  {nm[:70]}")
PY
cat gpurun_out/$TAG.traffic.txt
