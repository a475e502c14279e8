#!/usr/bin/env python3
"""Per-kernel averages of every counter found in the rocprofv3 counter_collection CSVs under a directory
(tools/pmc_layer.sh output): one line per kernel symbol, counters per launch."""
import collections
import csv
import glob
import os
import re
import sys


def short(name):
    n = re.sub(r'^void ', '', name).replace('(anonymous namespace)::', '')
    depth, out = 0, ''
    for ch in n:
        if ch == '<':
            depth += 1
        if ch == '(' and depth == 0:
            break
        if ch == '>':
            depth -= 1
        out += ch
    return out.replace(' ', '')[:60]


acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(sys.argv[1], '**', '*counter_collection.csv'), recursive=True):
    per_dispatch = collections.defaultdict(float)
    names = {}
    for r in csv.DictReader(open(f)):
        key = (f, r['Dispatch_Id'], r['Counter_Name'])
        per_dispatch[key] += float(r['Counter_Value'])
        names[(f, r['Dispatch_Id'])] = short(r['Kernel_Name'])
    for (ff, d, c), v in per_dispatch.items():
        acc[names[(ff, d)]][c].append(v)
only = sys.argv[2:] or None
for k in sorted(acc):
    if only and not any(o in k for o in only):
        continue
    if not (k.startswith('k_')):
        continue
    c = {n: sum(v) / len(v) for n, v in acc[k].items()}
    line = f"{k:52s} n={len(next(iter(acc[k].values()))):3d} "
    wc = c.get('SQ_WAVE_CYCLES')
    for n in sorted(c):
        line += f" {n}={c[n]:.3g}"
    if wc:
        for n in ('SQ_WAIT_INST_ANY', 'SQ_WAIT_ANY'):
            if n in c:
                line += f" | {n}/WAVE_CYCLES={c[n] / wc:.2f}"
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in c and 'GRBM_GUI_ACTIVE' in c:
        line += f" | mfma_util={(c['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024) / (c['GRBM_GUI_ACTIVE'] / 8):.3f}"
    print(line)
