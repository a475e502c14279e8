#!/usr/bin/env python3
"""Turn the rocprofv3 output of tools/collect_profiles.sh <tag> (merged back under gpurun_out/<tag>/) into the committed
evidence: profiles/<tag>_kernel_stats.csv, profiles/<tag>_pmc_traffic.json, profiles/<tag>_bench.json."""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src = os.path.join(ROOT, 'gpurun_out', tag)


def short(name):
    """'void (anonymous namespace)::k_b2s_fast<2, 2, 2, 2, false>(float const*, ...)' -> 'k_b2s_fast<2,2,2,2,false>'"""
    n = re.sub(r'^void ', '', name)
    n = n.replace('(anonymous namespace)::', '')
    depth, out = 0, ''
    for ch in n:
        if ch == '<':
            depth += 1
        if ch == '(' and depth == 0:
            break
        if ch == '>':
            depth -= 1
        out += ch
    return out.replace(' ', '')[:120]


def one(pattern):
    f = glob.glob(os.path.join(src, pattern), recursive=True)
    if not f:
        raise SystemExit(f'missing {pattern} under {src}')
    return f[0]


shutil.copy(one('stats/**/*kernel_stats.csv'), os.path.join(ROOT, 'profiles', f'{tag}_kernel_stats.csv'))


def alone_table():
    """rocprofv3's kernel trace does NOT serialise the two streams of the timed steps (round 6: dispatches of the second stream's queue overlap
    the compute stream's), so the per-kernel averages of *_kernel_stats.csv mix launches that ran alone with launches that shared the chip.
    This table keeps, per kernel symbol, the launches whose [start, end] overlaps no other dispatch -- the figure that bench.py's HIP events on
    one-stream sampling steps measure -- next to the average over all launches."""
    rows = list(csv.DictReader(open(one('stats/**/*kernel_trace.csv'))))
    iv = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name'])) for r in rows)
    acc = collections.defaultdict(lambda: [0, 0.0, 0, 0.0])
    for i, (s0, e0, name) in enumerate(iv):
        shared = (i > 0 and max(x[1] for x in iv[max(0, i - 8):i]) > s0) or (i + 1 < len(iv) and iv[i + 1][0] < e0)
        a = acc[name]
        a[0] += 1
        a[1] += e0 - s0
        if not shared:
            a[2] += 1
            a[3] += e0 - s0
    out = os.path.join(ROOT, 'profiles', f'{tag}_kernel_alone.csv')
    with open(out, 'w') as f:
        f.write('kernel,launches,avg_us_all,launches_alone_on_the_chip,avg_us_alone\n')
        for name, a in sorted(acc.items(), key=lambda kv: -kv[1][1]):
            f.write(f'"{name}",{a[0]},{a[1] / a[0] / 1e3:.1f},{a[2]},{(a[3] / a[2] / 1e3) if a[2] else float("nan"):.1f}\n')


alone_table()


def counters(sub):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(one(f'{sub}/**/*counter_collection.csv'))):
        acc[short(r['Kernel_Name'])][r['Counter_Name']].append(float(r['Counter_Value']))
    return acc


fetch, write, mfma = counters('fetch'), counters('write'), counters('mfma')
kernels = {}
for k in sorted(fetch):
    f = fetch[k]['FETCH_SIZE']
    w = write.get(k, {}).get('WRITE_SIZE', [0.0])
    e = {'launches': len(f), 'FETCH_SIZE_KiB_per_launch': round(sum(f) / len(f), 1),
         'WRITE_SIZE_KiB_per_launch': round(sum(w) / len(w), 1)}
    e['hbm_bytes_per_launch'] = int((2 * e['FETCH_SIZE_KiB_per_launch'] + e['WRITE_SIZE_KiB_per_launch']) * 1024)
    m = mfma.get(k)
    if m and sum(m['SQ_VALU_MFMA_BUSY_CYCLES']) > 0:
        busy = sum(m['SQ_VALU_MFMA_BUSY_CYCLES']) / len(m['SQ_VALU_MFMA_BUSY_CYCLES'])
        gui = sum(m['GRBM_GUI_ACTIVE']) / len(m['GRBM_GUI_ACTIVE'])
        # busy cycles are summed over the 1024 SIMDs, GRBM_GUI_ACTIVE over the 8 XCDs
        e['mfma_pipe_utilisation'] = round((busy / 1024) / (gui / 8), 3)
    kernels[k] = e
note = ('rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE in separate passes over '
        '`bench.py --steps 2 --warmup 1 --events none` (tools/collect_profiles.sh). Counter unit = KiB. hbm_bytes_per_launch = '
        '(2*FETCH_SIZE + WRITE_SIZE)*1024: MI355X_MICROARCH.md section HBM -- on gfx950 FETCH_SIZE reports exactly half of '
        'the bytes of 16-B-per-lane loads; WRITE_SIZE is exact. Infinity-Cache hits are counted (fabric-side requests), so '
        'this is an upper bound on DRAM traffic. mfma_pipe_utilisation = MFMA busy cycles per SIMD / kernel cycles.')
json.dump({'_note': note, 'kernels': kernels}, open(os.path.join(ROOT, 'profiles', f'{tag}_pmc_traffic.json'), 'w'), indent=1)
b = os.path.join(ROOT, 'gpurun_out', f'{tag}.bench.json')
if os.path.exists(b):
    shutil.copy(b, os.path.join(ROOT, 'profiles', f'{tag}_bench.json'))
top = sorted(kernels.items(), key=lambda kv: -kv[1]['hbm_bytes_per_launch'] * kv[1]['launches'])[:8]
for k, e in top:
    print(k, e)
