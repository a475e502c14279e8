"""Idle gaps of the compute queue in a rocprofv3 kernel trace (last full steps): usage: python tools/trace_gaps.py <kernel_trace.csv> [min_us]"""
import csv, collections, re, sys


def short(n):
    n = n.replace('void ', '').replace('(anonymous namespace)::', '')
    return re.sub(r'\(.*$', '', n)[:38]


rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
mainq = collections.Counter(r['Queue_Id'] for r in rows).most_common(1)[0][0]
adam = [i for i, r in enumerate(rows) if 'k_adam' in r['Kernel_Name'] and int(r['Grid_Size_X']) > 1000000]   # Adam(G): one per step
s, e = adam[-3], adam[-1]
t0 = int(rows[s]['End_Timestamp'])
prev, tot, n = None, 0.0, 0
for r in rows[s + 1:e + 1]:
    st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    tag = ''
    if r['Queue_Id'] == mainq:
        if prev is not None and (st - prev) / 1e3 > min_us:
            tag = f'  <== gap {(st - prev) / 1e3:.1f}'
            tot += (st - prev) / 1e3
            n += 1
        prev = en if prev is None else max(prev, en)
    if r['Queue_Id'] != mainq or tag:
        print(f"q{r['Queue_Id']} t={(st - t0) / 1e3:8.1f} d={(en - st) / 1e3:6.1f} {short(r['Kernel_Name'])}{tag}")
print(f'2 steps: {n} gaps > {min_us} us, {tot:.1f} us in all ({tot / 2:.1f} per step); span {(int(rows[e]["End_Timestamp"]) - t0) / 2e3:.1f} us per step')
