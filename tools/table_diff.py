"""Rows of two tools/step_launch_table.py outputs that differ by more than a few percent:  python tools/table_diff.py a.txt b.txt [pct=4]"""
import sys


def rows(p):
    out = {}
    for l in open(p):
        f = l.split()
        if len(f) >= 12 and f[0].isdigit():
            out[int(f[0])] = (f[1], f[3], f[4], f[5], int(f[7]), float(f[8]), ' '.join(f[11:]))
    return out


a, b = rows(sys.argv[1]), rows(sys.argv[2])
pct = float(sys.argv[3]) if len(sys.argv) > 3 else 4.0
ta = tb = 0.0
for i in sorted(a):
    if i not in b:
        continue
    ta += a[i][5]
    tb += b[i][5]
    if abs(b[i][5] - a[i][5]) > pct / 100 * a[i][5]:
        print(f'{i:3d} op {a[i][0]} Hb {a[i][1]:>4s} Ca {a[i][2]:>5s} Cb {a[i][3]:>5s} split {a[i][4]:4d} -> {b[i][4]:4d}  {a[i][5]:7.1f} -> {b[i][5]:7.1f} us  {a[i][6]} -> {b[i][6]}')
print(f'sum {ta:.1f} -> {tb:.1f} us')
