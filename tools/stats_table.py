#!/usr/bin/env python3
"""Print a rocprofv3 kernel_stats.csv as ms per step:  tools/stats_table.py <csv> [steps_in_trace=13] [rows=24]"""
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 13.0
top = int(sys.argv[3]) if len(sys.argv) > 3 else 24
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms/step", round(tot / steps / 1e6, 3))
for r in rows[:top]:
    n = r["Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:56]
    print(f"{n:58s} {int(r['Calls']) / steps:6.1f}/step {float(r['TotalDurationNs']) / steps / 1e6:7.4f} ms")
