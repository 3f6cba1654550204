#!/usr/bin/env python3
"""Per-step kernel table from a rocprofv3 --kernel-trace --stats run: tools/kstats.py <kernel_stats.csv> <steps traced>."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"GPU time per step: {tot / steps / 1e6:.3f} ms")
for r in rows[:top]:
    n = r["Name"].replace("(anonymous namespace)::", "")[:64]
    print(f"{n:64s} {int(r['Calls']) / steps:6.1f}/step {float(r['TotalDurationNs']) / steps / 1e6:7.3f} ms/step  avg {float(r['AverageNs']) / 1e3:8.1f} us")
