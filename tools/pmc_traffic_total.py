#!/usr/bin/env python3
"""Total HBM bytes per STEP of a workload from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; units and the gfx950 x2
correction of FETCH_SIZE as in tools/pmc_traffic.py / MI355X_MICROARCH.md §HBM), summed over every kernel; a step is counted by
the launches of a marker kernel that runs once per step.
usage: pmc_traffic_total.py <fetch_counter_collection.csv> <write_counter_collection.csv> <marker substring> <out.json>"""
import csv
import json
import os
import sys
from collections import defaultdict


def load(path, counter, marker):
    tot, steps, per = 0.0, 0, defaultdict(float)
    seen = set()
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            v = float(row["Counter_Value"])
            tot += v
            per[row["Kernel_Name"]] += v
            key = (row.get("Dispatch_Id"), row["Kernel_Name"])
            if marker in row["Kernel_Name"] and key not in seen:
                seen.add(key)
                steps += 1
    return tot, steps, per


def main():
    fp, wp, marker, outp = sys.argv[1:5]
    f, sf, perf = load(fp, "FETCH_SIZE", marker)
    w, sw, perw = load(wp, "WRITE_SIZE", marker)
    assert sf > 0 and sw > 0, "marker kernel not found"
    fetch, write = 2.0 * 1024.0 * f / sf, 1024.0 * w / sw
    top = sorted(((2048.0 * perf.get(k, 0.0) / sf + 1024.0 * perw.get(k, 0.0) / sw, k) for k in set(perf) | set(perw)), reverse=True)[:8]
    json.dump({"note": "sum over every kernel of the run / steps (marker launches); FETCH_SIZE doubled per MI355X_MICROARCH.md §HBM; separate --pmc passes",
               "steps_counted": [sf, sw], "fetch_bytes_per_step": fetch, "write_bytes_per_step": write, "hbm_bytes_per_step": fetch + write,
               "top_kernels_bytes_per_step": {k.replace("(anonymous namespace)::", "")[:70]: round(v) for v, k in top},
               "git_head": os.environ.get("PUNET_GIT_HEAD", "unknown")}, open(outp, "w"), indent=1)
    print(f"fetch {fetch / 1e6:.1f} MB + write {write / 1e6:.1f} MB per step over {sf} / {sw} steps")


if __name__ == "__main__":
    main()
