#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE counter CSVs (two separate passes, as
/opt/skills/guides/MI355X_MICROARCH.md §HBM prescribes) into per-launch HBM bytes per kernel.

Units/corrections from the guide: FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports exactly half of the
bytes of a wide coalesced (16 B/lane) streaming read -> doubled here (all staging loads of the MFMA kernels are float4);
WRITE_SIZE is exact for 16 B/lane stores (the conv epilogue stores 4 B/lane in 128-B runs: uncalibrated, taken as is).

usage: pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>"""
import csv
import json
import sys
from collections import defaultdict


def load(path, counter):
    agg = defaultdict(lambda: [0.0, 0])
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            a = agg[row["Kernel_Name"]]
            a[0] += float(row["Counter_Value"])
            a[1] += 1
    return agg


def short(name):
    """Kernel name without the namespace and argument list, template arguments kept:
    'void (anonymous namespace)::conv3x3_x6_kernel<3, false, 3>(...)' -> 'conv3x3_x6_kernel<3, false, 3>'."""
    import re
    m = re.search(r"([A-Za-z_][A-Za-z0-9_]*(?:<[^()]*>)?)\(", name.replace("(anonymous namespace)::", ""))
    return (m.group(1) if m else name)[:60]


def main():
    fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in set(fetch) | set(write):
        f, w = fetch.get(k, [0.0, 0]), write.get(k, [0.0, 0])
        n = max(f[1], w[1], 1)
        out[short(k)] = {"launches": n, "fetch_bytes_per_launch": 2.0 * 1024.0 * f[0] / max(f[1], 1),
                         "write_bytes_per_launch": 1024.0 * w[0] / max(w[1], 1)}
        out[short(k)]["hbm_bytes_per_launch"] = out[short(k)]["fetch_bytes_per_launch"] + out[short(k)]["write_bytes_per_launch"]
    # what the numbers belong to: the commit the collecting script was handed ($PUNET_GIT_HEAD; the GPU box has no .git) and a
    # digest of the kernel sources as they were when the counters ran - bench.py compares it with the sources it runs
    import hashlib
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    for f in ("conv_x6.hip", "wgrad_x6.hip", "common.h"):
        h.update(open(os.path.join(root, "probabilistic_domain_adaptation_amd", "csrc", f), "rb").read())
    json.dump({"note": "FETCH_SIZE doubled per MI355X_MICROARCH.md §HBM; separate --pmc passes", "kernels": out,
               "git_head": os.environ.get("PUNET_GIT_HEAD", "unknown"),
               "kernel_sources_sha16": os.environ.get("PUNET_SOURCES_SHA16") or h.hexdigest()[:16]},
              open(sys.argv[3], "w"), indent=1, sort_keys=True)
    for k, v in sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"])[:8]:
        print(f"{k:40s} launches {v['launches']:4d}  fetch {v['fetch_bytes_per_launch']/1e6:9.1f} MB  write "
              f"{v['write_bytes_per_launch']/1e6:9.1f} MB per launch")


if __name__ == "__main__":
    main()
