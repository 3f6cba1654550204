#!/usr/bin/env python3
"""LDS bank-conflict share per kernel from ONE rocprofv3 pass with `--pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE` (next to --kernel-trace
only).  SQ_LDS_BANK_CONFLICT = extra LDS cycles spent on conflicts, SQ_LDS_IDX_ACTIVE = all cycles the LDS arrays were active
(MI355X_MICROARCH.md): conflict_share = the first over the second, summed over a kernel's launches.

usage: pmc_lds.py <counter_collection.csv> <out.json>"""
import csv
import json
import sys
from collections import defaultdict

from pmc_traffic import short


def main():
    conf, act, launches = defaultdict(float), defaultdict(float), defaultdict(int)
    with open(sys.argv[1]) as f:
        for row in csv.DictReader(f):
            k = short(row["Kernel_Name"])
            if row["Counter_Name"] == "SQ_LDS_BANK_CONFLICT":
                conf[k] += float(row["Counter_Value"])
                launches[k] += 1
            elif row["Counter_Name"] == "SQ_LDS_IDX_ACTIVE":
                act[k] += float(row["Counter_Value"])
    out = {k: {"launches": launches[k], "lds_active_cycles_per_launch": act[k] / launches[k],
               "lds_bank_conflict_cycles_per_launch": conf[k] / launches[k], "conflict_share": conf[k] / act[k]}
           for k in act if act[k] > 0 and launches[k]}
    json.dump({"note": "SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE per kernel, one --pmc pass", "kernels": out}, open(sys.argv[2], "w"),
              indent=1, sort_keys=True)
    for k, v in sorted(out.items(), key=lambda kv: -kv[1]["lds_active_cycles_per_launch"] * kv[1]["launches"])[:16]:
        print(f"{k:46s} launches {v['launches']:4d}  LDS active {v['lds_active_cycles_per_launch'] / 1e6:8.1f} M cycles  "
              f"conflicts {100 * v['conflict_share']:5.2f} %")


if __name__ == "__main__":
    main()
