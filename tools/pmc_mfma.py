#!/usr/bin/env python3
"""Matrix-pipe utilisation per kernel from ONE rocprofv3 pass with `--pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE`.

SQ_VALU_MFMA_BUSY_CYCLES is summed over all SIMDs and counts shader cycles (= 32 per v_mfma_f32_32x32x16_bf16,
MI355X_MICROARCH.md units table). GRBM_GUI_ACTIVE comes back summed over the 8 XCDs (checked against the dispatch
timestamps of the same CSV: gui / 8 / duration = 2.0-2.5 GHz for every kernel longer than 100 us), so gui / 8 is the
kernel's duration in shader cycles at the clock DVFS actually gave it:
  utilisation_at_clock = busy / (1024 SIMDs * gui / 8)          share of the available matrix-pipe cycles that were used
  clock_ghz            = gui / 8 / (End - Start timestamps)
  utilisation_of_peak  = utilisation_at_clock * clock_ghz / 2.4  against the 2.4 GHz the datasheet peak is quoted at

usage: pmc_mfma.py <counter_collection.csv> <out.json>"""
import csv
import json
import sys
from collections import defaultdict

from pmc_traffic import short

N_SIMD, N_XCD, PEAK_GHZ = 256 * 4, 8, 2.4


def main():
    busy, gui, dur = defaultdict(float), defaultdict(float), defaultdict(float)
    launches = defaultdict(int)
    with open(sys.argv[1]) as f:
        for row in csv.DictReader(f):
            k = short(row["Kernel_Name"])
            if row["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES":
                busy[k] += float(row["Counter_Value"])
            elif row["Counter_Name"] == "GRBM_GUI_ACTIVE":
                gui[k] += float(row["Counter_Value"])
                dur[k] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
                launches[k] += 1
    out = {}
    for k in busy:
        if busy[k] == 0 or gui.get(k, 0) == 0:
            continue
        n, cyc = launches[k], gui[k] / N_XCD
        ghz = cyc / dur[k]
        out[k] = {"launches": n, "mfma_busy_cycles_per_launch": busy[k] / n, "shader_cycles_per_launch": cyc / n,
                  "duration_us_per_launch": dur[k] / n / 1e3, "clock_ghz": ghz,
                  "mfma_utilisation_at_clock": busy[k] / (N_SIMD * cyc),
                  "mfma_utilisation_of_peak": busy[k] / (N_SIMD * cyc) * ghz / PEAK_GHZ}
    json.dump({"note": "SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs), one --pmc pass; "
                       "of_peak rescales by clock / 2.4 GHz", "kernels": out}, open(sys.argv[2], "w"), indent=1, sort_keys=True)
    for k, v in sorted(out.items(), key=lambda kv: -kv[1]["mfma_busy_cycles_per_launch"] * kv[1]["launches"]):
        print(f"{k:36s} launches {v['launches']:4d}  {v['duration_us_per_launch']:7.1f} us  busy {v['mfma_busy_cycles_per_launch']/1e6:7.1f} M"
              f"  clock {v['clock_ghz']:4.2f} GHz  util {100*v['mfma_utilisation_at_clock']:5.1f} % at clock,"
              f" {100*v['mfma_utilisation_of_peak']:5.1f} % of peak")


if __name__ == "__main__":
    main()
