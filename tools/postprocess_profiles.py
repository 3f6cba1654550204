#!/usr/bin/env python3
"""Turn what tools/collect_profiles.sh wrote under gpurun_out/<dir> into the files profiles/rNN/ holds:
usage: postprocess_profiles.py gpurun_out/prof_r02b profiles/r02"""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_traffic import short  # noqa: E402

src, dst = sys.argv[1], sys.argv[2]
here = os.path.dirname(os.path.abspath(__file__))
os.makedirs(dst, exist_ok=True)


def one(pat):
    f = glob.glob(os.path.join(src, pat))
    assert len(f) == 1, (pat, f)
    return f[0]


shutil.copy(one("stats/*/*_kernel_stats.csv"), os.path.join(dst, "default_command_kernel_stats.csv"))
shutil.copy(one("stats/*/*_domain_stats.csv"), os.path.join(dst, "default_command_domain_stats.csv"))
shutil.copy(os.path.join(src, "bench_under_rocprof.json"), os.path.join(dst, "default_command_bench_under_rocprof.json"))
shutil.copy(os.path.join(src, "bench_default.json"), os.path.join(dst, "bench_default.json"))
shutil.copy(os.path.join(src, "graph_event_probe.txt"), os.path.join(dst, "graph_event_probe.txt"))
shutil.copy(os.path.join(src, "other_workloads.jsonl"), os.path.join(dst, "other_workloads.jsonl"))

# the roofline kernel inside the headline workload only: dispatches up to the 25th Adam step of the default command
rows = list(csv.DictReader(open(one("stats/*/*_kernel_trace.csv"))))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n_adam, tot, n = 0, 0.0, 0
per = {}
for r in rows:
    k = r["Kernel_Name"]
    if "adam_ema_dev" in k:
        n_adam += 1
        if n_adam == 25:
            break
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    if "conv3x3_x6b_kernel<3, false, false, false>" in k or "conv3x3_x6b_kernel<3, false, false>" in k or "conv3x3_x6b_kernel<3, false>" in k:
        tot += d
        n += 1
    key = short(k)
    per[key] = per.get(key, 0) + d
json.dump({"kernel": "conv3x3_x6b_kernel<3, false, false, false> (NP = 3 planes, forward, bf16 planes, tiles per image)",
           "population": "the default command's dispatches up to the 25th Adam step (= the headline workload: 1 sizing + 3 warm-up + "
                         "10 graph-replayed + 1 + 10 eagerly launched steps), from the default command's kernel_trace.csv",
           "launches": n, "avg_us": tot / n / 1e3,
           "gpu_ms_per_step_by_kernel_top": {k: round(v / 25 / 1e6, 3) for k, v in sorted(per.items(), key=lambda kv: -kv[1])[:14]},
           "gpu_ms_per_step_total": round(sum(per.values()) / 25 / 1e6, 3)},
          open(os.path.join(dst, "default_command_headline_part_roofline_kernel.json"), "w"), indent=1)

subprocess.check_call([sys.executable, os.path.join(here, "pmc_mfma.py"), one("pmc_mfma/*/*_counter_collection.csv"),
                       os.path.join(dst, "pmc_mfma_util_x6.json")])
# stamp the traffic file with the commit the collection ran at (tools/collect_profiles.sh records HEAD of the pushed snapshot in
# collected_at.txt: "<git head> <sha16 of the kernel sources on the box>")
env = dict(os.environ)
stamp = os.path.join(src, "collected_at.txt")
if os.path.exists(stamp):
    head, sha = open(stamp).read().split()[:2]
    env.update(PUNET_GIT_HEAD=head, PUNET_SOURCES_SHA16=sha)
subprocess.check_call([sys.executable, os.path.join(here, "pmc_traffic.py"), one("pmc_fetch/*/*_counter_collection.csv"),
                       one("pmc_write/*/*_counter_collection.csv"), os.path.join(dst, "pmc_hbm_traffic_x6.json")], env=env)
fi, wi = glob.glob(os.path.join(src, "pmc_fetch_inf/*/*_counter_collection.csv")), glob.glob(os.path.join(src, "pmc_write_inf/*/*_counter_collection.csv"))
if fi and wi:  # configs[4]: total HBM bytes of one inference step (marker: the MC sampler runs once per step)
    subprocess.check_call([sys.executable, os.path.join(here, "pmc_traffic_total.py"), fi[0], wi[0], "fcomb_sampler",
                           os.path.join(dst, "pmc_hbm_traffic_inference_1024.json")], env=env)
# the f16x3 fast mode: kernel stats + matrix-pipe utilisation of the same step
for pat, name in (("stats_f16x3/*/*_kernel_stats.csv", "f16x3_kernel_stats.csv"),):
    f = glob.glob(os.path.join(src, pat))
    if f:
        shutil.copy(f[0], os.path.join(dst, name))
if os.path.exists(os.path.join(src, "f16x3_bench_under_rocprof.json")):
    shutil.copy(os.path.join(src, "f16x3_bench_under_rocprof.json"), os.path.join(dst, "f16x3_bench_under_rocprof.json"))
f = glob.glob(os.path.join(src, "pmc_mfma_f16x3/*/*_counter_collection.csv"))
if f:
    subprocess.check_call([sys.executable, os.path.join(here, "pmc_mfma.py"), f[0], os.path.join(dst, "pmc_mfma_util_f16x3.json")])
print(open(os.path.join(dst, "default_command_headline_part_roofline_kernel.json")).read())
