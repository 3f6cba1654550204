#!/usr/bin/env python3
"""Same-box A/B of library variants (tools/ab_variants.sh): one child process per variant and round, alternating, so that
box-to-box and thermal drift cancel.  usage: ab_run.py wgrad|layers|step name1 name2 ... (name 'base' = libpunet_hip.so)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "probabilistic_domain_adaptation_amd")
what, names = sys.argv[1], sys.argv[2:]
cmd = {"wgrad": [sys.executable, os.path.join(ROOT, "tools", "time_wgrad.py")],
       "fcomb": [sys.executable, os.path.join(ROOT, "tools", "time_fcomb_train.py")],
       "sampler": [sys.executable, os.path.join(ROOT, "tools", "time_sampler.py")],
       "upbwd": [sys.executable, os.path.join(ROOT, "tools", "time_upsample_bwd.py")],
       "layers": [sys.executable, os.path.join(ROOT, "tools", "bench_x6_layers.py"), "--rounds", "3"],
       "step": [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "8", "--warmup", "2", "--no-cpu-baseline",
                "--no-other-configs", "--no-kernel-timing"]}[what]
for rnd in range(2):
    for n in names:
        lib = os.path.join(PKG, "libpunet_hip.so" if n == "base" else f"libpunet_hip_{n}.so")
        env = dict(os.environ, PUNET_HIP_LIB=lib)
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
        tail = [l for l in r.stdout.strip().splitlines() if l.strip()][-1] if r.stdout.strip() else ""
        if what == "step":
            import json
            try:
                d = json.loads(tail)
                tail = f"{d['ms_per_step']} ms/step  {d['config']['ms_each_step']}"
            except Exception:
                pass
        print(f"round {rnd} {n:12s} rc={r.returncode} {tail}", flush=True)
