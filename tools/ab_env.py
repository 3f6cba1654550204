#!/usr/bin/env python3
"""Same-box A/B of ENVIRONMENT settings of the whole step: one child process of bench.py per setting and round, alternating, so
that box-to-box and thermal drift cancel.   usage: ab_env.py VAR value1 value2 ... [-- extra bench.py args]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
extra = []
if "--" in args:
    i = args.index("--")
    args, extra = args[:i], args[i + 1:]
var, values = args[0], args[1:]
cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "10", "--warmup", "3", "--no-cpu-baseline", "--no-other-configs",
       "--no-kernel-timing"] + extra
for rnd in range(3):
    for v in values:
        env = dict(os.environ)
        if v == "unset":
            env.pop(var, None)
        else:
            env[var] = v
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=900)
        tail = [l for l in r.stdout.strip().splitlines() if l.strip()][-1] if r.stdout.strip() else ""
        try:
            d = json.loads(tail)
            tail = f"{d['ms_per_step']} ms/step  {d['config']['ms_each_step']}"
        except Exception:
            pass
        print(f"round {rnd} {var}={v:6s} rc={r.returncode} {tail}", flush=True)
