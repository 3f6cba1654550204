#!/bin/bash
# Every BASELINE.json configuration on one GPU, one JSON line each (appended to $1).
set -e
out=${1:-gpurun_out/configs.log}
: > $out
run() { echo "== $*" >> $out; timeout -k 10 400 python bench.py --no-cpu-baseline --kernel-timing all "$@" 2>/dev/null | tail -1 >> $out; }
run --steps 5 --warmup 2
run --workload punet_source --steps 5 --warmup 2
run --workload fixmatch --steps 5 --warmup 2
run --workload adamt --steps 4 --warmup 1
run --workload adamatch --channels 3 --size 704 --steps 3 --warmup 1
run --workload inference --batch 1 --size 1024 --samples 16 --steps 10 --warmup 2
run --workload mean_teacher --samples 16 --steps 4 --warmup 1
