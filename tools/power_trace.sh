#!/bin/bash
# Socket power and shader clock of GPU 0 while a workload runs (rocm-smi polled every ~0.3 s): evidence for the
# power-limited clocks the kernels see.  usage: tools/power_trace.sh <out.log> <command...>
out=$1; shift
"$@" > "$out.workload" 2>&1 &
pid=$!
: > "$out"
while kill -0 $pid 2>/dev/null; do
    echo "t=$(date +%s.%N) $(rocm-smi -d 0 --showpower --showclocks 2>/dev/null | grep -E 'Power|sclk|mclk' | tr -s ' ' | tr '\n' ';')" >> "$out"
    sleep 0.3
done
wait $pid
