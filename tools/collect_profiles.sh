#!/bin/bash
# One call on the GPU box: everything profiles/rNN/ cites (kernel stats of the default bench command, the two PMC families
# in separate passes - never combined with tracing domains other than --kernel-trace -, the other workloads, the graph probe).
out=${1:-gpurun_out/prof_r05}
mkdir -p $out
# what the counters belong to: the commit handed in by the caller (the box has no .git) and a digest of the kernel sources here
echo "${PUNET_GIT_HEAD:-unknown} $(cat probabilistic_domain_adaptation_amd/csrc/conv_x6.hip probabilistic_domain_adaptation_amd/csrc/wgrad_x6.hip probabilistic_domain_adaptation_amd/csrc/common.h | sha256sum | cut -c1-16)" > $out/collected_at.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 tools/probe_graph_events.py > $out/graph_event_probe.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py > $out/bench_under_rocprof.json 2> $out/stats.err
python3 bench.py > $out/bench_default.json 2> $out/bench_default.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/pmc_mfma -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs --no-kernel-timing > $out/pmc_mfma_bench.json 2> $out/pmc_mfma.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs --no-kernel-timing > /dev/null 2> $out/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmc_write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs --no-kernel-timing > /dev/null 2> $out/pmc_write.err
# configs[4]: HBM bytes of one N = 16 inference step on a 1024 x 1024 tile, every kernel (bench.py's hbm_report reads the sum)
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_fetch_inf -- python3 bench.py --workload inference --batch 1 --size 1024 --samples 16 --no-graph --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs --no-kernel-timing > /dev/null 2> $out/pmc_fetch_inf.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmc_write_inf -- python3 bench.py --workload inference --batch 1 --size 1024 --samples 16 --no-graph --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs --no-kernel-timing > /dev/null 2> $out/pmc_write_inf.err
for w in "--dice" "--rehearse-rccl" "--workload punet_source" "--workload fixmatch" "--workload inference --batch 1 --size 1024 --samples 16" "--workload mean_teacher --samples 16" "--conv-mode f16x3" "--conv-mode f16" "--conv-mode f16s" "--conv-mode x3" "--conv-mode bf16" "--conv-mode bf16s" "--conv-mode f16x3 --workload inference --batch 1 --size 1024 --samples 16" "--conv-mode f16s --workload inference --batch 1 --size 1024 --samples 16" "--conv-mode bf16s --workload inference --batch 1 --size 1024 --samples 16" "--conv-mode f32 --steps 4 --warmup 1"; do
  echo "== $w" >> $out/other_workloads.jsonl
  timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-other-configs $w 2>/dev/null | tail -1 >> $out/other_workloads.jsonl
done
# the f16x3 fast mode: kernel stats and matrix-pipe utilisation of the same step
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_f16x3 -- python3 bench.py --conv-mode f16x3 --no-graph --steps 10 --warmup 2 --no-cpu-baseline --no-other-configs --no-kernel-timing > $out/f16x3_bench_under_rocprof.json 2> $out/stats_f16x3.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/pmc_mfma_f16x3 -- python3 bench.py --conv-mode f16x3 --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs --no-kernel-timing > /dev/null 2> $out/pmc_mfma_f16x3.err
ls -R $out | head -60
