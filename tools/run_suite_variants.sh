#!/bin/bash
# The whole GPU suite under every switch that selects between equivalent implementations (profiles/rNN/suite_variants.txt): three
# gpurun calls of two suites each (a suite takes 5.5 - 7 minutes on a box, a call at most 20).  Run from the build container AFTER
# building (python __graft_entry__.py); do not touch kernel sources while it runs (every call snapshots the tree).
#   tools/run_suite_variants.sh [tag]      -> gpurun_out/<tag>_suite_<variant>.log
cd "$(dirname "$0")/.."
G=/usr/local/graft/bin/gpurun
TAG=${1:-r05f}
run2() {  # run2 "<env 1>" <name 1> "<env 2>" <name 2>
  $G --timeout 1200 -- "$1 timeout -k 10 570 python -m pytest tests -x -q -m gpu > gpurun_out/${TAG}_suite_$2.log 2>&1; tail -1 gpurun_out/${TAG}_suite_$2.log; $3 timeout -k 10 570 python -m pytest tests -x -q -m gpu > gpurun_out/${TAG}_suite_$4.log 2>&1; tail -1 gpurun_out/${TAG}_suite_$4.log"
}
run2 "" default "PUNET_CONV_MODE=f16x3" f16x3
run2 "PUNET_RUNTIME=python" python_runtime "PUNET_UPCONV=0" upconv0
run2 "PUNET_SPLITK=0" splitk0 "PUNET_FOLD=0" fold0
