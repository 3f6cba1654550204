#!/bin/bash
# Per-layer table of tools/bench_x6_layers.py for several library variants (tools/ab_variants.sh), one after the other on one box.
# usage: tools/ab_layers_detail.sh name1 name2 ...   ('base' = libpunet_hip.so)
cd "$(dirname "$0")/.."
for n in "$@"; do
  lib=probabilistic_domain_adaptation_amd/libpunet_hip.so
  [ "$n" != base ] && lib=probabilistic_domain_adaptation_amd/libpunet_hip_$n.so
  echo "== $n"
  PUNET_HIP_LIB=$PWD/$lib python tools/bench_x6_layers.py --rounds 3 2>/dev/null | grep -v amdgpu.ids
done
