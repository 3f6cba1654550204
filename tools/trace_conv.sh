#!/bin/bash
# Build conv_x6.hip with its s_memtime stamps and standalone driver (-DX6B_TRACE -DX6B_TRACE_MAIN) against the other objects
# of the library, into tools/trace_conv (untracked).  Run the binary on the GPU box: one workgroup's chunks 8..11, per wave.
set -e
cd "$(dirname "$0")/.."
C=probabilistic_domain_adaptation_amd/csrc
objs=""; for f in capi conv_mfma wgrad_x6 wgrad_mfma conv_small pointwise fcomb_sampler fcomb_train augment; do objs="$objs $C/build/$f.o"; done
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -DX6B_TRACE -DX6B_TRACE_MAIN $TRACE_FLAGS -I $C -I include -c $C/conv_x6.hip -o /tmp/trace_conv.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 /tmp/trace_conv.o $objs -o tools/trace_conv
