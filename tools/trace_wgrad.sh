#!/bin/bash
# Build wgrad_x6.hip with its s_memtime stamps and standalone driver (-DPUNET_WGRAD_TRACE[_MAIN]) against the other
# objects of the library, into tools/trace_wgrad (untracked).  Run the binary on the GPU box: one workgroup's tile, per wave:
# ids 0 loop top, 10 barrier reached, 11 barrier passed (stamps only at the region boundaries: they do not change the schedule).
set -e
cd "$(dirname "$0")/.."
C=probabilistic_domain_adaptation_amd/csrc
objs=""; for f in capi conv_mfma conv_x6 wgrad_mfma conv_small pointwise fcomb_sampler fcomb_train augment; do objs="$objs $C/build/$f.o"; done
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -DPUNET_WGRAD_TRACE -DPUNET_WGRAD_TRACE_MAIN $TRACE_FLAGS -I $C -I include \
    -c $C/wgrad_x6.hip -o /tmp/trace_wgrad.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 /tmp/trace_wgrad.o $objs -o tools/trace_wgrad
