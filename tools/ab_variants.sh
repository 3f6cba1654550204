#!/bin/bash
# Build variants of libpunet_hip.so that differ in ONE source file's -D flags (kernel experiments), for a same-box A/B:
#   tools/ab_variants.sh wgrad_x6.hip name1 "-DX=1" name2 "-DX=2" ...   (run in the build container; the .so files travel)
# then on the GPU box:  python tools/ab_run.py wgrad name1 name2 ...   (PUNET_HIP_LIB selects the library per process)
set -e
cd "$(dirname "$0")/.."
SRC=$1; shift
PKG=probabilistic_domain_adaptation_amd
TLIB=$(python -c "import torch,os;print(os.path.join(os.path.dirname(torch.__file__),'lib'))")
python __graft_entry__.py > /dev/null
while [ $# -gt 0 ]; do
  NAME=$1; FLAGS=$2; shift 2
  OBJ=$PKG/csrc/build/${SRC%.hip}_$NAME.o
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -I $PKG/csrc -I include $FLAGS -c $PKG/csrc/$SRC -o $OBJ
  OBJS=""
  for f in capi conv_mfma conv_x6 wgrad_mfma wgrad_x6 conv_small pointwise fcomb_sampler fcomb_train augment upconv runtime; do
    if [ "$f.hip" == "$SRC" ]; then OBJS="$OBJS $OBJ"; else OBJS="$OBJS $PKG/csrc/build/$f.o"; fi
  done
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -no-hip-rt -o $PKG/libpunet_hip_$NAME.so $OBJS -L$TLIB -lamdhip64 -Wl,-rpath,$TLIB
  echo "built $PKG/libpunet_hip_$NAME.so ($FLAGS)"
done
