#!/bin/bash
# build the library here (gfx950 cross-compile), then run the given command on a GPU box: tools/gpu.sh [--timeout N] -- '<command>'
set -e
cd "$(dirname "$0")/.."
python __graft_entry__.py >/dev/null
exec /usr/local/graft/bin/gpurun "$@"
