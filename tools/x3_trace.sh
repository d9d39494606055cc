#!/bin/bash
# Per-block phase timestamps of one conv_x3 launch (a -DCX_TRACE build of convx.hip next to the normal library):
#   tools/x3_trace.sh <fwd|dgrad> <layer> [env assignments...]   ->  gpurun_out/x3_trace_<kind>_<layer>.txt + summary
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
kind="$1"; layer="$2"; shift 2
lib=faster-rcnn.torch_amd/build/alt/libfrcnn_trace.so
if [ ! -f $lib ] || [ faster-rcnn.torch_amd/csrc/convx.hip -nt $lib ]; then
  (cd faster-rcnn.torch_amd/csrc && mkdir -p ../build/alt && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result -Wno-pass-failed \
     -DCX_TRACE -c convx.hip -o ../build/alt/convx_trace.o && \
   hipcc --offload-arch=gfx950 -shared -fPIC $(ls ../build/*.o | grep -v -e convx.o) ../build/alt/convx_trace.o -o ../build/alt/libfrcnn_trace.so) || exit 1
fi
mkdir -p gpurun_out
out=gpurun_out/x3_trace_${kind}_${layer}.txt
env "$@" FRCNN_LIB_PATH=$PWD/$lib FRCNN_X3_TRACE=$PWD/$out python tools/bench_conv.py $kind $layer
python tools/x3_trace.py $out
