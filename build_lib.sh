#!/bin/bash
# Builds faster-rcnn.torch_amd/libfrcnn_hip.so for gfx950 (hipcc cross-compiles without a GPU).
cd "$(dirname "$0")/faster-rcnn.torch_amd/csrc" || exit 1
OUT=../libfrcnn_hip.so
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result -Wno-pass-failed"
mkdir -p ../build
pids=(); names=()
for f in conv.hip elem.hip gemm.hip roi.hip rpn.hip nms.hip cnet.hip image.hip convx.hip wgradx.hip gemmx.hip detect.hip api.cpp net.cpp anchors.cpp comm.cpp; do
  o=../build/${f%.*}.o
  if [ "$FORCE" = "1" ] || [ ! -f $o ] || [ $f -nt $o ] || [ kernels.h -nt $o ] || [ common.h -nt $o ] || [ ../../include/frcnn_hip.h -nt $o ]; then
    case $f in *.cpp) X="-x hip";; *) X="";; esac
    hipcc $FLAGS $X -c $f -o $o &
    pids+=($!); names+=($f)
  fi
done
fail=0
for i in "${!pids[@]}"; do
  if ! wait ${pids[$i]}; then echo "build_lib.sh: error: compiling ${names[$i]} failed" >&2; fail=1; fi
done
[ $fail = 0 ] || exit 1
hipcc --offload-arch=gfx950 -shared -fPIC ../build/*.o -o $OUT || { echo "build_lib.sh: error: link failed" >&2; exit 1; }
echo "built $OUT"
