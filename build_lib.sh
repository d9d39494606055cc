#!/bin/bash
# Builds faster-rcnn.torch_amd/libfrcnn_hip.so for gfx950 (hipcc cross-compiles without a GPU).
set -e
cd "$(dirname "$0")/faster-rcnn.torch_amd/csrc"
OUT=../libfrcnn_hip.so
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result"
mkdir -p ../build
pids=()
for f in conv.hip elem.hip gemm.hip roi.hip rpn.hip nms.hip cnet.hip; do
  if [ ! -f ../build/${f%.hip}.o ] || [ $f -nt ../build/${f%.hip}.o ] || [ kernels.h -nt ../build/${f%.hip}.o ] || [ common.h -nt ../build/${f%.hip}.o ] || [ ../../include/frcnn_hip.h -nt ../build/${f%.hip}.o ]; then
    hipcc $FLAGS -c $f -o ../build/${f%.hip}.o &
    pids+=($!)
  fi
done
for f in api.cpp net.cpp; do
  hipcc $FLAGS -x hip -c $f -o ../build/${f%.cpp}.o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC ../build/*.o -o $OUT
echo "built $OUT"
