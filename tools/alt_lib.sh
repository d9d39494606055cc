#!/bin/bash
# A second build of the library with one source compiled differently (same-box A/B through FRCNN_LIB_PATH, tools/ab_lib.sh):
#   tools/alt_lib.sh <name> <source.hip> <compiler flags...>   ->  faster-rcnn.torch_amd/build/alt/libfrcnn_<name>.so
# (runs here: hipcc cross-compiles; the .so travels to the GPU box with the tree)
cd "$(dirname "$0")/../faster-rcnn.torch_amd/csrc" || exit 1
name="$1"; src="$2"; shift 2
base="${src%.hip}"
mkdir -p ../build/alt   # (build_lib.sh links ../build/*.o: the alternates live one level down)
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result -Wno-pass-failed "$@" -c "$src" -o "../build/alt/${base}_${name}.o" || exit 1
others=$(ls ../build/*.o | grep -v -e "/${base}.o")
hipcc --offload-arch=gfx950 -shared -fPIC $others "../build/alt/${base}_${name}.o" -o "../build/alt/libfrcnn_${name}.so" && echo "built build/alt/libfrcnn_${name}.so"
