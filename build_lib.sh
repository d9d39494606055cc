#!/bin/bash
# Builds faster-rcnn.torch_amd/libfrcnn_hip.so for gfx950 (hipcc cross-compiles without a GPU).
cd "$(dirname "$0")/faster-rcnn.torch_amd/csrc" || exit 1
OUT=../libfrcnn_hip.so
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result -Wno-pass-failed"
# convx.hip / wgradx.hip: no SLP vectoriser (packed fp32 VALU arithmetic in the operand staging is fenced off: tools/x3_isa_check.py,
# DESIGN.md section 4 "staging fault"), and the compiler's own device assembly kept beside the object for that check
X3FLAGS="-fno-slp-vectorize -save-temps=obj"
mkdir -p ../build
pids=(); names=()
for f in conv.hip elem.hip gemm.hip roi.hip rpn.hip nms.hip cnet.hip image.hip convx.hip wgradx.hip gemmx.hip detect.hip heads.hip api.cpp net.cpp anchors.cpp comm.cpp; do
  o=../build/${f%.*}.o
  if [ "$FORCE" = "1" ] || [ ! -f $o ] || [ $f -nt $o ] || [ kernels.h -nt $o ] || [ common.h -nt $o ] || [ amax.h -nt $o ] || [ ../../include/frcnn_hip.h -nt $o ]; then
    case $f in *.cpp) X="-x hip";; convx.hip|wgradx.hip) X="$X3FLAGS";; *) X="";; esac
    hipcc $FLAGS $X -c $f -o $o 2> >(grep -v "argument unused during compilation" >&2) &
    pids+=($!); names+=($f)
  fi
done
fail=0
for i in "${!pids[@]}"; do
  if ! wait ${pids[$i]}; then echo "build_lib.sh: error: compiling ${names[$i]} failed" >&2; fail=1; fi
done
[ $fail = 0 ] || exit 1
# keep the device assembly of the two guarded sources (build/isa/), drop the other -save-temps files
mkdir -p ../build/isa
for b in convx wgradx; do
  [ -f ../build/$b-hip-amdgcn-amd-amdhsa-gfx950.s ] && mv ../build/$b-hip-amdgcn-amd-amdhsa-gfx950.s ../build/isa/$b.s
  rm -f ../build/$b-hip-amdgcn-amd-amdhsa-gfx950.* ../build/$b-host-x86_64-unknown-linux-gnu.* ../build/$b.hip-hip-amdgcn-amd-amdhsa.hipfb
done
python3 ../../tools/x3_isa_check.py ../build/isa/convx.s ../build/isa/wgradx.s \
  || { echo "build_lib.sh: error: the split kernels' ISA check failed" >&2; exit 1; }
OBJS=""; for f in conv elem gemm roi rpn nms cnet image convx wgradx gemmx detect heads api net anchors comm; do OBJS="$OBJS ../build/$f.o"; done
hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o $OUT || { echo "build_lib.sh: error: link failed" >&2; exit 1; }
echo "built $OUT"
