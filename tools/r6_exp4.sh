#!/bin/bash
# round 6, GPU experiment 4: conv_wgradx with ONE prefetch register set (-DWX_ONE_SET: 320 registers instead of 368, so that a
# conv_x3 block of the input-gradient stream fits beside a weight-gradient block on one compute unit) against the shipped kernel.
# usage: tools/r6_exp4.sh <variant name under build/alt>
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/exp4; mkdir -p $O
A=$PWD/faster-rcnn.torch_amd/build/alt
V=${1:-oneset}
echo "== wgrad per layer" | tee $O/wgrad.txt
for v in base $V; do
  if [ $v = base ]; then unset FRCNN_LIB_PATH; else export FRCNN_LIB_PATH=$A/libfrcnn_$v.so; fi
  echo "-- $v" | tee -a $O/wgrad.txt
  python tools/bench_conv.py wgrad 2>/dev/null | tee -a $O/wgrad.txt
done
echo "== correctness of the variant" | tee $O/test.txt
FRCNN_LIB_PATH=$A/libfrcnn_$V.so python -m pytest tests/test_gpu_convx.py -q -k "weight_gradient" 2>&1 | tail -2 | tee -a $O/test.txt
echo "== A/B step time" | tee $O/ab.txt
for r in 1 2 3; do
  for v in base $V; do
    if [ $v = base ]; then unset FRCNN_LIB_PATH; else export FRCNN_LIB_PATH=$A/libfrcnn_$v.so; fi
    python bench.py --steps 60 --warmup 15 --no-sustained --no-cpu-baseline --no-upload-leg 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); k=d['config']['kernel_classes']; print('$v', '$r', d['ms_per_step'], d['value'], 'x3 live', d['roofline']['frac'], 'x3 ms', k['conv_x3']['ms_per_step'], 'wgx ms', k['conv_wgradx']['ms_per_step'])"
  done
done | tee -a $O/ab.txt
