#!/bin/bash
# round 6, GPU experiment 1: the bare stage-loop barriers (convx.hip CX_BARRIER) against the __syncthreads() form, the staging
# fault's code shape (plain multiplies) under both, -fno-slp-vectorize, and the phases of the step.  Output: gpurun_out/exp1/
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/exp1; mkdir -p $O
A=faster-rcnn.torch_amd/build/alt
echo "== x3_check, the shipped library" | tee $O/x3_check.txt
for r in 1 2 3; do timeout 300 python tools/x3_check.py 2>&1 | grep -v "bad elements 0$" ; done | tee -a $O/x3_check.txt
for v in plainsync2 plain; do
  echo "== x3_check, $v (plain C multiplies: the code shape of the round-5 fault)" | tee -a $O/x3_check.txt
  for r in 1 2 3; do FRCNN_LIB_PATH=$PWD/$A/libfrcnn_$v.so timeout 300 python tools/x3_check.py 2>&1 | grep -v "bad elements 0$"; done | tee -a $O/x3_check.txt
done
echo "== A/B step time (60 steps, 3 rounds): base = bare barriers" | tee $O/ab.txt
for r in 1 2 3; do
  for v in base sync2 noslp; do
    if [ $v = base ]; then unset FRCNN_LIB_PATH; else export FRCNN_LIB_PATH=$PWD/$A/libfrcnn_$v.so; fi
    python bench.py --steps 60 --warmup 15 --no-sustained --no-cpu-baseline --no-upload-leg 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$v', '$r', d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['isolated']['frac'], d['config']['kernel_classes']['conv_wgradx']['ms_per_step'])"
  done
done | tee -a $O/ab.txt
unset FRCNN_LIB_PATH
echo "== per-layer conv kernels" | tee $O/conv.txt
for v in base sync2; do
  if [ $v = base ]; then unset FRCNN_LIB_PATH; else export FRCNN_LIB_PATH=$PWD/$A/libfrcnn_$v.so; fi
  echo "-- $v" | tee -a $O/conv.txt
  python tools/bench_conv.py fwd 2>/dev/null | tee -a $O/conv.txt
  python tools/bench_conv.py dgrad b2c1 b2c2 b3c1 b3c2 b4c1 b4c2 2>/dev/null | tee -a $O/conv.txt
done
unset FRCNN_LIB_PATH
python tools/phase_probe.py 40 2>&1 | tee $O/phases.txt
python -m pytest tests/test_gpu_convx.py tests/test_gpu_fullsize.py -x -q -s 2>&1 | grep -E "rms ratio|passed|failed|Error|error|assert" | sort | uniq -c | sort -rn | head -60 | tee $O/pytest_convx.txt
