#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/eager; mkdir -p $O
for r in 1 2; do
for e in 1 0; do
  echo -n "FRCNN_HEAD_STREAMS=0 FRCNN_EAGER_UPDATE=$e run $r: "
  FRCNN_HEAD_STREAMS=0 FRCNN_EAGER_UPDATE=$e python bench.py --steps 80 --warmup 20 --no-sustained --no-cpu-baseline --no-upload-leg --no-other-legs 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['ms_per_step'], d['value'])"
done; done | tee $O/ab2.txt
for e in 1 0; do echo "== FRCNN_HEAD_STREAMS=0 FRCNN_EAGER_UPDATE=$e"; FRCNN_HEAD_STREAMS=0 FRCNN_EAGER_UPDATE=$e python tools/phase_probe.py 40 2>&1 | grep -v amdgpu; done | tee $O/phases2.txt
