#!/bin/bash
# round 6: the update beside the backward pass (state["eager"], FRCNN_EAGER_UPDATE=0 switches it off): tests, A/B of the step, phases
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/eager; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_eager.py -x -q 2>&1 | tail -15 | tee $O/test.txt
for r in 1 2 3; do
for e in 1 0; do
  echo -n "FRCNN_EAGER_UPDATE=$e run $r: "
  FRCNN_EAGER_UPDATE=$e python bench.py --steps 80 --warmup 20 --no-sustained --no-cpu-baseline --no-upload-leg --no-other-legs 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['ms_per_step'], d['value'], d['host'] if 'host' in d else '')"
done; done | tee $O/ab.txt
for e in 1 0; do echo "== FRCNN_EAGER_UPDATE=$e"; FRCNN_EAGER_UPDATE=$e python tools/phase_probe.py 40 2>&1 | grep -v amdgpu; done | tee $O/phases.txt
