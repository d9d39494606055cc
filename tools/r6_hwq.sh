#!/bin/bash
# round 6: does the number of hardware queues the runtime multiplexes the library's streams onto matter?  (GPU_MAX_HW_QUEUES, default 4)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/hwq; mkdir -p $O
for r in 1 2; do
for q in 4 3 5 6; do
  echo -n "GPU_MAX_HW_QUEUES=$q run $r: "
  GPU_MAX_HW_QUEUES=$q python bench.py --steps 80 --warmup 20 --no-sustained --no-cpu-baseline --no-upload-leg --no-other-legs 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['ms_per_step'], d['value'])"
done; done | tee $O/hwq2.txt
python tools/phase_probe.py 40 2>&1 | grep -v amdgpu | tee $O/phases_join.txt
for e in FRCNN_HEAD_STREAMS=0 FRCNN_CNET_WGRAD_ASYNC=0; do echo "== $e"; env $e python tools/phase_probe.py 40 2>&1 | grep -v amdgpu; done | tee $O/phases_opts.txt
