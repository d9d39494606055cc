#!/bin/bash
# round 6: A/B of environment switches on one box.  usage: tools/r6_ab.sh "ENV1=a ENV2=b" "ENV1=c" ...   (each argument one configuration)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/ab; mkdir -p $O
R=${ROUNDS:-2}
for r in $(seq $R); do
for cfg in "$@"; do
  echo -n "[$cfg] run $r: "
  env $cfg python bench.py --steps ${STEPS:-80} --warmup 20 --no-sustained --no-cpu-baseline --no-upload-leg --no-other-legs 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['ms_per_step'], d['value'], 'x3', d['roofline']['frac'], 'enqueue', d['host']['enqueue_ms_per_step'])"
done; done | tee -a $O/ab.txt
if [ -n "$PHASES" ]; then for cfg in "$@"; do echo "== [$cfg]"; env $cfg python tools/phase_probe.py 40 2>&1 | grep -v amdgpu; done | tee -a $O/phases.txt; fi
