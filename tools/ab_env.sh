#!/bin/bash
# A/B of one environment setting on ONE box: tools/ab_env.sh "NAME=value [NAME2=value2 ...]" [steps] [rounds]
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
setting="$1"; steps="${2:-60}"; rounds="${3:-3}"
mkdir -p gpurun_out
for r in $(seq 1 $rounds); do
  for mode in base with; do
    if [ "$mode" = with ]; then pre="env $setting"; else pre=""; fi
    $pre python bench.py --steps "$steps" --warmup 15 --no-sustained --no-cpu-baseline --no-upload-leg 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$mode', '$r', d['ms_per_step'], d['value'])"
  done
done | tee -a gpurun_out/ab_env.txt
