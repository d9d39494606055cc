#!/bin/bash
# A/B of one environment knob on ONE box (box-to-box variance is 3-5 %): tools/ab_env.sh KNOB=VALUE [steps]
# alternates bench.py with and without the knob three times and prints the ms_per_step of each run.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
knob="$1"; steps="${2:-60}"
mkdir -p gpurun_out
for r in 1 2 3; do
  for mode in base knob; do
    if [ "$mode" = knob ]; then export "$knob"; else unset "${knob%%=*}"; fi
    python bench.py --steps "$steps" --warmup 15 --no-sustained 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$mode', '$r', d['ms_per_step'], d['value'])"
  done
done | tee gpurun_out/ab_env.txt
