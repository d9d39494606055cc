#!/bin/bash
# like ab_env.sh for several knobs against one baseline: tools/ab_env2.sh steps KNOB=V [KNOB=V ...]
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
steps="$1"; shift
run() { python bench.py --steps "$steps" --warmup 15 --no-cpu-baseline --no-upload-leg --no-sustained 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['ms_per_step'], d['value'])"; }
for r in 1 2 3; do
  echo -n "base $r "; run
  for k in "$@"; do echo -n "$k $r "; env $k bash -c "$(declare -f run); steps=$steps; run"; done
done | tee gpurun_out/ab_env2.txt
