#!/bin/bash
# A/B of two builds of the library on ONE box: tools/ab_lib.sh <other.so> [steps] [bench_conv args...]
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
other="$1"; steps="${2:-60}"
for r in 1 2 3; do
  for mode in base other; do
    if [ "$mode" = other ]; then export FRCNN_LIB_PATH="$PWD/$other"; else unset FRCNN_LIB_PATH; fi
    python bench.py --steps "$steps" --warmup 15 --no-sustained 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$mode', '$r', d['ms_per_step'], d['value'])"
  done
done | tee gpurun_out/ab_lib.txt
