#!/bin/bash
# A/B of conv_x3 launch variants on ONE box: tools/x3_ab.sh "<env assignments>" ... (each argument is one configuration)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
layers="${X3_LAYERS:-b2c1 b2c2 b3c1 b3c2 b4c1 b4c2}"
for cfg in "$@"; do
  echo "=== $cfg"
  for kind in fwd dgrad; do
    env $cfg python tools/bench_conv.py $kind $layers 2>&1 | grep -v "^$"
  done
done
