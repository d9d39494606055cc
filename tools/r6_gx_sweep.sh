#!/bin/bash
# round 6: tile height / K split of the classification net's large products at R = 560 against gx_plan's choice
cd "${GRAFT_REPO_ROOT:-/root/repo}"
echo "== plan"; ONLY_R=560 python tools/bench_gemm.py 2>/dev/null
for tm in 64 128 192; do for sk in 1 2 3 4 6 8 12 16 24; do
  echo "== TM=$tm SPLITK=$sk"; FRCNN_GX_TM=$tm FRCNN_GX_SPLITK=$sk ONLY_R=560 python tools/bench_gemm.py 2>/dev/null | grep -E "fwd|dgrad"
done; done
