#!/bin/bash
# where the first-layer weight-gradient kernel's time goes: FRCNN_WG_DBG bits 2 = no MFMA phase, 4 = no LDS writes, 8 = loads dead
cd "${GRAFT_REPO_ROOT:-.}"
for d in 0 2 4 8 14; do echo -n "dbg=$d  "; FRCNN_WG_DBG=$d python tools/bench_conv.py wgrad b1c1 2>&1 | tail -1; done
for sp in 256 512; do echo -n "split=$sp  "; FRCNN_W1_SPLIT=$sp python tools/bench_conv.py wgrad b1c1 2>&1 | tail -1; done
for d in 0 2 4 8 14; do echo "fused dbg=$d  "; FRCNN_WG_DBG=$d bash tools/ktrace.sh wgrad -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-sustained 2>&1 | grep "first_kernel<3, true\|reduce"; done
