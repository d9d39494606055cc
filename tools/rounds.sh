#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$(mktemp -d /tmp/rounds.XXXX); cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-upload-leg --no-sustained > $O/log.txt 2>&1
python $R/tools/rounds.py $O > $R/gpurun_out/rounds.txt 2>&1; head -3 $R/gpurun_out/rounds.txt
