#!/bin/bash
# usage (on the GPU box): bash tools/prof_step.sh <outdir-name>   -> gpurun_out/<name>/ kernel stats of bench.py
out=/root/repo/gpurun_out/$1; mkdir -p $out; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python /root/repo/bench.py --no-cpu-baseline --steps 20 --warmup 3 > $out/bench.log 2>&1
tail -1 $out/bench.log | cut -c1-200
