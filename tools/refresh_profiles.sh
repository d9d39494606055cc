#!/bin/bash
# Runs on the GPU box (gpurun): PMC traffic passes, the rocprofv3 kernel-stats run and the bench line of the
# current tree, all into gpurun_out/refresh/ (copy what should be judged into profiles/).
set -x
R=/root/repo; O=$R/gpurun_out/refresh; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/write.log 2>&1
python $R/tools/pmc_traffic.py $O/fetch $O/write $R/profiles/pmc_traffic.json | tee $O/traffic.log
cp $R/profiles/pmc_traffic.json $O/pmc_traffic.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/stats.log
cd $R; python bench.py > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.json
