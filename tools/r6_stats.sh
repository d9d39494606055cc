#!/bin/bash
# round 6: kernel stats of the bench command under rocprofv3 (summary only), into gpurun_out/<tag>/
TAG=${1:-stats}; R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$TAG; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-upload-leg --no-sustained --no-other-legs > $O/bench_under_rocprof.json 2> $O/stats.log
cd $R; cp $O/stats/*/*kernel_stats.csv $O/kernel_stats.csv; python tools/timeline.py $O/stats 12 > $O/step_timeline.txt 2>&1; rm -rf $O/stats
python - "$O/kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
steps = 35.0
print("total kernel time per step %.3f ms" % (tot / 1e6 / steps))
for r in rows[:60]:
    print("%-96s n/step %5.1f avg %7.1f us  per-step %7.1f us" % (r['Name'][:96], int(r['Calls']) / steps, float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e3 / steps))
PY
