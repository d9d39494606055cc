#!/bin/bash
# Runs on the GPU box: rocprofv3 --kernel-trace of one command, then the average duration per kernel (us).
# usage: bash tools/ktrace.sh [kernel-name-substring] -- <command...>
PAT=""; if [ "$1" != "--" ]; then PAT=$1; shift; fi; shift
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$(mktemp -d /tmp/ktrace.XXXX); cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O -- "$@" > $O/log.txt 2>&1
grep -v "^\[\|^W2\|^E2\|rocprofv3" $O/log.txt | tail -20
python - "$O" "$PAT" <<'PY'
import collections, csv, glob, sys
O, PAT = sys.argv[1], sys.argv[2]
acc = collections.OrderedDict()
for f in glob.glob(O + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if PAT in n:
            k = (n[:90], r.get("Grid_Size_X", r.get("Grid_Size", "")))
            acc.setdefault(k, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for (n, g), v in acc.items():
    v2 = sorted(v)[: max(1, len(v) - 1)] if len(v) > 2 else v   # (drop the slowest: first launch)
    print("%-92s grid %-8s n=%3d  avg %8.1f us  min %8.1f" % (n, g, len(v), sum(v2) / len(v2), min(v)))
PY
