#!/bin/bash
# Runs on the GPU box: per-kernel PMC averages of one command, one rocprofv3 pass per counter group (kernel-trace only).
# usage: bash tools/pmc_kernels.sh <out-dir> <kernel-name-substring> -- <command...>
O=$(realpath -m $1); PAT=$2; shift 3
R=${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
i=0
for grp in "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS" \
           "FETCH_SIZE" "WRITE_SIZE" \
           "TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum TCC_EA0_RDREQ_sum" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU"; do
  i=$((i+1))
  timeout 120 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/p$i -- "$@" > $O/p$i.log 2>&1 || echo "pass $i ($grp): rc=$?"
done
cd $R
python - "$O" "$PAT" <<'PY'
import collections, csv, glob, sys
O, PAT = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(O + "/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if PAT in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        print("   %-34s n=%3d mean %.4g" % (c, len(v), sum(v) / len(v)))
PY
