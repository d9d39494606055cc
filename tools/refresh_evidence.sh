#!/bin/bash
# Runs on the GPU box after tools/refresh_profiles.sh would (same passes + the MFMA-busy pass): regenerates
# gpurun_out/evid/{r01_kernel_evidence.csv,r01_pmc_fetch_size.csv,r01_pmc_write_size.csv} for profiles/.
set -x
R=/root/repo; O=$R/gpurun_out/evid; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/mfma -- $B > $O/mfma.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- $B > $O/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- $B > $O/write.log 2>&1
cd $R
python tools/pmc_summary.py $O/mfma $O/fetch $O/write $O/r01_kernel_evidence.csv | tail -5
python - <<'PY'
import collections, csv, glob
for name, ctr in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(glob.glob("/root/repo/gpurun_out/evid/%s/*/*counter_collection.csv" % name)[0])):
        if r["Counter_Name"] == ctr:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    rows = sorted(acc.items(), key=lambda kv: -sum(kv[1]))
    with open("/root/repo/gpurun_out/evid/r01_pmc_%s_size.csv" % name, "w") as f:
        w = csv.writer(f); w.writerow(["kernel", "dispatches", "mean_%s_KB" % ctr, "total_KB"])
        for k, v in rows:
            w.writerow([k, len(v), round(sum(v) / len(v), 1), round(sum(v), 1)])
PY
ls -la $O/*.csv
