#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; O=$PWD/gpurun_out/nms; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
cat > /tmp/one.py <<'PY'
import sys, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo/oracle")
import frcnn_amd as F
from util import random_boxes
n = int(sys.argv[1]); thr = float(sys.argv[2])
b = random_boxes(np.random.RandomState(n), n); db = F.DeviceTensor.from_numpy(b)
for _ in range(12): p = F.nms(db, thr, None)
torch.cuda.synchronize()
import pyoracle as O
print("ids identical", list(p) == O.nms(b, thr).tolist(), len(p))
PY
for lib in base; do
if [ $lib = base ]; then unset FRCNN_LIB_PATH; else export FRCNN_LIB_PATH=$R/faster-rcnn.torch_amd/build/alt/libfrcnn_$lib.so; fi
for cfg in "26544 0.25" "8000 0.25" "8000 0.1" "1400 0.1" "300 0.25"; do
  set -- $cfg
  rm -rf /tmp/prof; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python /tmp/one.py $1 $2 2>/dev/null | grep identical | tr '\n' ' '
  echo -n "== $lib n=$1 thr=$2: "; python - <<'PY'
import csv, glob
f = glob.glob("/tmp/prof/*/*kernel_stats.csv")[0]
print("  ".join("%s %.1f" % (r["Name"].split("(")[0].replace("frcnn::nms_","").replace("_kernel",""), float(r["AverageNs"]) / 1e3) for r in csv.DictReader(open(f)) if "nms" in r["Name"]))
PY
done; done | tee $O/kernels3.txt
cd $R; unset FRCNN_LIB_PATH
python -m pytest tests/test_gpu_nms.py tests/test_gpu_detect_glue.py tests/test_gpu_model.py -x -q 2>&1 | tail -2
python tools/bench_detect.py 2>&1 | grep -v amdgpu.ids | tail -2
