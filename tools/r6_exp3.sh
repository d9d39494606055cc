#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/exp3; mkdir -p $O
for e in "FRCNN_X3_F16=1" "FRCNN_X3_F16=0" "FRCNN_SPLIT_BF16=0"; do
  echo -n "$e: "; env $e python bench.py --no-cpu-baseline --no-sustained --no-upload-leg --steps 40 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], 'images/s', d['ms_per_step'], 'ms/step', d['roofline']['frac'])"
done | tee $O/env.txt
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee $O/legs.txt
import bench, sys
import frcnn_amd as F
F._lib.load()
for name, opts in (("control", {}), ("exact_split", dict(x3_f16=0)), ("control2", {}), ("fp32_mfma", dict(split_bf16=0)), ("control3", {})):
    r = bench.arithmetic_leg(F, name, opts)
    print(name, r["value"], r["ms_per_step"], r["roofline"]["frac"])
PY
