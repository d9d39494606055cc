#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/nms; mkdir -p $O
python -m pytest tests/test_gpu_nms.py tests/test_gpu_detect_glue.py -x -q 2>&1 | tail -3 | tee $O/pytest.txt
python tests/perf_nms.py 2>&1 | grep -v amdgpu.ids | tee $O/perf.txt
python tools/bench_detect.py 2>&1 | grep -v amdgpu.ids | tail -15 | tee $O/detect.txt
