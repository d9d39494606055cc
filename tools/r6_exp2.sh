#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/exp2; mkdir -p $O
python -m pytest tests/test_gpu_convx.py -q -s -m gpu 2>&1 | grep -E "rms ratio|max error|passed|failed|FAILED" | sort | uniq -c | sort -k2 | tail -80 > $O/convx_ratios.txt
python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_convx.py 2>&1 | tail -15 > $O/pytest_gpu.txt
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/bench.err
tail -3 $O/pytest_gpu.txt; tail -3 $O/bench.err
