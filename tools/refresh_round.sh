#!/bin/bash
# Runs on the GPU box (gpurun): every measurement that profiles/ holds for one round, from the current tree, into
# gpurun_out/<tag>/ (copy what should be judged into profiles/).  usage: bash tools/refresh_round.sh r03
TAG=${1:-r04}
export FRCNN_GIT_HASH=${2:-${FRCNN_GIT_HASH:-unknown}}   # the tree the measurements belong to (the GPU box has no .git): pass `git rev-parse --short HEAD`
R=/root/repo; O=$R/gpurun_out/$TAG; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-upload-leg --no-sustained"
# 1. PMC passes (separate runs, --kernel-trace only, as MI355X_MICROARCH.md prescribes)
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- $B > $O/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- $B > $O/write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/mfma -- $B > $O/mfma.log 2>&1
cd $R
python tools/pmc_traffic.py $O/fetch $O/write $O/pmc_traffic.json | tee $O/traffic.log
python tools/pmc_summary.py $O/mfma $O/fetch $O/write $O/${TAG}_kernel_evidence.csv > $O/evidence.log 2>&1
python - "$O" "$TAG" <<'PY'
import collections, csv, glob, sys
O, TAG = sys.argv[1], sys.argv[2]
for name, ctr in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(glob.glob("%s/%s/*/*counter_collection.csv" % (O, name))[0])):
        if r["Counter_Name"] == ctr:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    rows = sorted(acc.items(), key=lambda kv: -sum(kv[1]))
    with open("%s/%s_pmc_%s_size.csv" % (O, TAG, name), "w") as f:
        w = csv.writer(f); w.writerow(["kernel", "dispatches", "mean_%s_KB" % ctr, "total_KB"])
        for k, v in rows:
            w.writerow([k, len(v), round(sum(v) / len(v), 1), round(sum(v), 1)])
PY
if [ -n "$PMC_ONLY" ]; then   # only the traffic table (after a change under csrc/ that leaves the measured kernels alone)
  cp $O/pmc_traffic.json $R/profiles/pmc_traffic.json; cp $O/pmc_traffic.json $R/gpurun_out/pmc_traffic.json
  rm -rf $O/fetch $O/write $O/mfma; exit 0
fi
# 2. kernel stats of the bench command (rocprofv3 --kernel-trace --stats) and the step timeline of the same trace
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --no-cpu-baseline --no-upload-leg --no-sustained > $O/${TAG}_bench_under_rocprof.json 2> $O/stats.log
cd $R
cp $O/stats/*/*kernel_stats.csv $O/${TAG}_bench_kernel_stats.csv 2>/dev/null
python tools/timeline.py $O/stats 12 > $O/${TAG}_step_timeline.txt 2>&1
# 3. the bench line itself (un-profiled), with the CPU baseline, the parity object and the other legs (roofline.traffic is
#    read from profiles/pmc_traffic.json: the table of pass 1 goes there first)
cp $O/pmc_traffic.json $R/profiles/pmc_traffic.json
python bench.py > $O/${TAG}_bench.json 2> $O/bench.err
# 4. the other configurations and options
{
  echo "# MI355X, $TAG: configurations and options beside the bench line (commands as run on the GPU box)"
  echo "# per-layer convolution kernels, vgg_small 800x450 shapes -- python tools/bench_conv.py fwd|dgrad|wgrad"
  python tools/bench_conv.py fwd 2>/dev/null; python tools/bench_conv.py dgrad b2c1 b2c2 b3c1 b3c2 b4c1 b4c2 2>/dev/null; python tools/bench_conv.py wgrad 2>/dev/null
  echo "# the same with the fp32 matrix-core kernels only -- FRCNN_SPLIT_BF16=0 python tools/bench_conv.py fwd|dgrad|wgrad"
  FRCNN_SPLIT_BF16=0 python tools/bench_conv.py fwd 2>/dev/null; FRCNN_SPLIT_BF16=0 python tools/bench_conv.py dgrad b2c1 b2c2 b3c1 b3c2 b4c1 b4c2 2>/dev/null; FRCNN_SPLIT_BF16=0 python tools/bench_conv.py wgrad 2>/dev/null
  echo "# sustained v_mfma_f32_32x32x16_bf16 rate (no memory traffic) -- tools/bin/mfma_peak_bf16"
  [ -x tools/bin/mfma_peak_bf16 ] || { mkdir -p tools/bin; hipcc --offload-arch=gfx950 -O3 tools/mfma_peak_bf16.hip -o tools/bin/mfma_peak_bf16 >/dev/null 2>&1; }
  tools/bin/mfma_peak_bf16 2>/dev/null
  echo "# what an LDS-DMA / register load stream delivers per CU (L2-resident, shared, gathered, HBM) -- tools/bin/lds_dma_probe"
  [ -x tools/bin/lds_dma_probe ] || hipcc --offload-arch=gfx950 -O3 tools/lds_dma_probe.hip -o tools/bin/lds_dma_probe >/dev/null 2>&1
  timeout 60 tools/bin/lds_dma_probe 2>/dev/null
  echo "# training step with options -- python bench.py --no-cpu-baseline --steps 40 (images/s, ms/step, roofline.frac live)"
  for e in "FRCNN_SPLIT_BF16=1" "FRCNN_DROP_COMPACT=0" "FRCNN_SPARSE_HEADS=0" "FRCNN_DROP_COMPACT=0 FRCNN_SPARSE_HEADS=0" "FRCNN_EAGER_UPDATE=1" "FRCNN_X3_F16=0" "FRCNN_SPLIT_BF16=0" "FRCNN_GEMM_X=0" "FRCNN_DETERMINISTIC=1" "FRCNN_SIDE_STREAM=0" "FRCNN_HEAD_STREAMS=0" "FRCNN_CNET_WGRAD_ASYNC=0" "FRCNN_FUSE_ACT=0" "FRCNN_FIRST_POOLED=0"; do
    echo -n "$e: "; env $e python bench.py --no-cpu-baseline --no-sustained --steps 40 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], 'images/s', d['ms_per_step'], 'ms/step', d['roofline']['frac'])"
  done
  echo "# config 5 shapes on one GPU -- python bench.py --model vgg_large --height 600 --width 1000 --steps 10 --no-cpu-baseline"
  for e in "FRCNN_SPLIT_BF16=1" "FRCNN_SPLIT_BF16=0"; do
    echo -n "$e: "; env $e python bench.py --model vgg_large --height 600 --width 1000 --steps 10 --no-cpu-baseline --no-sustained 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], 'images/s', d['ms_per_step'], 'ms/step; conv_igemm 3x3', r['achieved'], 'TFLOP/s live,', r['isolated']['achieved'], 'alone')"
  done
  echo "# the cnet's Linear(13824,1024) in its three roles, split-bf16 form and (FRCNN_GEMM_X=0) fp32 matrix-core kernels -- python tools/bench_gemm.py"
  python tools/bench_gemm.py 2>/dev/null | grep "I=13824"; FRCNN_GEMM_X=0 python tools/bench_gemm.py 2>/dev/null | grep "I=13824"
  echo "# config 2 -- python tools/bench_detect.py 30   (Detector:detect on 3x450x800 frames; CLASSES=200: config/imagenet.lua class count)"
  python tools/bench_detect.py 30 2>/dev/null; CLASSES=200 CLS_GAIN=2000 python tools/bench_detect.py 30 2>/dev/null
} > $O/${TAG}_other_configs.txt 2>&1
# 5. Detector:detect under rocprofv3: kernel stats of the inference leg (BASELINE config 2)
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/detect -- python $R/tools/bench_detect.py 30 > $O/detect.log 2>&1
cd $R
cp $O/detect/*/*kernel_stats.csv $O/${TAG}_detect_kernel_stats.csv 2>/dev/null
# 5b. BASELINE config 5's one-GPU workload (vgg_large 3x600x1000) under rocprofv3: kernel stats of its training step
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/large -- python $R/bench.py --model vgg_large --height 600 --width 1000 --steps 10 --no-cpu-baseline --no-upload-leg --no-sustained > $O/large.log 2>&1
cd $R
cp $O/large/*/*kernel_stats.csv $O/${TAG}_vgg_large_kernel_stats.csv 2>/dev/null
# 6. the weight-gradient kernel by itself: per-launch durations of kernel and fold, and the PMC groups behind the LDS / matrix-pipe figures
{
  echo "# conv_wgradx_kernel + wgrad_reduce4_kernel per layer -- bash tools/ktrace.sh wgrad -- python tools/bench_conv.py wgrad <layers>"
  bash tools/ktrace.sh wgrad -- python $R/tools/bench_conv.py wgrad b2c1 b2c2 b3c1 b3c2 b4c1 b4c2 2>/dev/null
  echo "# PMC groups, b2c2 (128 -> 128 @ 225x400), 256 blocks of 4 waves -- bash tools/pmc_kernels.sh <dir> wgradx -- python tools/bench_conv.py wgrad b2c2"
  bash tools/pmc_kernels.sh $O/pmcw wgradx -- python $R/tools/bench_conv.py wgrad b2c2 2>/dev/null
} > $O/${TAG}_wgradx_evidence.txt 2>&1
# 7. the un-profiled timeline of one step from the library's own event brackets
python tools/ev_timeline.py 3 > $O/${TAG}_ev_timeline.txt 2>/dev/null
rm -rf $O/fetch $O/write $O/mfma $O/stats $O/detect $O/pmcw
ls -la $O | head -40
