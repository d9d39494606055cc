#!/bin/bash
# Critical-path sensitivity: build the library with -DFR_SKIPTEST (tools only), then drop one kernel family at a time
# (results are garbage, timing is not) and report the step time.  Runs on the GPU box.
set -e
cd /root/repo/faster-rcnn.torch_amd/csrc
mkdir -p /tmp/sk
for f in conv elem gemm roi rpn nms cnet; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result -Wno-pass-failed -DFR_SKIPTEST -c $f.hip -o /tmp/sk/$f.o & done
for f in api net; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -DFR_SKIPTEST -c $f.cpp -o /tmp/sk/$f.o & done
wait
cp ../libfrcnn_hip.so /tmp/sk/orig.so
hipcc --offload-arch=gfx950 -shared -fPIC /tmp/sk/*.o -o ../libfrcnn_hip.so
cd /root/repo
for k in NONE roi_pool_forward roi_pool_backward splitk_reduce wgrad_reduce maxpool_act_forward "act_backward_kernel<true" "act_backward_kernel<false" pack_weights rmsprop gemm_reduce channel_sum bn_ conv_wgrad_first gather_positions col2im im2col gemm_dma gemm_kernel "conv_igemm_kernel<1" "conv_igemm_kernel<7" "conv_igemm_kernel<5" bernoulli prelu_dropout log_softmax cnet_losses rpn_loss; do
  v=$(FRCNN_EXP_SKIP="$k" python bench.py --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")
  echo "skip $k : $v ms/step"
done
cp /tmp/sk/orig.so faster-rcnn.torch_amd/libfrcnn_hip.so
