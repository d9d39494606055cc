for tm in 64 128; do for tn in 64 128; do for sk in 0 4 8; do
 if [ $sk -eq 0 ]; then unset FRCNN_GEMM_SPLITK; else export FRCNN_GEMM_SPLITK=$sk; fi
 echo "TM=$tm TN=$tn SK=$sk"; FRCNN_GEMM_TM=$tm FRCNN_GEMM_TN=$tn python tools/bench_gemm.py 2>&1 | grep "R=560 I=13824\|R=138 I=13824\|R=320 I=13824"
done; done; done
