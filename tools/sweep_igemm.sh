for L in a2 a3 a4; do
 for bm in 0 1; do for sk in 1 2 4 8 12 16 24; do
  echo -n "$L bm64=$bm sk=$sk : "; FRCNN_IG_BM64=$bm FRCNN_IG_SPLITK=$sk python tools/bench_conv.py fwd $L | grep fwd
 done; done
done
for L in b2c1 b2c2 b3c1 b4c1 a1 a2 a3 a4; do
 for bm in 0 1; do for sk in 1 2 4 8 16; do
  echo -n "$L bm64=$bm sk=$sk : "; FRCNN_IG_BM64=$bm FRCNN_IG_SPLITK=$sk python tools/bench_conv.py dgrad $L | grep dgrad
 done; done
done
