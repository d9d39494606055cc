import sys, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import frcnn_amd as F
F._lib.load()
import test_gpu_dropcompact as T
H, W = 225, 400
la, ga, ka, model = T._step(F, True, H, W, True, 77)
lb, gb, kb, _ = T._step(F, False, H, W, True, 77)
lc, gc, kc, _ = T._step(F, False, H, W, True, 77)
print("loss", la, lb, lc)
nat = model["native"]
for off, cnt, kind, aux in nat.param_table:
    a, b, c = (x[off:off + cnt].astype(np.float64) for x in (ga, gb, gc))
    nb = max(np.linalg.norm(b), 1e-30)
    print("off %9d n %8d kind %d  compact-vs-dense %.2e   dense-vs-dense %.2e   |g| %.3e" % (off, cnt, kind, np.linalg.norm(a - b) / nb, np.linalg.norm(c - b) / nb, nb))
