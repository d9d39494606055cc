"""Host-side enqueue times of one training step (un-profiled): when each C-ABI call is issued relative to the
start of the step, next to the step's wall time.  Shows whether the host stays ahead of the device."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, frcnn_amd as F
from frcnn_amd import _lib
cfg = dict(F.duplo_cfg); model = F.vgg_small(cfg)
w, g = F.combine_and_flatten_parameters(model["pnet"], model["cnet"])
it = F.SyntheticBatchIterator(model, pool=4)
stats = dict(pcls=[], preg=[], dcls=[], dreg=[])
f = F.create_objective(model, w, g, it, stats); st = dict(learningRate=1e-4, alpha=0.9)
for _ in range(5): F.rmsprop(f, w, st)
torch.cuda.synchronize()
log = []
orig = _lib.call
def call(name, *a):
    t = time.perf_counter(); r = orig(name, *a); log.append((name, t, time.perf_counter())); return r
_lib.call = call
for m in (F.objective, F.model_utilities, F.utilities):
    if hasattr(m, "_lib"): m._lib.call = call
marks = []
for i in range(8):
    marks.append((len(log), time.perf_counter()))
    F.rmsprop(f, w, st)
torch.cuda.synchronize(); tend = time.perf_counter()
print("ms/step %.3f" % ((tend - marks[0][1]) / 8 * 1e3))
k, t0 = marks[5]; k1, t1 = marks[6]
print("step 5: host span %.0f us" % ((t1 - t0) * 1e6))
for name, a, b in log[k:k1]:
    if name in ("frcnn_pnet_set_sparse_deltas", "frcnn_pnet_output", "frcnn_pnet_delta", "frcnn_get_option"): continue
    print("+%7.0f us  %6.0f us  %s" % ((a - t0) * 1e6, (b - a) * 1e6, name))
