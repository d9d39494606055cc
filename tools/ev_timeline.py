"""Un-profiled GPU timeline of one training step from the library's own HIP-event brackets (FRCNN_PROF_DUMP): class, stream,
start offset and duration of every launch -- no rocprofv3 in the process, so the host keeps its real lead over the device.
usage: python tools/ev_timeline.py [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dump = "/tmp/frcnn_prof_dump.txt"
os.environ["FRCNN_PROF_DUMP"] = dump
if os.path.exists(dump): os.unlink(dump)
import ctypes as C
import torch, frcnn_amd as F
cfg = dict(F.duplo_cfg); model = F.vgg_small(cfg)
w, g = F.combine_and_flatten_parameters(model["pnet"], model["cnet"])
it = F.SyntheticBatchIterator(model, pool=4)
stats = dict(pcls=[], preg=[], dcls=[], dreg=[])
f = F.create_objective(model, w, g, it, stats); st = dict(learningRate=1e-4, alpha=0.9)
for _ in range(8): F.rmsprop(f, w, st)
torch.cuda.synchronize()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
mask = int(os.environ.get("EV_MASK", "0x1FFF"), 0)
F._lib.call("frcnn_prof_enable", mask)
for _ in range(n): F.rmsprop(f, w, st)
torch.cuda.synchronize()
F._lib.call("frcnn_prof_enable", 0)
nk = len(F._lib.KC_NAMES)
a = (C.c_longlong * nk)(); b = (C.c_double * nk)(); c = (C.c_double * nk)(); d = (C.c_double * nk)()
F._lib.call("frcnn_prof_collect", a, b, c, d)
rows = [l.split() for l in open(dump)]
rows = [(int(r[0]), r[1], float(r[2]), float(r[3])) for r in rows if int(r[0]) >= 0]
rows.sort(key=lambda r: r[2])
opt = [i for i, r in enumerate(rows) if r[0] == 9]     # rmsprop marks the step boundaries
lo, hi = opt[-2], opt[-1]
step = rows[lo + 1:hi + 1]
t0 = step[0][2]
streams = {}
last_end = {}
print("step span %.1f us, %d bracketed launches" % (step[-1][2] + step[-1][3] - t0, len(step)))
for k, s, t, dur in step:
    q = streams.setdefault(s, len(streams) + 1)
    gap = t - last_end.get(q, t)
    last_end[q] = t + dur
    print("q%d +%8.1f us  dur %7.1f  gap %6.1f  %s" % (q, t - t0, dur, gap, F._lib.KC_NAMES[k]))
