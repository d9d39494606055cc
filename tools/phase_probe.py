"""Un-profiled phase lengths of the training step on the caller's stream: events recorded right after selected C-ABI calls
(pnet forward | ROI-pooling backward = end of the classification net's chain | anchor nets joined | backbone backward |
update).  usage: python tools/phase_probe.py [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import frcnn_amd as F
MARK = {"frcnn_pnet_forward_async_heads": "forward", "frcnn_roi_pool_forward": "roi_fwd", "frcnn_cnet_forward": "cnet_fwd",
        "frcnn_cnet_backward": "cnet_bwd", "frcnn_roi_pool_backward": "cnet_chain_end",
        "frcnn_pnet_backward_heads_join": "heads_joined", "frcnn_pnet_backward": "backward", "frcnn_rmsprop": "update",
        "frcnn_scale_rmsprop": "update"}
marks = []
JOIN = os.environ.get("PROBE_JOIN", "1") != "0"
orig = F._lib.call
def call(name, *a):
    if name == "frcnn_pnet_backward" and rec[0] and JOIN:   # the wait for the anchor nets' backward chains, by itself
        e = torch.cuda.Event(enable_timing=True); e.record(); marks.append(("pre_join", e))
        import ctypes
        j = ctypes.c_int(0)
        orig("frcnn_pnet_backward_heads_join", a[0], a[3], ctypes.byref(j))
        e = torch.cuda.Event(enable_timing=True); e.record(); marks.append(("heads_joined", e))
    r = orig(name, *a)
    if name in MARK and rec[0]:
        e = torch.cuda.Event(enable_timing=True); e.record(); marks.append((MARK[name], e))
    return r
rec = [False]
F._lib.call = call
for name, m in list(sys.modules.items()):       # the package's modules hold `_lib` by reference: one patch reaches them all
    pass
cfg = dict(F.duplo_cfg); model = F.vgg_small(cfg)
w, g = F.combine_and_flatten_parameters(model["pnet"], model["cnet"])
it = F.SyntheticBatchIterator(model, pool=4)
f = F.create_objective(model, w, g, it, dict(pcls=[], preg=[], dcls=[], dreg=[])); st = dict(learningRate=1e-4, alpha=0.9)
for _ in range(8): F.rmsprop(f, w, st)
torch.cuda.synchronize()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rec[0] = True
for _ in range(n): F.rmsprop(f, w, st)
torch.cuda.synchronize()
import collections
acc = collections.OrderedDict(); cnt = collections.Counter()
for (a, ea), (b, eb) in zip(marks[:-1], marks[1:]):
    k = "%s -> %s" % (a, b)
    acc[k] = acc.get(k, 0.0) + ea.elapsed_time(eb); cnt[k] += 1
tot = 0.0
for k, v in acc.items():
    print("%-34s %8.1f us  (n=%d)" % (k, 1e3 * v / cnt[k], cnt[k])); tot += v / n
print("sum %.1f us per step" % (1e3 * tot))
