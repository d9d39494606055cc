import os, sys
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch, frcnn_amd as F
from test_gpu_model import _OneBatch, _masks
H, W = 128, 176
cfg = dict(F.duplo_cfg); model = F.vgg_small(cfg)
weights, gradient = F.combine_and_flatten_parameters(model["pnet"], model["cnet"], seed=42)
anchors = F.Anchors(model["pnet"], cfg["scales"])
rois = F.synthetic_rois(cfg, W, H, 3, 7, 1)
pos, neg = F.assemble_examples(anchors, cfg, rois, W, H, F.MT19937(11), negatives=8)
sizes = F.output_map_sizes(model, H, W)
pos, neg = F.clean_examples(pos, sizes), F.clean_examples(neg, sizes)
batch = [dict(img=F.synthetic_image(H, W, 1), positive=pos, negative=neg)]
R = len(pos) + len(neg)
rng = np.random.RandomState(5)
nat = model["native"]
bn0 = nat.bn_running.cpu().numpy().copy()
model["pnet"].drop_masks = _masks(rng, model)
model["cnet"].drop_masks = [(rng.rand(R, 1024) > 0.5).astype(np.float32), (rng.rand(R, 512) > 0.5).astype(np.float32)]
seen = {}
for it in range(40):
    mode = it % 2
    F._lib.call("frcnn_set_option", b"side_stream", mode)
    stats = dict(pcls=[], preg=[], dcls=[], dreg=[])
    f = F.create_objective(model, weights, gradient, _OneBatch(batch, anchors), stats)
    loss, grad = f(weights)
    key = tuple(stats[k][-1] for k in ("pcls", "preg", "dcls", "dreg"))
    g = grad.cpu().numpy()
    gk = float(np.abs(g).sum())
    seen.setdefault(mode, set()).add(key + (gk,))
    nat.bn_running.copy_(torch.from_numpy(bn0))
for m, v in seen.items():
    print("mode", m, "distinct results:", len(v))
    for x in sorted(v): print("   ", x)
