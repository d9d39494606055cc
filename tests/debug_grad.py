import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import frcnn_amd as F
import pyoracle as O
from util import oracle_model
cfg = dict(F.duplo_cfg); model = F.vgg_small(cfg)
weights, gradient = F.combine_and_flatten_parameters(model["pnet"], model["cnet"], seed=42)
H, W = 128, 176
it = F.SyntheticBatchIterator(model, H=H, W=W, images_per_batch=1, pool=1, device_images=False)
model["pnet"].drop_masks = [np.ones(l["filters"], np.float32) for l in model["layers"]]
ex = it.pool[0]
sizes = F.output_map_sizes(model, H, W)
ex["positive"] = F.clean_examples(ex["positive"], sizes); ex["negative"] = F.clean_examples(ex["negative"], sizes)
R = len(ex["positive"]) + len(ex["negative"])
cm = [np.ones((R, 1024), np.float32), np.ones((R, 512), np.float32)]
model["cnet"].drop_masks = cm
stats = dict(pcls=[], preg=[], dcls=[], dreg=[])
f = F.create_objective(model, weights, gradient, it, stats)
loss, grad = f(weights)
om = oracle_model(O, cfg); w = weights.cpu().numpy()
g_want = np.zeros_like(w); acc = np.zeros(8); rois = ex["rois"]
pos_idx = np.array([[a.layer, a.aspect, a.index[1], a.index[2], rois.index(r) + 1] for a, r in ex["positive"]], dtype=np.int32).reshape(-1, 5)
pos_rect = np.array([[a.minX, a.minY, a.maxX, a.maxY] for a, r in ex["positive"]], dtype=np.float64).reshape(-1, 4)
neg_idx = np.array([[e[0].layer, e[0].aspect, e[0].index[1], e[0].index[2]] for e in ex["negative"]], dtype=np.int32).reshape(-1, 4)
neg_rect = np.array([[e[0].minX, e[0].minY, e[0].maxX, e[0].maxY] for e in ex["negative"]], dtype=np.float64).reshape(-1, 4)
roi_rect = np.array([[r.rect.minX, r.rect.minY, r.rect.maxX, r.rect.maxY] for r in rois], dtype=np.float64)
roi_cls = np.array([r.class_index for r in rois], dtype=np.int32)
bn = np.concatenate([np.zeros(1024, np.float32), np.ones(1024, np.float32)])
O.train_image(om, w, g_want, ex["img"], pos_idx, pos_rect, roi_rect, roi_cls, neg_idx, neg_rect, model["pnet"].drop_masks, cm, bn, acc)
g_want /= acc[2]
g = grad.cpu().numpy()
print("stats", stats, acc)
for off, cnt, kind, aux in model["native"].param_table:
    a, b = g[off:off+cnt].astype(np.float64), g_want[off:off+cnt].astype(np.float64)
    print("off %9d cnt %8d kind %d  |b| %.3e  rel %.3e  maxabs %.3e" % (off, cnt, kind, np.linalg.norm(b), np.linalg.norm(a-b)/max(np.linalg.norm(b),1e-30), np.abs(a-b).max()))
# ---- structure of the error in the b4c2 weight gradient
off, cnt = 1993606, 1327104
a = g[off:off+cnt].reshape(384, 384, 3, 3).astype(np.float64); b = g_want[off:off+cnt].reshape(384, 384, 3, 3).astype(np.float64)
e = a - b
print("per-tap err norm", np.sqrt((e**2).sum(axis=(0, 1))).round(4), "per-tap |b|", np.sqrt((b**2).sum(axis=(0, 1))).round(2))
eo = np.sqrt((e**2).sum(axis=(1, 2, 3))); ec = np.sqrt((e**2).sum(axis=(0, 2, 3)))
print("worst o", np.argsort(-eo)[:8], eo[np.argsort(-eo)[:8]].round(4), "median", np.median(eo))
print("worst c", np.argsort(-ec)[:8], ec[np.argsort(-ec)[:8]].round(4), "median", np.median(ec))
# ---- unit wgrad with activation, block-4 geometry at this image size (16x22)
rng = np.random.RandomState(0)
C_, Hh, Ww, O_ = 384, 16, 22, 384
x = rng.randn(C_, Hh, Ww).astype(np.float32); gg = rng.randn(O_, Hh, Ww).astype(np.float32) / 10
sl = np.float32(0.25); sc = np.ones(C_, np.float32)
act = np.where(x > 0, x, sl * x) * sc[:, None, None]
gw_want, gb_want = O.conv2d_bwd_weight(act, gg, 3, 3, 1)
dx, dg, dsl, dsc = [F.DeviceTensor.from_numpy(v) for v in (x, gg, np.array([sl]), sc)]
gw = F.DeviceTensor.zeros((O_, C_, 3, 3)); gb = F.DeviceTensor.zeros((O_,))
F._lib.call("frcnn_conv2d_backward_weight", F.ptr(dx), C_, Hh, Ww, F.ptr(dsl), F.ptr(dsc), F.ptr(dg), O_, 3, 1, F.ptr(gw), F.ptr(gb), F.stream_ptr())
ee = gw.numpy().astype(np.float64) - gw_want
print("unit wgrad+act rel err", np.linalg.norm(ee) / np.linalg.norm(gw_want), "per-tap", np.sqrt((ee**2).sum(axis=(0, 1))).round(4))
