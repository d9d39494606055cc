"""Does it train?  N RMSprop steps on the 4-image synthetic pool (vgg_small, 800x450): prints the running mean of the
four loss terms; they must fall and stay finite.  usage: python tools/train_sanity.py [steps]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import frcnn_amd as F

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
cfg = dict(F.duplo_cfg); model = F.vgg_small(cfg)
weights, gradient = F.combine_and_flatten_parameters(model["pnet"], model["cnet"], seed=42)
it = F.SyntheticBatchIterator(model, H=450, W=800, images_per_batch=1, pool=4)
stats = dict(pcls=[], preg=[], dcls=[], dreg=[])
f = F.create_objective(model, weights, gradient, it, stats)
state = dict(learningRate=1e-4, alpha=0.9)
t0 = time.time()
for i in range(steps):
    F.rmsprop(f, weights, state)
    if (i + 1) % 50 == 0:
        k = 48
        print("step %4d  pcls %.4f  preg %.4f  dcls %.4f  dreg %.4f   (%.1f img/s)" % (
            i + 1, np.mean(stats["pcls"][-k:]), np.mean(stats["preg"][-k:]), np.mean(stats["dcls"][-k:]), np.mean(stats["dreg"][-k:]),
            (i + 1) / (time.time() - t0)), flush=True)
w = weights.cpu().numpy()
assert np.isfinite(w).all()
print("weights finite; free/total HBM (GB): %.1f / %.1f" % tuple(x / 2**30 for x in torch.cuda.mem_get_info()))
