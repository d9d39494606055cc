"""One-rank run of the data-parallel exchange through the library's communicator (FRCNN_COMM_FORCE=1: every collective an
identity on real RCCL) with per-bucket timing -- what bench.py --gpus N prints in config.exchange.buckets.
usage: FRCNN_COMM_FORCE=1 python tools/bucket_probe.py"""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import frcnn_amd as F
os.environ["FRCNN_COMM_FORCE"] = "1"
F._lib.call("frcnn_set_device", 0)
cfg = dict(F.duplo_cfg)
model = F.vgg_small(cfg)
weights, gradient = F.combine_and_flatten_parameters(model["pnet"], model["cnet"], seed=42)
it = F.SyntheticBatchIterator(model, H=450, W=800, images_per_batch=1, pool=4)
comm = F.Comm(0, 1, path=os.path.join(tempfile.mkdtemp(), "id"))
F.comm.activate(comm)
f = F.create_objective(model, weights, gradient, it, dict(pcls=[], preg=[], dcls=[], dreg=[]))
state = dict(learningRate=1e-4, alpha=0.9)
for _ in range(6):
    F.rmsprop(f, weights, state)
comm.timing = []
for _ in range(8):
    F.rmsprop(f, weights, state)
print(comm.bucket_times())
comm.destroy()
