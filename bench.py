#!/usr/bin/env python
"""bench.py -- images/sec of the reference's training step (objective.lua lossAndGradient +
optim.rmsprop, main.lua:133) on vgg_small with synthetic 800x450 frames, one image per GPU per
step, data-parallel over N GPUs of one node (gradient all-reduce on RCCL).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Prints ONE JSON line on rank 0 (contract in the task statement), carrying
  "roofline"     -- achieved TFLOP/s of the dominant kernel (conv_x3: forward + input-gradient of the convolutions
                    in the split operand form; conv_igemm 3x3, fp32 MFMA, with the option off)
                    = algorithmic FLOPs of its launches / their HIP-event durations, measured live over the timed
                    steps, vs the matrix-core peak FOR THAT FORMULATION: the fp16 / bf16 dense peak / 3 (two fp16
                    planes per operand, three partial products per fp32 product: option x3_f16 = 1, the default) or
                    / 6 (three bf16 planes, six partial products: x3_f16 = 0); the fp32-MFMA peak for conv_igemm;
  "cpu_baseline" -- the CPU restatement of the reference (oracle/, "port") timed on this host, N=1 only:
                    the training step on the benchmarked 3x450x800 frame with all cores (1 warm-up + median
                    of 3) and with one thread on a 1/16-area frame;
  "parity"       -- the GPU step on the oracle's inputs against the oracle's loss and gradient (the run
                    exits non-zero when they differ);
  "other_legs"   -- BASELINE config 2 (Detector:detect, images/sec, with its own roofline / glue-time split) and nms()
                    alone at n = 300 ... 26 544, each with its CPU-restatement time and its id parity, BASELINE
                    config 5's one-GPU workload (vgg_large), and the headline workload in the two other arithmetic forms
                    (exact_split: three bf16 planes, 24 bits; fp32_mfma: plain fp32 matrix-core products) and with option
                    drop_compact off (dense_dropout: the channels nn.SpatialDropout zeroes are multiplied out), measured
                    after the timed region;
  "sustained"    -- the same metric over >= 12 s of back-to-back steps (~5 000; steady clocks, visible to an SMI sampler; not `value`);
  "roofline_hbm" -- the HBM-bound kernel classes of the step (RMSprop, ROI pooling, element-wise passes): algorithmic
                    bytes / HIP-event time against the 8 TB/s peak.
  --comm native  -- the exchange step through the library's own communicator (frcnn_comm_*, the calls a LuaJIT
                    host makes) instead of torch.distributed (the default since round 5: N > 1 has never run on
                    hardware, and the torch.distributed path is the one the world-size-2 tests exercise).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CUs x 4 SIMD x 64 FLOP/clk x 2.4 GHz
BF16_MFMA_PEAK_TFLOPS = 2516.6  # ... v_mfma_f32_32x32x16_bf16: 256 CUs x 4 SIMD x 1024 FLOP/clk x 2.4 GHz (dense)
SPLIT_PRODUCTS = 6              # bf16 x bf16 partial products per fp32 product in the three-plane split form (convx.hip, x3_f16 = 0)
F16_PRODUCTS = 3                # fp16 x fp16 partial products per fp32 product in the two-plane form (x3_f16 = 1, the default)


def split_products(F):
    """Partial products per fp32 product of the split launches as configured: 3 (two fp16 planes) or 6 (three bf16 planes).
    v_mfma_f32_32x32x16_f16 and _bf16 have the same dense peak."""
    import ctypes
    v = ctypes.c_int(0)
    F._lib.call("frcnn_get_option", b"x3_f16", ctypes.byref(v))
    return F16_PRODUCTS if v.value else SPLIT_PRODUCTS
CONV_CLASSES = ("conv_igemm_k3", "conv_igemm_other", "conv_wgrad_k3", "conv_wgrad_other", "conv_x3", "conv_wgradx")
FULL_H, FULL_W = 450, 800
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec); ~6.3 TB/s is what a float4 copy reaches
SUSTAINED_SECONDS = 12.5     # the sustained pass: long enough for an external SMI sampler
SUSTAINED_STEPS_MAX = 8000


def csrc_sha256():
    """Hash of the kernel sources + the ABI header: profiles/pmc_traffic.json records it when the PMC passes are taken, and
    roofline.traffic is only printed while it still matches (the GPU box has no .git to ask)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "faster-rcnn.torch_amd", "csrc", "*")) + [os.path.join(ROOT, "include", "frcnn_hip.h")])
    for fn in files:
        if os.path.isfile(fn) and fn.rsplit(".", 1)[-1] in ("hip", "cpp", "h"):
            h.update(os.path.basename(fn).encode()); h.update(open(fn, "rb").read())
    return h.hexdigest()[:16]


def hbm_rows(F, launches, ms, by, names, steps):
    """HBM-bound kernels of the step (RMSprop, ROI pooling, the element-wise passes): algorithmic bytes of the
    class's launches (the library's own counters: what each launch must read and write) / their HIP-event durations,
    against the 8 TB/s HBM peak."""
    rows = []
    for name in names:
        i = F._lib.KC_NAMES.index(name)
        if launches[i] and ms[i] > 0:
            gbs = by[i] / 1e9 / (ms[i] / 1e3)
            rows.append(dict(bound="hbm", kernel_class=name, launches_per_step=round(launches[i] / steps, 2), ms_per_step=round(ms[i] / steps, 4),
                             algorithmic_mb_per_step=round(by[i] / 1e6 / steps, 2), achieved=round(gbs, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                             frac=round(gbs / HBM_PEAK_GBS, 4)))
    return rows


def conv_flops_per_image(model, H, W):
    """Algorithmic conv FLOPs of one training step: fwd + dgrad + wgrad, first-layer dgrad skipped."""
    fwd = 0.0
    first = None
    h, w, cin = H, W, 3
    import math
    per_block = []
    for l in model["layers"]:
        for _ in range(l["conv_steps"]):
            f = 2.0 * l["filters"] * cin * l["kW"] * l["kH"] * h * w
            if first is None:
                first = f
            fwd += f
            cin = l["filters"]
        h = int(math.ceil((h - 2) / 2.0)) + 1; w = int(math.ceil((w - 2) / 2.0)) + 1
        per_block.append((h, w, cin))
    for a in model["anchor_nets"]:
        bh, bw, bc = per_block[a["input"] - 1]
        oh, ow = bh - a["kW"] + 1, bw - a["kW"] + 1
        fwd += 2.0 * a["n"] * bc * a["kW"] * a["kW"] * oh * ow + 2.0 * 18 * a["n"] * oh * ow
    return fwd, 3 * fwd - first


def backbone_x3_dense_flops(model, H, W):
    """FLOPs of the launches the conv_x3 class runs in a training step -- forward and input gradient of every backbone convolution
    but the first -- by the DENSE algorithm's formula, whatever option drop_compact leaves out."""
    import math
    tot, h, w, cin, first = 0.0, H, W, 3, True
    for l in model["layers"]:
        for _ in range(l["conv_steps"]):
            if not first:
                tot += 2 * 2.0 * l["filters"] * cin * l["kW"] * l["kH"] * h * w
            first = False
            cin = l["filters"]
        h = int(math.ceil((h - 2) / 2.0)) + 1; w = int(math.ceil((w - 2) / 2.0)) + 1
    return tot


def _host_info():
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return dict(cpu_model=model, logical_cpus=os.cpu_count())


def _oracle():
    sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import pyoracle as O
    from util import oracle_model, oracle_tables
    return O, oracle_model, oracle_tables


def parity_inputs(F, cfg, model, H, W):
    """The inputs of SURVEY 8d for image 0 (what SyntheticBatchIterator holds as pool[0] on rank 0): 4 boxes seed 7,
    examples from findPositive + 16 sampled negatives (MT19937 seed 7) + nearby aversion, N(0,1) frame seed 1000,
    fixed dropout masks."""
    anchors = F.Anchors(model["pnet"], cfg["scales"])
    rois = F.synthetic_rois(cfg, W, H, 4, 7, 0)
    pos, neg = F.assemble_examples(anchors, cfg, rois, W, H, F.MT19937(7))
    sizes = F.output_map_sizes(model, H, W)
    pos, neg = F.clean_examples(pos, sizes), F.clean_examples(neg, sizes)
    R = len(pos) + len(neg)
    rng = np.random.RandomState(0)
    pm = [None if l["dropout"] <= 0 else (rng.rand(l["filters"]) > l["dropout"]).astype(np.float32) for l in model["layers"]]
    cm = [(rng.rand(R, l["n"]) > 0.5).astype(np.float32) for l in model["class_layers"]]
    return dict(anchors=anchors, rois=rois, pos=pos, neg=neg, img=F.synthetic_image(H, W, 0), pm=pm, cm=cm, R=R)


def cpu_train_step(O, om, tables, w0, inp, bn0):
    """One training step of the CPU restatement: lossAndGradient (objective.lua:45-218) + optim.rmsprop (main.lua:133).
    Returns (seconds, loss, gradient)."""
    w = w0.copy(); g = np.zeros_like(w); m = np.zeros_like(w); acc = np.zeros(8); bn = bn0.copy()
    t0 = time.perf_counter()
    O.train_image(om, w, g, inp["img"], *tables, inp["pm"], inp["cm"], bn, acc)
    g /= max(acc[2], 1.0)
    gk = g.copy()
    O.rmsprop(w, g, m, 1e-4, 0.9, 1e-8)
    dt = time.perf_counter() - t0
    loss = acc[0] / acc[2] + (acc[1] / acc[3] if acc[3] else float("nan"))
    return dt, loss, gk


def cpu_baseline(F, cfg, model, w0, bn0):
    """the CPU restatement of the reference semantics (oracle/, fp64 accumulation; NOT Torch7 --
    the Lua reference cannot run here) timed on this host: the training step on the benchmarked 3x450x800 frame with all
    cores (OpenMP), 1 warm-up + median of 3, and with ONE thread on a bounded sample (a 3x113x200 frame, 1/16 of the
    pixels, rate scaled by the pixel ratio: conv work is proportional to pixels).  Also returns the oracle's loss and
    gradient of the full frame for the parity check of the GPU step."""
    O, oracle_model, oracle_tables = _oracle()
    om = oracle_model(O, cfg)
    inp = parity_inputs(F, cfg, model, FULL_H, FULL_W)
    tables = oracle_tables(inp["pos"], inp["neg"], inp["rois"])
    nthreads = O.get_threads()
    times = []
    loss = grad = None
    for k in range(4):      # 1 warm-up + 3
        dt, loss, grad = cpu_train_step(O, om, tables, w0, inp, bn0)
        if k:
            times.append(dt)
    med = sorted(times)[1]
    # one thread, bounded sample
    h1, w1 = 113, 200
    small = parity_inputs(F, cfg, model, h1, w1)
    st = oracle_tables(small["pos"], small["neg"], small["rois"])
    O.set_threads(1)
    try:
        dt1, _, _ = cpu_train_step(O, om, st, w0, small, bn0)
    finally:
        O.set_threads(nthreads)
    ratio = (h1 * w1) / float(FULL_H * FULL_W)
    out = dict(value=round(1.0 / med, 5), unit="images/sec", cores=nthreads, kind="port",
               sample="CPU restatement of reference semantics (oracle/, plain C, fp64 accumulation, OpenMP; not Torch7): training step "
                      "(pnet fwd, RPN loss, ROI pool, cnet fwd/bwd, ROI-pool bwd, pnet bwd, rmsprop) on the benchmarked 3x%dx%d frame, "
                      "%d examples, %d threads, 1 warm-up + median of 3 (%s s)" % (FULL_H, FULL_W, inp["R"], nthreads,
                                                                               "/".join("%.2f" % t for t in times)),
               single_thread=dict(value=round(ratio / dt1, 6), unit="images/sec", cores=1,
                                  sample="same step, 1 thread, one 3x%dx%d frame (%.4f of the pixels, rate scaled by that ratio), "
                                         "%d examples, %.2f s" % (h1, w1, ratio, small["R"], dt1)),
               host=_host_info())
    return out, inp, loss, grad


def gpu_parity_step(F, model, weights, gradient, w0, bn0, inp):
    """The GPU step on the oracle's inputs (explicit dropout masks): returns (loss, gradient as numpy)."""
    import torch
    nat = model["native"]
    weights.copy_(torch.from_numpy(w0)); nat.bn_running.copy_(torch.from_numpy(bn0))
    model["pnet"].drop_masks = inp["pm"]; model["cnet"].drop_masks = inp["cm"]

    class _One(object):
        def nextTraining(self, count=None):
            return [dict(img=inp["img"], positive=inp["pos"], negative=inp["neg"])]
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import decisions
    try:
        stats = dict(pcls=[], preg=[], dcls=[], dreg=[])
        f = F.create_objective(model, weights, gradient, _One(), stats)
        with decisions.CaptureBeforeBackward(F, model, f) as cap:    # the device's pool winners / PReLU branches (tests/decisions.py)
            loss, grad = f(weights)
        g = grad.cpu().numpy().copy()
    finally:
        model["pnet"].drop_masks = None; model["cnet"].drop_masks = None
    return loss, g, cap.captured[0]


def injected_parity(O, om, tables, w0, inp, bn0, captured, g_gpu):
    """The gradient comparison with the device's DISCRETE decisions (2x2 pool winners, ROI arg-max, PReLU branches) handed to the
    oracle, as the strict tests do (tests/test_gpu_fullsize.py): what remains is arithmetic.  Returns (relative L2 of the
    gradient, {kind: (decisions the oracle would have taken differently, of how many)}).  One more oracle step."""
    import decisions
    g = np.zeros_like(w0); acc = np.zeros(8); bn = bn0.copy()
    own = decisions.blank_like(captured)
    with O.decisions(inject=captured, record=own):
        O.train_image(om, w0.copy(), g, inp["img"], *tables, inp["pm"], inp["cm"], bn, acc)
    g /= max(acc[2], 1.0)
    rel = float(np.linalg.norm(g_gpu.astype(np.float64) - g) / np.linalg.norm(g.astype(np.float64)))
    diff = decisions.count_differences(captured, own)
    return rel, {k: dict(differing=int(nd), of=int(nt)) for k, (nd, nt) in diff.items()}


def amplified(nat, w, ncls, gain=30.0):
    """Head logits x30 so that a realistic number of anchors passes Detector.lua:54's p > 0.95 on random weights, class
    head x200 so that some candidates pass the 0.2 confidence gate (Detector.lua:115)."""
    w = w.copy()
    for off, cnt, kind, aux in nat.param_table:
        if kind == 0 and aux == 18:
            v = w[off:off + cnt].reshape(18, -1)
            for a in range(3):
                v[a * 6:a * 6 + 2] *= gain
        if kind == 3 and cnt == 512 * ncls:
            w[off:off + cnt] *= 200.0
    return w


def inference_leg(F, cfg, model, weights, w0, bn0, with_cpu):
    """BASELINE config 2: Detector:detect (Detector.lua:17-141) on synthetic 3x450x800 frames, images resident in HBM;
    beside it the CPU restatement on the same frames (median of 3 after 1 warm-up) and the parity of its outputs."""
    import torch
    nat = model["native"]
    wa = amplified(nat, w0, cfg["class_count"] + 1)
    weights.copy_(torch.from_numpy(wa)); nat.bn_running.copy_(torch.from_numpy(bn0))
    host = [F.synthetic_image(FULL_H, FULL_W, i) for i in range(4)]
    imgs = [F.to_device(x) for x in host]
    n = 40

    def run(static):
        det = F.Detector(model, static_weights=static)
        if not static:
            F._lib.call("frcnn_set_option", b"static_weights", 0)
        for i in range(4):
            res = det.detect(imgs[i])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            res = det.detect(imgs[i % 4])
        torch.cuda.synchronize()
        return det, res, (time.perf_counter() - t0) / n
    # a detector serves a trained model: the weights do not change between frames and their packed / split copies are made once
    # (option static_weights); the time with the copies remade for every frame, as a training loop's validation pass would, beside it
    _, _, dt_repack = run(False)
    d, r, dt = run(True)
    out = dict(metric="images/sec (vgg_small 800x450 inference: Detector:detect)", value=round(1.0 / dt, 2), ms_per_image=round(dt * 1e3, 3),
               frames=n, matches=int(d.last_scan["n"]), candidates=int(len(d.last_pick)), winners=len(r),
               weights="packed once (frcnn_set_option static_weights = 1: the host does not write the weights between frames)",
               ms_per_image_weights_repacked_every_frame=round(dt_repack * 1e3, 3),
               note="head logits amplified x30 (random weights would pass no anchor at p > 0.95)")
    # where a frame's time goes: 8 more frames with every kernel class bracketed by HIP events (outside the timed frames)
    nk = len(F._lib.KC_NAMES)
    sink = [(C.c_longlong * nk)(), (C.c_double * nk)(), (C.c_double * nk)(), (C.c_double * nk)()]
    F._lib.call("frcnn_prof_collect", *sink)
    F._lib.call("frcnn_prof_enable", (1 << nk) - 1)
    for i in range(8):
        d.detect(imgs[i % 4])
    torch.cuda.synchronize()
    F._lib.call("frcnn_prof_enable", 0)
    la, ms, fl, by = [(C.c_longlong * nk)(), (C.c_double * nk)(), (C.c_double * nk)(), (C.c_double * nk)()]
    F._lib.call("frcnn_prof_collect", la, ms, fl, by)
    k = F._lib.KC_NAMES.index("conv_x3")
    if ms[k] > 0:
        peak = BF16_MFMA_PEAK_TFLOPS / split_products(F)
        ach = (fl[k] / 1e12) / (ms[k] / 1e3)
        conv_ms = sum(ms[F._lib.KC_NAMES.index(c)] for c in CONV_CLASSES) / 8.0
        out["roofline"] = dict(bound="mfma", kernel="conv_x3_kernel (3x3 forward launches of the frame)", achieved=round(ach, 2), peak=round(peak, 1),
                               unit="TFLOP/s", frac=round(ach / peak, 4), launches_per_frame=la[k] / 8.0, avg_launch_ms=round(ms[k] / max(la[k], 1), 4))
        out["kernel_ms_per_frame"] = {name: round(ms[i] / 8.0, 4) for i, name in enumerate(F._lib.KC_NAMES) if la[i]}
        out["conv_ms_per_frame"] = round(conv_ms, 4)
        out["glue_ms_per_frame"] = round(dt * 1e3 - conv_ms, 4)   # scan, NMS, ROI pooling, classification net, read-backs, host
        out["roofline_hbm"] = hbm_rows(F, la, ms, by, ("roi", "nms", "rpn"), 8)
    if with_cpu:
        O, oracle_model, _ = _oracle()
        om = oracle_model(O, cfg)
        times = []
        same_matches = same_picks = same_winners = None
        for k in range(4):
            t0 = time.perf_counter()
            ref = O.detect(om, wa, bn0, host[k % 4])
            if k:
                times.append(time.perf_counter() - t0)
        # parity on the last frame the oracle saw (k = 3): anchor indices, NMS candidate ids, winner classes
        win = d.detect(imgs[3])
        gi = d.last_scan["idx"].numpy()
        same_matches = bool(gi.shape == ref["match_idx"].shape and np.array_equal(gi, ref["match_idx"]))
        same_picks = bool(list(d.last_pick) == ref["cand_ids"].tolist())
        same_winners = bool([x["class"] for x in win] == [int(v[0]) for v in ref["winners"]])
        # NMS ids on identical boxes are always required to be bit-exact (fp32 activations may move an anchor across 0.95)
        picks_on_gpu_boxes = bool(list(d.last_pick) == O.nms(d.last_scan["box"].numpy(), 0.25).tolist())
        out["cpu_baseline"] = dict(value=round(1.0 / sorted(times)[1], 4), unit="images/sec", cores=O.get_threads(), kind="port",
                                   sample="orc_detect on the same frames, 1 warm-up + median of 3 (%s s)" % "/".join("%.2f" % t for t in times))
        out["parity"] = dict(match_indices_identical=same_matches, nms_candidates_identical=same_picks,
                             winner_classes_identical=same_winners, nms_ids_identical_on_gpu_boxes=picks_on_gpu_boxes,
                             oracle_matches=int(len(ref["match_idx"])), oracle_candidates=int(len(ref["cand_ids"])),
                             oracle_winners=int(len(ref["winners"])))
    weights.copy_(torch.from_numpy(w0))
    F._lib.call("frcnn_set_option", b"static_weights", 0)   # (the legs that follow write the weights)
    return out


def large_leg(F, with_cpu, steps=12):
    """BASELINE config 5's one-GPU workload: vgg_large (models/vgg_large.lua:5-22), one synthetic 3x600x1000 frame per step,
    config/imagenet.lua values (200 classes, scales 48..384, 6x6 ROI pooling): images/sec of the training step, the live
    roofline fraction of its dominant kernel, and -- one sample, ~30 s -- the CPU restatement's step on the same frame with
    the parity of loss and gradient."""
    import torch
    H, W = 600, 1000
    cfg = dict(F.imgnet_cfg)
    model = F.vgg_large(cfg)
    weights, gradient = F.combine_and_flatten_parameters(model["pnet"], model["cnet"], seed=42)
    it = F.SyntheticBatchIterator(model, H=H, W=W, images_per_batch=1, pool=4)
    stats = dict(pcls=[], preg=[], dcls=[], dreg=[])
    f = F.create_objective(model, weights, gradient, it, stats)
    state = dict(learningRate=1e-4, alpha=0.9)
    tw = time.perf_counter()
    nw = 0
    while nw < 6 or time.perf_counter() - tw < 1.0:    # (about a second: the device idled through the legs before, see arithmetic_leg)
        F.rmsprop(f, weights, state); nw += 1
    torch.cuda.synchronize()
    conv_mask = sum(1 << F._lib.KC_NAMES.index(n) for n in CONV_CLASSES)
    nk = len(F._lib.KC_NAMES)
    sink = [(C.c_longlong * nk)(), (C.c_double * nk)(), (C.c_double * nk)(), (C.c_double * nk)()]
    F._lib.call("frcnn_prof_collect", *sink)      # (drain)
    sampled = 0
    t0 = time.perf_counter()
    for i in range(steps):
        on = i % 4 == 0
        if on:
            F._lib.call("frcnn_prof_enable", conv_mask); sampled += 1
        F.rmsprop(f, weights, state)
        if on:
            F._lib.call("frcnn_prof_enable", 0)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    launches, ms, fl, by = [(C.c_longlong * nk)(), (C.c_double * nk)(), (C.c_double * nk)(), (C.c_double * nk)()]
    F._lib.call("frcnn_prof_collect", launches, ms, fl, by)
    k = F._lib.KC_NAMES.index("conv_x3")
    peak = BF16_MFMA_PEAK_TFLOPS / split_products(F)
    ach = (fl[k] / 1e12) / (ms[k] / 1e3) if ms[k] > 0 else 0.0
    _, train_flops = conv_flops_per_image(model, H, W)
    out = dict(metric="images/sec (vgg_large 1000x600 fwd+bwd, config/imagenet.lua: 200 classes, 45 015 anchors, 6x6 ROI pooling)",
               value=round(1.0 / dt, 2), ms_per_step=round(dt * 1e3, 3), steps=steps,
               examples_per_image=[len(b["positive"]) + len(b["negative"]) for b in it.pool],
               conv_gflop_per_image=round(train_flops / 1e9, 1), whole_step_conv_tflops=round(train_flops / 1e12 / dt, 1),
               roofline=dict(bound="mfma", kernel="conv_x3_kernel", achieved=round(ach, 2), peak=round(peak, 1), unit="TFLOP/s",
                             frac=round(ach / peak, 4), launches_per_step=launches[k] / max(sampled, 1),
                             avg_launch_ms=round(ms[k] / max(launches[k], 1), 4)))
    out["kernel_classes"] = {name: dict(launches_per_step=launches[i] / max(sampled, 1), ms_per_step=round(ms[i] / max(sampled, 1), 4),
                                        tflops=round((fl[i] / 1e12) / (ms[i] / 1e3), 2) if fl[i] > 0 and ms[i] > 0 else None)
                             for i, name in enumerate(F._lib.KC_NAMES) if launches[i]}
    if with_cpu:
        O, oracle_model, oracle_tables = _oracle()
        om = oracle_model(O, cfg, model["layers"], model["anchor_nets"], model["class_layers"])
        nat = model["native"]
        w0 = nat.init_parameters(42)
        bn0 = np.concatenate([np.zeros(1024, np.float32), np.ones(1024, np.float32)])
        inp = parity_inputs(F, cfg, model, H, W)
        tables = oracle_tables(inp["pos"], inp["neg"], inp["rois"])
        dtc, o_loss, o_grad = cpu_train_step(O, om, tables, w0, inp, bn0)
        g_loss, g_grad, captured = gpu_parity_step(F, model, weights, gradient, w0, bn0, inp)
        rel = float(np.linalg.norm(g_grad.astype(np.float64) - o_grad) / np.linalg.norm(o_grad.astype(np.float64)))
        rel_inj, differing = injected_parity(O, om, tables, w0, inp, bn0, captured, g_grad)
        out["cpu_baseline"] = dict(value=round(1.0 / dtc, 5), unit="images/sec", cores=O.get_threads(), kind="port",
                                   sample="ONE training step of the CPU restatement on the same 3x600x1000 frame, %d examples (%.1f s, no warm-up)" % (inp["R"], dtc))
        out["parity"] = dict(loss_gpu=g_loss, loss_oracle=float(o_loss), gradient_rel_l2=rel, gradient_rel_l2_injected=rel_inj,
                             decisions_differing=differing, examples=inp["R"],
                             ok=bool(abs(g_loss - o_loss) <= 1e-5 * max(1.0, abs(o_loss)) and rel <= 1e-3 and rel_inj <= 1e-3))
    del f, it, model, weights, gradient
    torch.cuda.empty_cache()
    return out


def dtype_text(F):
    """`dtype` of the JSON line: tensors, accumulation and results are fp32; what the options change is the form in which the
    OPERANDS of the matrix-core products are presented (say so in the field itself)."""
    v = C.c_int(0)
    F._lib.call("frcnn_get_option", b"split_bf16", C.byref(v)); split = v.value
    F._lib.call("frcnn_get_option", b"x3_f16", C.byref(v)); f16 = v.value
    if not split:
        return "f32 (fp32 operands, v_mfma_f32_32x32x2_f32)"
    if f16:
        return "f32 (conv / Linear(13824,1024) operands as 2 x fp16 split planes: 22 significand bits, 3 MFMA partial products; fp32 accumulate)"
    return "f32 (conv operands as 3 x bf16 split planes: 24 significand bits, 6 MFMA partial products; fp32 accumulate)"


def dropout_text(F):
    """What the pass does with the channels nn.SpatialDropout(0.4) drops (option drop_compact, include/frcnn_hip.h)."""
    v = C.c_int(0)
    F._lib.call("frcnn_get_option", b"drop_compact", C.byref(v))
    return ("drop_compact = 1: the channels nn.SpatialDropout (models/model_utilities.lua:10-12) zeroes are left out of the step's "
            "convolutions (same results as multiplying by the zeros; the roofline counts the FLOPs of the launches as run); "
            "other_legs.dense_dropout times the step with the option off" if v.value else
            "drop_compact = 0: the convolutions multiply the zeros of nn.SpatialDropout like the reference does")


def arithmetic_leg(F, name, options, steps=20):
    """The headline workload (vgg_small 800x450 training step) in another arithmetic form, driver-timed beside the headline
    (VERDICT r5 next 3): `options` are frcnn_set_option pairs set before the model is shaped (split_bf16 only takes effect for
    models shaped afterwards) and restored afterwards.  20 steps after about a second of warm-up steps, the convolution classes bracketed on
    every 4th step, own roofline against the peak of THAT form."""
    import torch
    before = {}
    v = C.c_int(0)
    for k, val in options.items():
        F._lib.call("frcnn_get_option", k.encode(), C.byref(v)); before[k] = v.value
        F._lib.call("frcnn_set_option", k.encode(), val)
    try:
        cfg = dict(F.duplo_cfg)
        model = F.vgg_small(cfg)
        weights, gradient = F.combine_and_flatten_parameters(model["pnet"], model["cnet"], seed=42)
        it = F.SyntheticBatchIterator(model, H=FULL_H, W=FULL_W, images_per_batch=1, pool=4)
        f = F.create_objective(model, weights, gradient, it, dict(pcls=[], preg=[], dcls=[], dreg=[]))
        state = dict(learningRate=1e-4, alpha=0.9)
        # warm-up: every pooled image once (workspaces), then about a second of steps -- the legs before this one end with tens of
        # seconds of CPU-oracle work during which the device idles, and the first ~100 ms after that run at ramping clocks
        # (round 6: 300 images/s measured right after the idle period against 352 in a fresh process)
        tw = time.perf_counter()
        nw = 0
        while nw < 7 or time.perf_counter() - tw < 1.0:
            F.rmsprop(f, weights, state); nw += 1
        torch.cuda.synchronize()
        conv_mask = sum(1 << F._lib.KC_NAMES.index(n) for n in CONV_CLASSES)
        nk = len(F._lib.KC_NAMES)
        sink = [(C.c_longlong * nk)(), (C.c_double * nk)(), (C.c_double * nk)(), (C.c_double * nk)()]
        F._lib.call("frcnn_prof_collect", *sink)      # (drain)
        sampled = 0
        t0 = time.perf_counter()
        for i in range(steps):
            on = i % 4 == 0
            if on:
                F._lib.call("frcnn_prof_enable", conv_mask); sampled += 1
            F.rmsprop(f, weights, state)
            if on:
                F._lib.call("frcnn_prof_enable", 0)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        la, ms, fl, by = [(C.c_longlong * nk)(), (C.c_double * nk)(), (C.c_double * nk)(), (C.c_double * nk)()]
        F._lib.call("frcnn_prof_collect", la, ms, fl, by)
        split_on = la[F._lib.KC_NAMES.index("conv_x3")] > 0
        k = F._lib.KC_NAMES.index("conv_x3" if split_on else "conv_igemm_k3")
        nprod = split_products(F)
        peak = BF16_MFMA_PEAK_TFLOPS / nprod if split_on else FP32_MFMA_PEAK_TFLOPS
        ach = (fl[k] / 1e12) / (ms[k] / 1e3) if ms[k] > 0 else 0.0
        out = dict(metric="images/sec (vgg_small %dx%d fwd+bwd)" % (FULL_W, FULL_H), options=options, dtype=dtype_text(F),
                   value=round(1.0 / dt, 2), unit="images/sec", ms_per_step=round(dt * 1e3, 3), steps=steps,
                   roofline=dict(bound="mfma", kernel=F._lib.KC_NAMES[k], achieved=round(ach, 2), peak=round(peak, 1), unit="TFLOP/s",
                                 frac=round(ach / peak, 4), launches_per_step=la[k] / max(sampled, 1),
                                 avg_launch_ms=round(ms[k] / max(la[k], 1), 4),
                                 executed_tflops=round(nprod * ach, 1) if split_on else round(ach, 2)))
        del f, it, model, weights, gradient
        torch.cuda.empty_cache()
        return out
    finally:
        for k, val in before.items():
            F._lib.call("frcnn_set_option", k.encode(), val)


def nms_leg(F, with_cpu):
    """nms() alone at n = 300 / 2000 / 6000 / 26544 boxes (unique y2 keys), thresholds 0.25 and 0.1;
    boxes resident in HBM, ids read back; median of 10.  CPU restatement: 1 thread (nms.lua is a serial loop)."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from util import random_boxes
    O = _oracle()[0] if with_cpu else None
    rows = []
    for n in (300, 2000, 6000, 26544):
        b = random_boxes(np.random.RandomState(n), n)
        db = F.DeviceTensor.from_numpy(b)
        for thr in (0.25, 0.1):
            pick = F.nms(db, thr, None)
            ts = []
            for _ in range(10):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                pick = F.nms(db, thr, None)
                ts.append(time.perf_counter() - t0)
            row = dict(n=n, overlap=thr, kept=int(len(pick)), gpu_ms=round(sorted(ts)[5] * 1e3, 4))
            if O is not None:
                cs = []
                for _ in range(10 if n <= 6000 else 3):
                    t0 = time.perf_counter()
                    want = O.nms(b, thr)
                    cs.append(time.perf_counter() - t0)
                row["cpu_ms"] = round(sorted(cs)[len(cs) // 2] * 1e3, 3)
                row["ids_identical"] = bool(list(pick) == want.tolist())
            rows.append(row)
    return rows


# xGMI figures of SURVEY 8e / MI355X_MICROARCH.md: 8 GPUs, full mesh, 7 links x ~153 GB/s per GPU and direction
XGMI_LINK_GBS = 153.0


def exchange_schedule(F, step, n_probe=3, ranks=8):
    """N = 1: the bucket schedule of the data-parallel exchange as the step would issue it (objective.exchange_probe marks the
    program points where each bucket's all-reduce starts at N > 1), with the time into the step at which each bucket's
    gradient slice is final, and a MODEL of the all-reduce time at 8 ranks from the xGMI figures -- something a first
    multi-GPU run can be compared with.  Not a measurement of any collective."""
    import sys as _sys
    import torch
    obj = _sys.modules["frcnn_amd.objective"]
    rows = {}
    order = []
    span = []
    for _ in range(n_probe):
        obj.exchange_probe = []
        try:
            step()
            torch.cuda.synchronize()
            marks = obj.exchange_probe
        finally:
            obj.exchange_probe = None
        t0 = [m for m in marks if m[0] == "step_begin"][0][4]
        for label, lo, hi, waits_on, ev in marks:
            if label == "step_begin":
                continue
            t = t0.elapsed_time(ev)
            if label == "backward_end":
                span.append(t); continue
            if label not in rows:
                rows[label] = dict(bucket=label, elements=hi - lo, bytes=4 * (hi - lo), waits_on=waits_on, final_ms=[])
                order.append(label)
            rows[label]["final_ms"].append(t)
    out = []
    for label in order:
        r = rows[label]
        r["final_ms_into_step"] = round(sorted(r.pop("final_ms"))[len(span) // 2], 3)
        out.append(r)
    bwd_end = sorted(span)[len(span) // 2]
    # model: a bucket's all-reduce starts when its slice is final and the previous bucket's collective is done (one RCCL stream),
    # and moves 2 (N-1)/N x bytes per rank; (a) one ring over ONE link per direction (the per-link bound of SURVEY 8e), (b) seven
    # rings / a direct reduce-scatter + all-gather over all 7 links (the best a full mesh allows).  Latency ~ 20 us per collective.
    preds = {}
    for name, links in (("one_ring_one_link", 1), ("all_seven_links", 7)):
        t_end = 0.0
        for r in sorted(out, key=lambda r: r["final_ms_into_step"]):
            dur = 0.020 + 2.0 * (ranks - 1) / ranks * r["bytes"] / (links * XGMI_LINK_GBS * 1e9) * 1e3
            t_end = max(t_end, r["final_ms_into_step"]) + dur
            r.setdefault("model_allreduce_ms", {})[name] = round(dur, 3)
        preds[name] = dict(last_bucket_done_ms_into_step=round(t_end, 3), exposed_ms_after_backward=round(max(0.0, t_end - bwd_end), 3))
    return dict(what="N = 1 probe: where each bucket of the flat gradient becomes final in the step, and a model (not a measurement) of "
                     "its all-reduce at %d ranks over xGMI" % ranks,
                buckets=out, backward_end_ms_into_step=round(bwd_end, 3), ranks_modelled=ranks, link_GBs=XGMI_LINK_GBS,
                model=preds,
                note="exposed = time the update would wait for the last bucket after the backward pass ends; the buckets run on RCCL's "
                     "own stream beside the backbone's backward pass (they take CUs from it: FRCNN_COMM_CHANNELS has no measured default)")


def _fail(msg, code=2):
    sys.stderr.write("bench.py: ERROR: %s\n" % msg)
    sys.stderr.flush()
    os._exit(code)


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N ranks of this script, one per device of this node (what
    `python -m torch.distributed.run --nproc-per-node N` would do), each with RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_ADDR / MASTER_PORT and a job nonce for the communicator's rendezvous (frcnn_comm_exchange_id_file).  Fails
    loudly when the node has fewer than N devices; the exit status is the first failing rank's."""
    import socket
    import subprocess
    import torch
    n = args.gpus
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    shared = os.environ.get("FRCNN_DIST_BACKEND", "nccl") != "nccl"   # (debug: gloo lets the ranks share devices, see main)
    if have < n and not (shared and have >= 1):
        _fail("--gpus %d needs %d HIP devices on this node, found %d (no oversubscription, no CPU fallback)" % (n, n, have))
    sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
    env = dict(os.environ)
    env.update(WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LOCAL_WORLD_SIZE=str(n),
               FRCNN_COMM_NONCE="bench:%d:%d:%d" % (os.getpid(), port, time.time_ns()))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL's intra-node transport needs it on this driver
    procs = []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=e))
    rc = 0
    try:
        alive = list(procs)
        while alive:
            for p in list(alive):
                c = p.poll()
                if c is None:
                    continue
                alive.remove(p)
                if c != 0 and rc == 0:
                    rc = c
                    for q in alive:      # one rank failed: the others would wait in a collective for ever
                        q.terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    sys.exit(rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--height", type=int, default=FULL_H)
    ap.add_argument("--width", type=int, default=FULL_W)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--model", default="vgg_small", choices=["vgg_small", "vgg_large"],
                    help="vgg_large = SURVEY 8d config 5 (config/imagenet.lua, use --height 600 --width 1000); not the bench line")
    ap.add_argument("--profile-all", action="store_true", help="HIP-event profile of every kernel class (adds overhead)")
    ap.add_argument("--no-other-legs", action="store_true", help="skip the inference / nms legs")
    ap.add_argument("--no-upload-leg", action="store_true", help="skip the PCIe-inclusive pass (frames uploaded every step)")
    ap.add_argument("--no-sustained", action="store_true", help="skip the %.0f-second sustained pass" % SUSTAINED_SECONDS)
    ap.add_argument("--comm", default=os.environ.get("FRCNN_COMM", "torch"), choices=["torch", "native"],
                    help="exchange back end at N > 1: torch.distributed ('nccl' = RCCL; the default -- the multi-rank path the "
                         "world-size-2 tests cover) or the C ABI's own frcnn_comm_* (RCCL bound by the library; what a LuaJIT "
                         "host calls)")
    args = ap.parse_args()
    if args.gpus < 1:
        _fail("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return spawn_ranks(args)     # no launcher: this process becomes the launcher of N ranks

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        _fail("--gpus %d but the launcher started WORLD_SIZE=%d ranks: the line would be mislabelled" % (args.gpus, world))
    if not torch.cuda.is_available():
        _fail("bench.py needs a HIP device (the product path has no CPU fallback)")
    if os.environ.get("FRCNN_DIST_BACKEND", "nccl") == "nccl" and torch.cuda.device_count() < world:
        _fail("%d ranks on this node but only %d HIP devices" % (world, torch.cuda.device_count()))
    # (debug only: FRCNN_DIST_BACKEND=gloo lets several ranks share one GPU to exercise this branch on a 1-GPU box)
    backend = os.environ.get("FRCNN_DIST_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    native_comm = None
    exchange_info = None
    # RCCL (and gloo) print banners while a communicator comes up -- on stdout, which must carry ONE JSON line: file
    # descriptor 1 points at stderr until the communicator exists
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)
    if world > 1 and args.comm == "native":
        import frcnn_amd as F0
        F0._lib.call("frcnn_set_device", local_rank)
        native_comm = F0.Comm.from_env()       # RCCL through the C ABI; no torch.distributed process group at all
        F0.comm.activate(native_comm)
        # what RCCL itself says about the communicator: N ranks, this rank, this device -- on N distinct devices
        cnt, urank, dev = native_comm.query()
        counts = native_comm.gather_ints(cnt); devs = native_comm.gather_ints(dev)
        if cnt != world or urank != rank or dev != local_rank or counts != [world] * world or sorted(devs) != list(range(world)):
            _fail("communicator check failed on rank %d: ncclCommCount %d (want %d), user rank %d, device %d; all ranks: "
                  "counts %s devices %s" % (rank, cnt, world, urank, dev, counts, devs), 4)
        exchange_info = dict(ncclCommCount_per_rank=counts, device_per_rank=devs)
    elif world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if os.environ["MASTER_ADDR"] in ("127.0.0.1", "localhost"):
            # one node: the host-side (gloo) subgroup of the objective binds to loopback instead of resolving the hostname
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
        dist.init_process_group(backend)  # "nccl" = RCCL over xGMI
        if dist.get_world_size() != world:
            _fail("process group has %d ranks, expected %d" % (dist.get_world_size(), world), 4)
        exchange_info = dict(world_size_per_rank=[dist.get_world_size()] * world)

    import frcnn_amd as F
    L = F._lib.load()
    F._lib.call("frcnn_set_device", local_rank)
    if world > 1:    # first collective (communicators may finish their set-up lazily), still with stdout parked
        if native_comm is not None:
            native_comm.barrier()
        else:
            dist.barrier()
    sys.stdout.flush()
    os.dup2(saved_stdout, 1)
    os.close(saved_stdout)
    cfg = dict(F.duplo_cfg if args.model == "vgg_small" else F.imgnet_cfg)
    model = (F.vgg_small if args.model == "vgg_small" else F.vgg_large)(cfg)
    weights, gradient = F.combine_and_flatten_parameters(model["pnet"], model["cnet"], seed=42)  # same seed on every rank
    H, W = args.height, args.width
    it = F.SyntheticBatchIterator(model, H=H, W=W, images_per_batch=1, rank=rank, world_size=world, pool=4)
    stats = dict(pcls=[], preg=[], dcls=[], dreg=[])
    f = F.create_objective(model, weights, gradient, it, stats)
    state = dict(learningRate=1e-4, alpha=0.9)  # main.lua:122

    # (experiments: FRCNN_BENCH_STREAM=1 a stream of the caller's own instead of the NULL stream, =hi the same with high priority)
    user_stream = (torch.cuda.Stream(priority=-1) if os.environ.get("FRCNN_BENCH_STREAM") == "hi"
                   else torch.cuda.Stream() if os.environ.get("FRCNN_BENCH_STREAM") else None)

    def step():
        if user_stream is not None:
            with torch.cuda.stream(user_stream):
                F.rmsprop(f, weights, state)
            return
        F.rmsprop(f, weights, state)  # main.lua:133

    def barrier():
        if native_comm is not None:
            native_comm.barrier()
        elif world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(len(it.pool)):   # set-up: every pooled image once, so that no workspace grows inside the timed region
        step()
    for _ in range(args.warmup):
        step()
    # Live HIP-event bracketing puts two event packets around every bracketed launch of the dependent chain (~3 % of
    # the step when every step is bracketed): the conv classes are bracketed on every 4th step of the timed region
    # (a sample of the same launches), --profile-all brackets every class on every step.
    conv_mask = sum(1 << F._lib.KC_NAMES.index(n) for n in CONV_CLASSES)
    mask = (1 << len(F._lib.KC_NAMES)) - 1 if args.profile_all else conv_mask
    every = 1 if args.profile_all else 4
    sampled = 0
    barrier()
    if native_comm is not None:
        native_comm.timing = []       # per-bucket all-reduce durations of the timed region (config.exchange.buckets)
    state["_timing"] = dict(enqueue=0.0, wait=0.0, steps=0)   # utilities.rmsprop fills it: host time per step
    t0 = time.perf_counter()
    for i in range(args.steps):
        on = (i % every) == 0
        if on:
            F._lib.call("frcnn_prof_enable", mask)
            sampled += 1
        step()
        if on:
            F._lib.call("frcnn_prof_enable", 0)
    barrier()
    dt = time.perf_counter() - t0
    if native_comm is not None:      # the job's time is the slowest rank's (every rank gets the same number: it also sizes
        dt = native_comm.gather_max(dt)   # the sustained pass, which is a sequence of collectives)
    elif world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    host_t = state.pop("_timing")
    bucket_times = None
    if native_comm is not None:
        bucket_times = native_comm.bucket_times()
        native_comm.timing = None
    nk = len(F._lib.KC_NAMES)
    launches = (C.c_longlong * nk)(); ms = (C.c_double * nk)(); fl = (C.c_double * nk)(); by = (C.c_double * nk)()
    F._lib.call("frcnn_prof_collect", launches, ms, fl, by)
    # The same metric over SUSTAINED_SECONDS of back-to-back steps (nothing bracketed): clocks and
    # thermals are steady by then and an external SMI sampler sees the device busy.  `value` stays the K timed steps above.
    sustained = None
    if not args.no_sustained:
        # (SUSTAINED_SECONDS of steps at the speed just measured -- about 5 000 of them; a debug back end that takes hundreds of
        # milliseconds per step gets the same seconds, not the same count)
        n_sus = max(20, min(SUSTAINED_STEPS_MAX, int(SUSTAINED_SECONDS / max(dt / args.steps, 1e-4))))
        barrier()
        ts = time.perf_counter()
        for _ in range(n_sus):
            step()
        barrier()
        ts = time.perf_counter() - ts
        if native_comm is not None:
            ts = native_comm.gather_max(ts)
        elif world > 1:
            tmx = torch.tensor([ts], dtype=torch.float64, device="cuda")
            dist.all_reduce(tmx, op=dist.ReduceOp.MAX)
            ts = float(tmx.item())
        sustained = dict(steps=n_sus, value=round(world * n_sus / ts, 3), unit="images/sec",
                         ms_per_step=round(1e3 * ts / n_sus, 3), seconds=round(ts, 2))
    # HBM-bound kernels of the live step: four more steps with those classes bracketed
    hbm = None
    if world == 1:
        hbm_names = ("optim", "roi", "elemwise")
        F._lib.call("frcnn_prof_enable", sum(1 << F._lib.KC_NAMES.index(n) for n in hbm_names))
        for _ in range(4):
            step()
        torch.cuda.synchronize()
        F._lib.call("frcnn_prof_enable", 0)
        lh = (C.c_longlong * nk)(); mh = (C.c_double * nk)(); fh = (C.c_double * nk)(); bh = (C.c_double * nk)()
        F._lib.call("frcnn_prof_collect", lh, mh, fh, bh)
        hbm = hbm_rows(F, lh, mh, bh, hbm_names, 4)
    sched = None
    if world == 1:
        try:
            sched = exchange_schedule(F, step)
        except Exception as e:    # (a probe: never fails the bench line)
            sched = dict(error=repr(e))
    # Second, untimed pass with the library's side stream off: in the timed region the 3x3 input-gradient
    # launches share the CUs with the weight-gradient launches of the side stream, so their live duration
    # (roofline.achieved, as prescribed) is longer than the kernel needs when it has the GPU to itself.
    iso = None
    iso_classes = None
    if world == 1:   # (a lone rank cannot step: the objective all-reduces)
        F._lib.call("frcnn_set_option", b"side_stream", 0)
        step()
        barrier_local = torch.cuda.synchronize
        barrier_local()
        F._lib.call("frcnn_prof_enable", conv_mask)
        for _ in range(3):
            step()
        barrier_local()
        F._lib.call("frcnn_prof_enable", 0)
        l2 = (C.c_longlong * nk)(); m2 = (C.c_double * nk)(); f2 = (C.c_double * nk)(); b2 = (C.c_double * nk)()
        F._lib.call("frcnn_prof_collect", l2, m2, f2, b2)
        F._lib.call("frcnn_set_option", b"side_stream", 1)
        iso_classes = {}
        for i, name in enumerate(F._lib.KC_NAMES):
            if l2[i]:
                iso_classes[name] = dict(launches_per_step=l2[i] / 3.0, ms_per_step=round(m2[i] / 3.0, 4),
                                         tflops=round((f2[i] / 1e12) / (m2[i] / 1e3), 2) if f2[i] > 0 and m2[i] > 0 else None)
        split_on = l2[F._lib.KC_NAMES.index("conv_x3")] > 0
        kdom = F._lib.KC_NAMES.index("conv_x3" if split_on else "conv_igemm_k3")
        peak_dom = BF16_MFMA_PEAK_TFLOPS / split_products(F) if split_on else FP32_MFMA_PEAK_TFLOPS
        if m2[kdom] > 0:
            a2 = (f2[kdom] / 1e12) / (m2[kdom] / 1e3)
            iso = dict(achieved=round(a2, 2), frac=round(a2 / peak_dom, 4), avg_launch_ms=round(m2[kdom] / max(l2[kdom], 1), 4),
                       note="same kernel with frcnn_set_option('side_stream', 0): no concurrent weight-gradient launches; with the side stream off the "
                            "anchor nets run as dense convolutions (option sparse_heads needs it), so this pass has their four conv_x3 launches too")
    # objective.lua:66 uploads every frame (`x.img:cuda()`); the bench contract keeps inputs resident in HBM for `value`.
    # The PCIe-inclusive rate is measured here, outside the timed region: the same step fed by an iterator whose frames
    # live in page-locked host memory and cross PCIe every step (copy stream + ring of device buffers, one step ahead).
    upload_leg = None
    if world == 1 and not args.no_upload_leg:
        it_up = F.SyntheticBatchIterator(model, H=H, W=W, images_per_batch=1, rank=rank, world_size=world, pool=4, upload=True)
        f_up = F.create_objective(model, weights, gradient, it_up, dict(pcls=[], preg=[], dcls=[], dreg=[]))
        for _ in range(4 + args.warmup):
            F.rmsprop(f_up, weights, state)
        torch.cuda.synchronize()
        tu = time.perf_counter()
        for _ in range(args.steps):
            F.rmsprop(f_up, weights, state)
        torch.cuda.synchronize()
        tu = time.perf_counter() - tu
        upload_leg = dict(value=round(args.steps / tu, 3), unit="images/sec", ms_per_step=round(1e3 * tu / args.steps, 3), steps=args.steps,
                          bytes_per_frame=int(3 * H * W * 4),
                          note="same step with the frame uploaded from page-locked host memory every step (objective.lua:66 "
                               "x.img:cuda()), asynchronously on a copy stream one step ahead; not the headline value")
    if rank == 0:
        fwd_flops, train_flops = conv_flops_per_image(model, H, W)
        split_on = launches[F._lib.KC_NAMES.index("conv_x3")] > 0
        k = F._lib.KC_NAMES.index("conv_x3" if split_on else "conv_igemm_k3")
        nprod = split_products(F)
        peak = BF16_MFMA_PEAK_TFLOPS / nprod if split_on else FP32_MFMA_PEAK_TFLOPS
        ach = (fl[k] / 1e12) / (ms[k] / 1e3) if ms[k] > 0 else 0.0
        traffic = None
        traffic_taken = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                traffic = tj.get("%s_bytes_per_launch" % F._lib.KC_NAMES[k])
                traffic_taken = tj.get("taken")    # {git, date, csrc_sha256}: which tree the PMC passes measured
                # a table taken from other kernel sources than the ones that just ran is not evidence about them: refuse it
                if not traffic_taken or traffic_taken.get("csrc_sha256") != csrc_sha256():
                    traffic_taken = dict(traffic_taken or {}, stale="the PMC table was taken from other kernel sources (csrc hash %s now): "
                                                                    "roofline.traffic withheld; tools/refresh_round.sh renews it" % csrc_sha256())
                    traffic = None
            except Exception:
                traffic = None
        classes = {}
        for i, name in enumerate(F._lib.KC_NAMES):
            if launches[i]:
                classes[name] = dict(launches_per_step=launches[i] / sampled, ms_per_step=round(ms[i] / sampled, 4),
                                     tflops=round((fl[i] / 1e12) / (ms[i] / 1e3), 2) if fl[i] > 0 and ms[i] > 0 else None)
        conv_ms = sum(ms[F._lib.KC_NAMES.index(n)] for n in CONV_CLASSES) / sampled
        out = dict(
            metric="images/sec (%s %dx%d fwd+bwd)" % (args.model, W, H), value=round(world * args.steps / dt, 3), unit="images/sec",
            n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=round(1e3 * dt / args.steps, 3),
            higher_is_better=True, scaling="weak", vs_baseline=None, dtype=dtype_text(F), data="synthetic",
            arithmetic=(("fp32 tensors, fp32 accumulation; the 3x3 / 5x5 / 7x7 convolutions (forward, input gradient, 3x3 weight gradient) scale "
                         "each operand tensor by a power of two chosen from its largest magnitude, split it into two fp16 planes (22 significand "
                         "bits) and form every product from three exact fp16 x fp16 partial products (v_mfma_f32_32x32x16_f16); measured "
                         "error against fp64 at the level of the fp32 matrix-core kernel (tools/x3_f16_check.py); the classification net's "
                         "Linear(13824,1024) forward / input-gradient products " + ("take the same two-plane form" if os.environ.get("FRCNN_GEMM_F16", "1") != "0"
                                                                                   else "use three bf16 planes / six partial products (FRCNN_GEMM_F16=0)") +
                         (", its weight-gradient product two fp16 planes of both operands" if os.environ.get("FRCNN_GEMM_WGRAD_F16", "1") != "0" and os.environ.get("FRCNN_GEMM_F16", "1") != "0"
                          else ", its weight-gradient product three bf16 planes / six partial products") + "; every other product is a plain fp32 product"
                         if nprod == F16_PRODUCTS else
                         "fp32 tensors, fp32 accumulation; the 3x3 convolutions (forward, input gradient, weight gradient) form every fp32 "
                         "product from six exact bf16 x bf16 partial products of three-way split operands (24 significand bits, "
                         "v_mfma_f32_32x32x16_bf16); every other product is a plain fp32 product (v_mfma_f32_32x32x2_f32 / VALU)")
                        if split_on else "fp32 tensors, fp32 products (v_mfma_f32_32x32x2_f32 / VALU), fp32 accumulation"),
            config=dict(workload=args.model + " %dx%d train step: lossAndGradient (pnet fwd, sparse RPN loss, ROI pool, cnet fwd/bwd, "
                                 "ROI-pool bwd, pnet bwd) + gradient all-reduce + rmsprop; config/%s.lua values; frames resident in HBM when the timed "
                                 "region starts (the per-image upload of objective.lua:66 is not in `value`: see with_h2d_upload)"
                                 % (W, H, "duplo" if args.model == "vgg_small" else "imagenet"),
                        with_h2d_upload=upload_leg,
                        images_per_gpu_per_step=1,
                        dropout=dropout_text(F),
                        examples_per_image=[len(b["positive"]) + len(b["negative"]) for b in it.pool], global_batch=world, parallelism="dp%d" % world,
                        conv_gflop_per_image=round(train_flops / 1e9, 2),   # (the dense algorithm: every channel multiplied out)
                        conv_gflop_per_image_as_run=round(sum(fl[F._lib.KC_NAMES.index(n)] for n in CONV_CLASSES) / 1e9 / sampled, 2),
                        whole_step_conv_tflops=round(sum(fl[F._lib.KC_NAMES.index(n)] for n in CONV_CLASSES) / 1e12 / sampled / (dt / args.steps), 2),
                        conv_kernel_ms_per_step=round(conv_ms, 3), kernel_classes=classes,
                        kernel_classes_serial_pass=iso_classes,
                        last_loss=stats["pcls"][-1] + stats["preg"][-1] if stats["pcls"] else None),
            host=dict(enqueue_ms_per_step=round(1e3 * host_t["enqueue"] / max(host_t["steps"], 1), 3),
                      wait_ms_per_step=round(1e3 * host_t["wait"] / max(host_t["steps"], 1), 3),
                      note="rank 0, timed region: time the host needs to queue one step through the C ABI (batch draw, example tables, "
                           "every launch, the update) and the time it then waits for the step's 64-byte statistics; while wait > 0 the "
                           "host is ahead of the device and does not bound `value`"),
            roofline=dict(bound="mfma",
                          kernel=(("conv_x3_kernel (conv forward + input-gradient, two fp16 planes per operand: 3 fp16 MFMA partial products "
                                   "per fp32 product)" if nprod == F16_PRODUCTS else
                                   "conv_x3_kernel (3x3 conv forward + input-gradient, split-bf16 operands: 6 bf16 MFMA partial products "
                                   "per fp32 product)") if split_on else
                                  "conv_igemm_kernel<3,8,*> (3x3 conv forward + input-gradient, fp32 MFMA)"),
                          achieved=round(ach, 2), peak=round(peak, 1), unit="TFLOP/s",
                          frac=round(ach / peak, 4), traffic=traffic,
                          traffic_source=("profiles/pmc_traffic.json: HBM bytes per launch from separate rocprofv3 --pmc FETCH_SIZE / "
                                          "--pmc WRITE_SIZE passes of this command (FETCH x2, the gfx950 correction; tools/pmc_traffic.py), "
                                          "taken when the profile was made -- not counted in this run" if traffic is not None else None),
                          traffic_taken=traffic_taken,
                          peak_note=("algorithmic (fp32-product) TFLOP/s against the dense fp16 / bf16 matrix-core peak 2516.6 / %d partial "
                                     "products; executed MFMA rate = %d x achieved = %.0f TFLOP/s (%.3f of the dense peak); the fp32 matrix-core "
                                     "peak is 157.3 TFLOP/s" % (nprod, nprod, nprod * ach, nprod * ach / BF16_MFMA_PEAK_TFLOPS)
                                     if split_on else "fp32 matrix-core peak"),
                          executed=(dict(partial_products_per_fp32_product=nprod, tflops=round(nprod * ach, 1),
                                         frac_of_dense_16bit_peak=round(nprod * ach / BF16_MFMA_PEAK_TFLOPS, 4),
                                         note="matrix-core FLOPs actually issued (the MFMA-utilisation figure); the two-plane fp16 form "
                                              "issues half of what the three-plane bf16 form does for the same algorithmic work")
                                    if split_on else None),
                          sampled_steps=sampled, launches_per_step=launches[k] / sampled, avg_launch_ms=round(ms[k] / max(launches[k], 1), 4),
                          algorithmic_bytes_per_launch=round(by[k] / max(launches[k], 1)),
                          algorithmic_gflop_per_launch=round(fl[k] / 1e9 / max(launches[k], 1), 3), isolated=iso,
                          dense_equivalent=(dict(gflop_per_step=round(backbone_x3_dense_flops(model, H, W) / 1e9, 2),
                                                 tflops=round(backbone_x3_dense_flops(model, H, W) / 1e12 / (ms[k] / 1e3 / sampled), 2),
                                                 frac=round(backbone_x3_dense_flops(model, H, W) / 1e12 / (ms[k] / 1e3 / sampled) / peak, 4),
                                                 note="the same launches priced by the DENSE algorithm's FLOP formula (SURVEY 8d): what option "
                                                      "drop_compact leaves out -- products with the zeros of nn.SpatialDropout -- counted as if "
                                                      "computed.  Not a matrix-core utilisation: `frac` above counts the launches as run")
                                            if split_on and launches[k] / sampled == 12.0 else None)),
            roofline_hbm=hbm,
            sustained=sustained,
        )
        out["config"]["exchange"] = dict(
            backend=("frcnn_comm (RCCL through the C ABI: frcnn_comm_init_rank_file / frcnn_allreduce_f32 / _f64)" if native_comm is not None
                     else ("torch.distributed nccl (RCCL)" if backend == "nccl" else "torch.distributed %s (debug: ranks may share a device)" % backend)
                     if world > 1 else "none (single process)"),
            ranks=world, buckets=bucket_times, schedule=sched, **(exchange_info or {}))
        ok = True
        full = args.model == "vgg_small" and (H, W) == (FULL_H, FULL_W)
        if world == 1 and not args.no_cpu_baseline and full:
            nat = model["native"]
            w0 = nat.init_parameters(42)
            bn0 = np.concatenate([np.zeros(1024, np.float32), np.ones(1024, np.float32)])
            base, inp, o_loss, o_grad = cpu_baseline(F, cfg, model, w0, bn0)
            out["cpu_baseline"] = base
            g_loss, g_grad, captured = gpu_parity_step(F, model, weights, gradient, w0, bn0, inp)
            rel = float(np.linalg.norm(g_grad.astype(np.float64) - o_grad) / np.linalg.norm(o_grad.astype(np.float64)))
            O_, oracle_model_, oracle_tables_ = _oracle()
            rel_inj, differing = injected_parity(O_, oracle_model_(O_, cfg), oracle_tables_(inp["pos"], inp["neg"], inp["rois"]),
                                                 w0, inp, bn0, captured, g_grad)
            ok = abs(g_loss - o_loss) <= 1e-5 * max(1.0, abs(o_loss)) and rel <= 1e-3 and rel_inj <= 1e-3
            out["parity"] = dict(what="lossAndGradient on the benchmarked frame, same inputs and dropout masks: GPU vs CPU restatement",
                                 loss_gpu=g_loss, loss_oracle=float(o_loss), loss_tolerance=1e-5,
                                 gradient_rel_l2=rel, gradient_tolerance=1e-3,
                                 gradient_rel_l2_injected=rel_inj, decisions_differing=differing,
                                 note="gradient_rel_l2: the oracle takes its own discrete decisions (2x2 pool winners, ROI arg-max, PReLU "
                                      "branches); _injected: the device's decisions are handed to the oracle (tests/decisions.py), what "
                                      "remains is arithmetic; decisions_differing counts the ones the oracle would have taken differently",
                                 examples=inp["R"], ok=bool(ok))
            if not args.no_other_legs:
                out["other_legs"] = dict(inference=inference_leg(F, cfg, model, weights, w0, bn0, True), nms=nms_leg(F, True),
                                         vgg_large=large_leg(F, True),
                                         # the same workload in the two other arithmetic forms (each priced against its own peak)
                                         exact_split=arithmetic_leg(F, "exact_split", dict(x3_f16=0)),
                                         fp32_mfma=arithmetic_leg(F, "fp32_mfma", dict(split_bf16=0)),
                                         # ... and with the dropped channels of nn.SpatialDropout multiplied out like the reference does
                                         dense_dropout=arithmetic_leg(F, "dense_dropout", dict(drop_compact=0)))
                ok = ok and all(r["ids_identical"] for r in out["other_legs"]["nms"]) \
                    and out["other_legs"]["inference"]["parity"]["nms_ids_identical_on_gpu_boxes"] \
                    and out["other_legs"]["vgg_large"]["parity"]["ok"]
        print(json.dumps(out))
        if not ok:
            sys.stderr.write("bench.py: PARITY FAILURE (see the 'parity' / 'other_legs' objects of the JSON line)\n")
            sys.stdout.flush()
            os._exit(3)
    if native_comm is not None:
        native_comm.destroy()
    elif world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
