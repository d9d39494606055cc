"""The library's OWN communicator (csrc/comm.cpp through F.Comm: file rendezvous, initialisation under the watchdog thread,
frcnn_allreduce_f32 / _f64, frcnn_broadcast_f32, settle()) with world size 2 -- on ONE GPU, through a stand-in for librccl
(tests/stub_rccl.cpp, bound with FRCNN_RCCL_LIB; real RCCL refuses two ranks on one device, and no box this project has seen
holds two).  What is exercised is everything above ncclAllReduce: the bucket schedule of the training step on the communicator's
own stream, the device-side divisor, the replicas staying bit-identical, the result equal to the single-process step on the
two-image batch (objective.lua:49,65,189,200) -- and, with STUB_RCCL_INPROGRESS, a communicator whose collectives
answer ncclInProgress (non-blocking behind the caller's back: settle() must poll ncclCommGetAsyncError until it has left that
state).  VERDICT r5 next 8."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUB = os.path.join(ROOT, "tests", "_stub", "librccl_stub.so")


def _build_stub():
    """(the GPU box gets the prebuilt file from __graft_entry__.build(); this is the fallback for a tree that was never built)"""
    if os.path.exists(STUB) and os.path.getmtime(STUB) >= os.path.getmtime(os.path.join(ROOT, "tests", "stub_rccl.cpp")):
        return
    os.makedirs(os.path.dirname(STUB), exist_ok=True)
    subprocess.check_call(["hipcc", "-O2", "-shared", "-fPIC", os.path.join(ROOT, "tests", "stub_rccl.cpp"), "-o", STUB, "-lrt"])


def _worker(rank, world, out_dir, inprogress):
    os.environ["FRCNN_RCCL_LIB"] = STUB
    os.environ["FRCNN_COMM_NONCE"] = "stub-test:%s" % out_dir
    if inprogress:
        os.environ["STUB_RCCL_INPROGRESS"] = str(inprogress)
        os.environ["STUB_RCCL_POLL_LOG"] = os.path.join(out_dir, "polls")
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    from test_gpu_dp import _setup, _step
    F, model, weights, gradient, anchors, images, pm, cms = _setup()
    comm = F.Comm(rank, world, path=os.path.join(out_dir, "id"), timeout_ms=60000)
    try:
        assert comm.query() == (world, rank, torch.cuda.current_device())
        assert comm.gather_ints(10 + rank) == [10 + r for r in range(world)]
        # main.lua:92-98 under data parallelism: rank 1 starts from garbage and receives rank 0's weights
        if rank == 1:
            weights.add_(1.0)
        comm.broadcast(weights, root=0)
        F.comm.activate(comm)
        st = _step(F, model, weights, gradient, [images[rank]], pm, [cms[rank]])
        torch.cuda.synchronize()
        np.save(os.path.join(out_dir, "g%d.npy" % rank), gradient.cpu().numpy())
        np.save(os.path.join(out_dir, "w%d.npy" % rank), weights.cpu().numpy())
        np.save(os.path.join(out_dir, "s%d.npy" % rank), np.array(st))
    finally:
        F.comm.activate(None)
        comm.destroy()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("inprogress", [0, 3])
def test_two_ranks_through_the_native_communicator_equal_single_process(tmp_path, inprogress):
    _build_stub()
    import torch.multiprocessing as mp
    mp.spawn(_worker, args=(2, str(tmp_path), inprogress), nprocs=2, join=True)
    g0, g1 = np.load(tmp_path / "g0.npy"), np.load(tmp_path / "g1.npy")
    w0, w1 = np.load(tmp_path / "w0.npy"), np.load(tmp_path / "w1.npy")
    s0, s1 = np.load(tmp_path / "s0.npy"), np.load(tmp_path / "s1.npy")
    assert np.array_equal(g0, g1) and np.array_equal(w0, w1) and np.array_equal(s0, s1)   # replicas stay identical
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_gpu_dp import _setup, _step
    F, model, weights, gradient, anchors, images, pm, cms = _setup()
    w_init = weights.cpu().numpy().copy()
    st = _step(F, model, weights, gradient, images, pm, cms)
    g = gradient.cpu().numpy(); w = weights.cpu().numpy()
    assert np.allclose(s0, st, rtol=1e-6, atol=0)
    assert np.linalg.norm(g0 - g) <= 1e-5 * np.linalg.norm(g)
    assert np.abs(w - w_init).max() > 0
    assert np.linalg.norm(w0 - w) <= 1e-3 * np.linalg.norm(w - w_init)
    if inprogress:   # every collective answered ncclInProgress: settle() asked ncclCommGetAsyncError (k + 1 times each)
        for r in range(2):
            polls = int(open(str(tmp_path / ("polls.%d" % r))).read())
            assert polls >= 4 * (inprogress + 1), polls
