"""BASELINE config 4 on real hardware, world size 2: two processes on TWO GPUs of one node exchange through RCCL
(the C ABI's communicator: frcnn_comm_init_rank_file / frcnn_allreduce_f32 / _f64, i.e. what a LuaJIT host calls; and
torch.distributed's "nccl" backend), rank r takes image r, and the data-parallel RMSprop step must equal the
single-process step on the two-image batch (objective.lua:49,65,189,197-200).  Skipped on a one-GPU box,
where test_gpu_dp.py runs the same exchange code path over gloo and this file still checks that `bench.py --gpus 2`
refuses to run."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _ndev():
    import torch
    return torch.cuda.device_count() if torch.cuda.is_available() else 0


def _worker(rank, world, port, out_dir, backend, nonce):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), FRCNN_COMM_NONCE=nonce, HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch
    torch.cuda.set_device(rank)
    import frcnn_amd as F
    F._lib.call("frcnn_set_device", rank)
    comm = None
    if backend == "native":
        comm = F.Comm(rank, world, path=os.path.join(out_dir, "rendezvous"))
        F.comm.activate(comm)
        assert comm.query() == (world, rank, rank)
        assert comm.gather_ints(rank * 10 + 1) == [r * 10 + 1 for r in range(world)]
        # a known pattern through both all-reduces and the broadcast
        t = torch.full((1 << 20,), float(rank + 1), device="cuda")
        comm.all_reduce(t)
        d = torch.arange(8, dtype=torch.float64, device="cuda") * (rank + 1)
        comm.all_reduce(d)
        b = torch.full((1000,), float(rank), device="cuda")
        comm.broadcast(b, 1)
        torch.cuda.synchronize()
        s = world * (world + 1) / 2
        assert torch.all(t == s) and d.cpu().tolist() == [k * s for k in range(8)] and torch.all(b == 1.0)
    else:
        import torch.distributed as dist
        dist.init_process_group("nccl", rank=rank, world_size=world)
    from test_gpu_dp import _setup, _step
    F2, model, weights, gradient, anchors, images, pm, cms = _setup()
    st = _step(F2, model, weights, gradient, [images[rank]], pm, [cms[rank]])
    torch.cuda.synchronize()
    np.save(os.path.join(out_dir, "g%d.npy" % rank), gradient.cpu().numpy())
    np.save(os.path.join(out_dir, "w%d.npy" % rank), weights.cpu().numpy())
    np.save(os.path.join(out_dir, "s%d.npy" % rank), np.array(st))
    if comm is not None:
        F.comm.activate(None)
        comm.destroy()
    else:
        import torch.distributed as dist
        dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("backend", ["native", "torch"])
def test_two_gpus_rccl_equal_single_process(tmp_path, backend):
    if _ndev() < 2:
        pytest.skip("needs two HIP devices on this node (found %d)" % _ndev())
    import torch.multiprocessing as mp
    port = 29800 + (os.getpid() % 1000)
    mp.spawn(_worker, args=(2, port, str(tmp_path), backend, "rccl2:%d" % os.getpid()), nprocs=2, join=True)
    g0, g1 = np.load(tmp_path / "g0.npy"), np.load(tmp_path / "g1.npy")
    w0, w1 = np.load(tmp_path / "w0.npy"), np.load(tmp_path / "w1.npy")
    s0, s1 = np.load(tmp_path / "s0.npy"), np.load(tmp_path / "s1.npy")
    assert np.array_equal(g0, g1) and np.array_equal(w0, w1) and np.array_equal(s0, s1)   # replicas stay identical
    from test_gpu_dp import _setup, _step
    F, model, weights, gradient, anchors, images, pm, cms = _setup()
    w_init = weights.cpu().numpy().copy()
    st = _step(F, model, weights, gradient, images, pm, cms)
    g = gradient.cpu().numpy(); w = weights.cpu().numpy()
    assert np.allclose(s0, st, rtol=1e-6, atol=0)
    assert np.linalg.norm(g0 - g) <= 1e-5 * np.linalg.norm(g)
    assert np.linalg.norm(w0 - w) <= 1e-3 * np.linalg.norm(w - w_init)


@pytest.mark.timeout(600)
def test_bench_two_ranks_line_or_loud_failure():
    """`python bench.py --gpus 2` with no launcher: on a node with two devices it must print ONE line with n_gpus 2 and
    the per-rank ncclCommCount; on a one-GPU box it must exit non-zero without printing a line."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=580)
    if _ndev() < 2:
        assert r.returncode != 0 and "needs 2 HIP devices" in r.stderr and r.stdout.strip() == ""
        return
    assert r.returncode == 0, r.stderr[-2000:]
    import json
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["exchange"]["ncclCommCount_per_rank"] == [2, 2]
    assert sorted(out["config"]["exchange"]["device_per_rank"]) == [0, 1]
    assert out["value"] > 0 and out["scaling"] == "weak"
