"""The communicator entry points of the C ABI (include/frcnn_hip.h, SURVEY 8b last row / 8e).
CPU: the file rendezvous between two processes (no RCCL involved).  GPU box (one MI355X): a one-rank communicator
over the real librccl -- in-place all-reduces of both dtypes, the weight broadcast, and one whole training step whose
exchange runs through it (every collective an identity) against the step without any exchange."""
import ctypes as C
import multiprocessing as mp
import os
import time

import numpy as np
import pytest


def _rank1(path, q):
    import frcnn_amd as F
    buf = C.create_string_buffer(128)
    F._lib.call("frcnn_comm_exchange_id_file", path.encode(), 1, buf, 20000)
    q.put(buf.raw)


def test_file_rendezvous_between_two_processes(F, tmp_path):
    path = str(tmp_path / "id")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rank1, args=(path, q))
    p.start()
    time.sleep(0.5)                      # rank 1 is polling by now; the file appears atomically
    ident = bytes(range(128))
    F._lib.call("frcnn_comm_exchange_id_file", path.encode(), 0, C.create_string_buffer(ident, 128), 1000)
    got = q.get(timeout=60)
    p.join(60)
    assert got == ident and p.exitcode == 0
    assert not os.path.exists(path + ".tmp")


def test_file_rendezvous_times_out_with_a_message(F, tmp_path):
    t0 = time.time()
    with pytest.raises(F.FrcnnError, match="waited 200 ms"):
        F._lib.call("frcnn_comm_exchange_id_file", str(tmp_path / "never").encode(), 3, C.create_string_buffer(128), 200)
    assert time.time() - t0 < 5
    with pytest.raises(F.FrcnnError):
        F._lib.call("frcnn_comm_init_rank", C.byref(C.c_void_p()), 2, 5, C.create_string_buffer(128))   # rank >= nranks


def _write_id_file(path, nonce, pid_alive, host=None):
    """rank 0's side of the rendezvous in a child process with its own FRCNN_COMM_NONCE; the child exits (dead writer)
    unless pid_alive, in which case it lingers until the file is removed."""
    import subprocess, sys
    code = ("import ctypes as C, os, sys, time; sys.path.insert(0, %r); import frcnn_amd as F; "
            "F._lib.call('frcnn_comm_exchange_id_file', %r.encode(), 0, C.create_string_buffer(bytes([7]) * 128, 128), 1000); "
            "print('written', flush=True); "
            + ("[time.sleep(0.05) for _ in range(600) if os.path.exists(%r)]" % path if pid_alive else "pass")) % (
                os.path.dirname(os.path.dirname(os.path.abspath(__file__))), path)
    env = dict(os.environ, FRCNN_COMM_NONCE=nonce)
    if host:
        env["FRCNN_COMM_HOSTNAME"] = host
    p = subprocess.Popen([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE)
    assert p.stdout.readline().strip() == b"written"
    return p


def test_stale_id_files_are_not_joined(F, tmp_path, monkeypatch):
    """an id file a crashed job left behind (dead writer, or another job's nonce) must not be accepted; a
    live writer of THIS job is; and rank 0 replaces whatever it finds."""
    path = str(tmp_path / "id")
    monkeypatch.setenv("FRCNN_COMM_NONCE", "job-A")
    # 1) same nonce, but the writer is dead (the job crashed and was restarted under the same launcher id)
    p = _write_id_file(path, "job-A", pid_alive=False); p.wait(30)
    assert os.path.exists(path)
    with pytest.raises(F.FrcnnError, match="writer is gone"):
        F._lib.call("frcnn_comm_exchange_id_file", path.encode(), 1, C.create_string_buffer(128), 300)
    # 2) a live writer, another job's nonce
    p = _write_id_file(path, "job-B", pid_alive=True)
    try:
        with pytest.raises(F.FrcnnError, match="another job"):
            F._lib.call("frcnn_comm_exchange_id_file", path.encode(), 1, C.create_string_buffer(128), 300)
    finally:
        os.unlink(path); p.wait(60)
    # 3) a live writer of this job is accepted
    p = _write_id_file(path, "job-A", pid_alive=True)
    try:
        buf = C.create_string_buffer(128)
        F._lib.call("frcnn_comm_exchange_id_file", path.encode(), 1, buf, 2000)
        assert buf.raw == bytes([7]) * 128
    finally:
        os.unlink(path); p.wait(60)
    # 4) rank 0 overwrites a stale file with its own record
    open(path, "wb").write(b"x" * 144)
    F._lib.call("frcnn_comm_exchange_id_file", path.encode(), 0, C.create_string_buffer(bytes(range(128)), 128), 1000)
    buf = C.create_string_buffer(128)
    F._lib.call("frcnn_comm_exchange_id_file", path.encode(), 1, buf, 2000)   # (this process is the live writer)
    assert buf.raw == bytes(range(128)) and os.path.getsize(path) == 208   # id 128 + nonce 8 + pid 8 + host 64


def test_id_file_of_a_writer_on_another_host_is_judged_by_its_nonce(F, tmp_path, monkeypatch):
    """the liveness probe (kill(pid, 0)) only means something on the writer's own host.  A record written under
    another host name -- the file on a shared file system, rank 0 in another container -- is accepted on its nonce although
    no process of that pid exists here; the same record under THIS host's name is a dead job's and is refused."""
    path = str(tmp_path / "id")
    monkeypatch.setenv("FRCNN_COMM_NONCE", "job-A")
    p = _write_id_file(path, "job-A", pid_alive=False, host="node-17"); p.wait(30)      # the writer's pid is gone HERE
    buf = C.create_string_buffer(128)
    F._lib.call("frcnn_comm_exchange_id_file", path.encode(), 1, buf, 300)
    assert buf.raw == bytes([7]) * 128
    monkeypatch.setenv("FRCNN_COMM_HOSTNAME", "node-17")                               # ... and now we ARE that host
    with pytest.raises(F.FrcnnError, match="writer is gone"):
        F._lib.call("frcnn_comm_exchange_id_file", path.encode(), 1, C.create_string_buffer(128), 300)


def test_a_dead_peer_surfaces_as_an_error_within_the_timeout(F, tmp_path, monkeypatch):
    """frcnn_comm_init_rank_timeout / frcnn_comm_init_rank_file: the collective initialisation under a watchdog.  The fault
    knob makes the initialisation hang exactly as ncclCommInitRank does when a peer died after the rendezvous (the real
    thing -- one rank of two missing, real RCCL -- runs on the GPU box below)."""
    monkeypatch.setenv("FRCNN_COMM_FAULT", "hang_init")
    t0 = time.time()
    with pytest.raises(F.FrcnnError, match="did not complete within 400 ms"):
        F._lib.call("frcnn_comm_init_rank_timeout", C.byref(C.c_void_p()), 2, 0, C.create_string_buffer(128), 400)
    assert time.time() - t0 < 5
    t0 = time.time()
    with pytest.raises(F.FrcnnError, match="did not complete within"):
        F._lib.call("frcnn_comm_init_rank_file", C.byref(C.c_void_p()), 2, 0, str(tmp_path / "id").encode(), 500)
    assert time.time() - t0 < 6


@pytest.mark.gpu
def test_a_missing_rank_times_out_on_real_rccl(F, tmp_path):
    """One rank of a two-rank communicator never shows up: real ncclCommInitRank under the watchdog returns an error
    (and the process goes on to build a working one-rank communicator afterwards)."""
    F._lib.call("frcnn_set_device", 0)
    ident = C.create_string_buffer(128)
    F._lib.call("frcnn_comm_get_unique_id", ident)
    t0 = time.time()
    with pytest.raises(F.FrcnnError, match="did not complete within 3000 ms"):
        F._lib.call("frcnn_comm_init_rank_timeout", C.byref(C.c_void_p()), 2, 0, ident, 3000)
    assert time.time() - t0 < 30
    F._lib.call("frcnn_comm_get_unique_id", ident)
    h = C.c_void_p()
    F._lib.call("frcnn_comm_init_rank_timeout", C.byref(h), 1, 0, ident, 60000)
    n = C.c_int()
    F._lib.call("frcnn_comm_query", h, C.byref(n), None, None)
    assert n.value == 1
    F._lib.call("frcnn_comm_destroy", h)


def test_bench_refuses_a_world_it_cannot_run():
    """VERDICT r2 missing #1: `bench.py --gpus N` must never print an N=1 line labelled otherwise.  Without N devices on
    the node it exits non-zero with a message (here: no device at all; on a 1-GPU box: tests/test_gpu_rccl2.py)."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip("this node could run two ranks")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "needs 2 HIP devices" in r.stderr and r.stdout.strip() == ""
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=dict(env, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "mislabelled" in r.stderr and r.stdout.strip() == ""


@pytest.mark.gpu
def test_one_rank_communicator_on_the_gpu(F, tmp_path):
    import torch
    comm = F.Comm(0, 1, path=str(tmp_path / "rdv"))
    try:
        n, r = C.c_int(), C.c_int()
        F._lib.call("frcnn_comm_info", comm.h, C.byref(n), C.byref(r))
        assert (n.value, r.value) == (1, 0)
        assert comm.query() == (1, 0, torch.cuda.current_device())    # ncclCommCount / UserRank / CuDevice
        assert comm.gather_ints(41) == [41]
        with pytest.raises(F.FrcnnError):
            comm.broadcast(torch.zeros(4, dtype=torch.float64, device="cuda"))    # frcnn_broadcast_f32 counts 4-byte elements
        g = torch.randn(26784106, device="cuda")          # the flat gradient of vgg_small / duplo (107 MB)
        want = g.clone()
        w1 = comm.all_reduce(g[1000:5_000_000], async_op=True)    # buckets, as the objective issues them
        w2 = comm.all_reduce(g[5_000_000:], async_op=True)
        w3 = comm.all_reduce(g[:1000], async_op=True)
        acc = torch.arange(8, dtype=torch.float64, device="cuda") * 0.1
        w4 = comm.all_reduce(acc, async_op=True)
        for w in (w1, w2, w3, w4):
            w.wait()
        torch.cuda.synchronize()
        assert torch.equal(g, want)
        assert acc.cpu().tolist() == [k * 0.1 for k in range(8)]
        comm.broadcast(g, 0)
        assert torch.equal(g, want)
        assert comm.gather_max(3.5) == 3.5
        comm.barrier()
        with pytest.raises(F.FrcnnError):
            comm.all_reduce(torch.zeros(4, dtype=torch.int32, device="cuda"))
        with pytest.raises(F.FrcnnError):
            F._lib.call("frcnn_broadcast_f32", comm.h, F.ptr(g), 10, 1, F.stream_ptr())   # root outside the communicator
    finally:
        comm.destroy()


@pytest.mark.gpu
def test_training_step_through_the_native_communicator(F, small_cfg, tmp_path, monkeypatch):
    """lossAndGradient + rmsprop with the exchange step on frcnn_allreduce_* (one rank: sums are identities) equals the
    plain single-process step; the bucket schedule (cnet slice, anchor nets, deep blocks, rest) is the data-parallel one."""
    import torch
    cfg = dict(small_cfg)
    model = F.vgg_small(cfg)
    weights, gradient = F.combine_and_flatten_parameters(model["pnet"], model["cnet"], seed=42)
    w0 = weights.clone()
    nat = model["native"]
    bn0 = nat.bn_running.clone()
    it = F.SyntheticBatchIterator(model, H=128, W=176, images_per_batch=1, pool=1)
    R = len(F.clean_examples(it.pool[0]["positive"], F.output_map_sizes(model, 128, 176))) + \
        len(F.clean_examples(it.pool[0]["negative"], F.output_map_sizes(model, 128, 176)))
    rng = np.random.RandomState(1)
    model["pnet"].drop_masks = [None if l["dropout"] <= 0 else (rng.rand(l["filters"]) > l["dropout"]).astype(np.float32) for l in model["layers"]]
    model["cnet"].drop_masks = [(rng.rand(R, 1024) > 0.5).astype(np.float32), (rng.rand(R, 512) > 0.5).astype(np.float32)]
    res = {}
    comm = F.Comm(0, 1, path=str(tmp_path / "rdv2"))
    calls = []
    orig = comm.all_reduce
    comm.all_reduce = lambda t, async_op=False, group=None: (calls.append((t.dtype, t.numel())), orig(t, async_op, group))[1]
    try:
        for mode in ("native", "single"):
            if mode == "native":
                monkeypatch.setenv("FRCNN_COMM_FORCE", "1")
                F.comm.activate(comm)
            else:
                monkeypatch.delenv("FRCNN_COMM_FORCE")
                F.comm.activate(None)
            stats = dict(pcls=[], preg=[], dcls=[], dreg=[])
            f = F.create_objective(model, weights, gradient, it, stats)
            state = dict(learningRate=1e-4, alpha=0.9)
            _, fx = F.rmsprop(f, weights, state)
            torch.cuda.synchronize()
            res[mode] = (fx[0], gradient.cpu().numpy().copy(), weights.cpu().numpy().copy())
            weights.copy_(w0); nat.bn_running.copy_(bn0)
    finally:
        F.comm.activate(None)
        comm.destroy()
        model["pnet"].drop_masks = None
        model["cnet"].drop_masks = None
    assert abs(res["native"][0] - res["single"][0]) <= 1e-6 * abs(res["single"][0])
    a, b = res["native"][1], res["single"][1]
    assert np.linalg.norm(a - b) <= 1e-6 * np.linalg.norm(b)
    w0h = w0.cpu().numpy()
    assert np.linalg.norm(res["native"][2] - res["single"][2]) <= 1e-3 * np.linalg.norm(res["single"][2] - w0h)
    f32 = sorted(n for d, n in calls if d == torch.float32)
    assert sum(f32) == nat.total_params, "every gradient element is exchanged exactly once"
    assert nat.total_params - nat.pnet_params in f32          # the cnet slice goes first, as one bucket
    assert any(d == torch.float64 and n == 8 for d, n in calls)   # the accumulators of objective.lua:52-58
