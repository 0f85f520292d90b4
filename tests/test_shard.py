"""Several ranks on one stream (include/ldpc_amd.h part 3).  CPU: the host shared-memory communicator between real
processes.  GPU: two ranks sharing cuda:0 decode disjoint parts of the same noise stream — every frame keeps the result
it has in a one-rank run, and the sharded simulation returns the one-rank counters and result-file lines."""
import multiprocessing as mp
import os
import subprocess
import sys

import numpy as np
import pytest

import orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _gather_worker(rank, world, name, q):
    sys.path.insert(0, ROOT)
    import libldpc_amd
    c = libldpc_amd.Comm(rank, world, shm_name=name)
    res = []
    for k in range(50):  # alternating payload sizes, back to back: the two-slot reuse rule is exercised
        v = np.arange(1 + k % 4, dtype=np.uint64) + 1000 * rank + k
        res.append(c.all_gather(v))
    q.put((rank, res))


@pytest.mark.parametrize("world", [2, 5])
def test_shm_communicator_all_gather(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    name = f"/ldpc_amd_test_{os.getpid()}_{world}"
    ps = [ctx.Process(target=_gather_worker, args=(r, world, name, q)) for r in range(world)]
    [p.start() for p in ps]
    got = dict(q.get(timeout=120) for _ in range(world))
    [p.join(timeout=60) for p in ps]
    for k in range(50):
        want = np.stack([np.arange(1 + k % 4, dtype=np.uint64) + 1000 * r + k for r in range(world)])
        for r in range(world):
            assert np.array_equal(got[r][k], want), (k, r)


def _shard_worker(rank, world, name, case, q, code_file=None):
    sys.path.insert(0, ROOT)
    import libldpc_amd
    chan, x, seed, target, steps, decoding, gen = case
    dec = libldpc_amd.HipDecoder(code_file or orc.H_TXT, orc.G_TXT if gen else "")
    dec.set_bec_compat(True)
    comm = libldpc_amd.Comm(rank, world, shm_name=name)
    dec.stream_begin(chan, seed, x)
    out = []
    for _ in range(steps):
        bufs, step = dec.stream_decode_sharded(comm, target, decoding=decoding)
        out.append((step, bufs["iters"][:step[3]].copy(), bufs["bit_errors"][:step[3]].copy()))
    q.put((rank, out))


@pytest.mark.gpu
@pytest.mark.parametrize("case", [("AWGN", -4.0, 0, 6000, 3, "BP", False), ("AWGN", -4.5, 5, 1, 2, "BP_MS", False),
                                  ("BSC", 0.24, 2, 5001, 2, "BP", False), ("BEC", 0.8, 1, 3000, 2, "BP", True),
                                  ("AWGN", -4.0, 11, 4000, 2, "BP", True)])
@pytest.mark.parametrize("world", [2, 3])
def test_sharded_frames_keep_their_results(case, world):
    """Every frame of the sharded steps — whichever rank decoded it — has the iteration count and bit-error count
    of the same frame in a one-rank run; the ranks' ranges tile the step without gaps."""
    import libldpc_amd
    chan, x, seed, target, steps, decoding, gen = case
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    name = f"/ldpc_amd_test_{os.getpid()}_s{world}"
    ps = [ctx.Process(target=_shard_worker, args=(r, world, name, case, q)) for r in range(world)]
    [p.start() for p in ps]
    got = dict(q.get(timeout=300) for _ in range(world))
    [p.join(timeout=60) for p in ps]
    total = got[0][-1][0][0] + got[0][-1][0][1]  # frames covered by all steps
    dec = libldpc_amd.HipDecoder(orc.H_TXT, orc.G_TXT if gen else "")
    dec.set_bec_compat(True)
    dec.stream_begin(chan, seed, x)
    ref = dec.stream_decode(total, decoding=decoding)
    pos = 0
    for s in range(steps):
        step0 = got[0][s][0]
        assert step0[0] == pos
        nxt = step0[0]
        for r in range(world):
            step, it, be = got[r][s]
            assert step[:2] == step0[:2] and step[2] == nxt
            assert np.array_equal(it, ref["iters"][step[2]:step[2] + step[3]]), (s, r)
            assert np.array_equal(be, ref["bit_errors"][step[2]:step[2] + step[3]]), (s, r)
            nxt += step[3]
        assert nxt == step0[0] + step0[1]
        pos = nxt
    assert total >= target * steps * 0.7  # (a rank's piece is a whole number of generator chunks: about 536 frames of this code each)


@pytest.mark.gpu
@pytest.mark.parametrize("case", [("AWGN", 2.0, 0, 600, 3, "BP", False), ("AWGN", 1.2, 3, 50, 2, "BP_MS", False),
                                  ("BEC", 0.42, 1, 500, 2, "BP", False), ("BSC", 0.06, 2, 301, 2, "BP", False)])
@pytest.mark.parametrize("world", [2, 3])
def test_sharded_frames_keep_their_results_config4_code(case, world, h8k_file):
    """The same on the code BASELINE.json shards (config 4: (3,6)-regular n=8192, register-resident decoder, one
    workgroup per CU): pieces of three chunks and of one chunk per rank, AWGN / BEC / BSC."""
    import libldpc_amd
    chan, x, seed, target, steps, decoding, gen = case
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    name = f"/ldpc_amd_test_{os.getpid()}_k{world}"
    ps = [ctx.Process(target=_shard_worker, args=(r, world, name, case, q, h8k_file)) for r in range(world)]
    [p.start() for p in ps]
    got = dict(q.get(timeout=300) for _ in range(world))
    [p.join(timeout=60) for p in ps]
    total = got[0][-1][0][0] + got[0][-1][0][1]
    dec = libldpc_amd.HipDecoder(h8k_file)
    dec.stream_begin(chan, seed, x)
    ref = dec.stream_decode(total, decoding=decoding)
    pos = 0
    for s in range(steps):
        step0 = got[0][s][0]
        assert step0[0] == pos
        nxt = step0[0]
        for r in range(world):
            step, it, be = got[r][s]
            assert step[:2] == step0[:2] and step[2] == nxt
            assert np.array_equal(it, ref["iters"][step[2]:step[2] + step[3]]), (s, r)
            assert np.array_equal(be, ref["bit_errors"][step[2]:step[2] + step[3]]), (s, r)
            nxt += step[3]
        assert nxt == step0[0] + step0[1]
        pos = nxt
    assert total >= target * steps * 0.7


def test_sharded_step_failure_reaches_every_rank():
    """A rank that fails inside a sharded step publishes a status word in the step's all-gather: every rank raises from
    the same call instead of waiting in the next collective (round-2 ADVICE).  CPU: the engine is never reached — an
    unusable device makes the step fail on every rank, and the ranks must all return, promptly."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    name = f"/ldpc_amd_test_{os.getpid()}_fail"
    ps = [ctx.Process(target=_failing_worker, args=(r, 2, name, q)) for r in range(2)]
    [p.start() for p in ps]
    got = dict(q.get(timeout=120) for _ in range(2))
    [p.join(timeout=60) for p in ps]
    assert all("sharded step failed" in got[r] or "HIP" in got[r] or "device" in got[r] for r in range(2)), got


def _failing_worker(rank, world, name, q):
    sys.path.insert(0, ROOT)
    import libldpc_amd
    dec = libldpc_amd.HipDecoder(orc.H_TXT, device=0 if rank == 0 else 7)  # rank 1's device does not exist anywhere we test
    comm = libldpc_amd.Comm(rank, world, shm_name=name)
    try:
        dec.stream_begin("AWGN", 0, -4.0)
        dec.stream_decode_sharded(comm, 1000)
        q.put((rank, "no error"))
    except Exception as e:
        q.put((rank, str(e)))


def _failing_sim_worker(rank, world, name, q):
    sys.path.insert(0, ROOT)
    os.environ["LDPC_AMD_COMM_TIMEOUT_S"] = "60"
    import time
    import libldpc_amd
    dec = libldpc_amd.HipDecoder(orc.H_TXT, orc.G_TXT, device=0 if rank == 0 else 7)  # with G: the loop snapshots the encoder
    comm = libldpc_amd.Comm(rank, world, shm_name=name)
    t0 = time.time()
    try:
        dec.simulate("AWGN", [-4.0, -3.9, 1.0], max_frames=2000, fec=5, comm=comm)
        q.put((rank, "no error", time.time() - t0))
    except Exception as e:
        q.put((rank, str(e), time.time() - t0))


def _failing_sim(world=2):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    name = f"/ldpc_amd_test_{os.getpid()}_fsim"
    ps = [ctx.Process(target=_failing_sim_worker, args=(r, world, name, q)) for r in range(world)]
    [p.start() for p in ps]
    got = {}
    for _ in range(world):
        rank, msg, dt = q.get(timeout=180)
        got[rank] = (msg, dt)
    [p.join(timeout=60) for p in ps]
    return got


def test_sharded_simulation_failure_with_generator_reaches_every_rank():
    """Round-3 ADVICE: with a generator matrix loaded the simulation loop snapshots the encoder before the sharded step; a rank
    on which that fails (no usable device) must still take part in the step's own exchange, or the other ranks sit in it
    while this one has moved on to the loop's.  Every rank returns an error from ldpc_hip_simulate_sharded, promptly.  (No
    GPU here: both ranks fail; test_..._one_rank_only below is the asymmetric case on the GPU box.)"""
    got = _failing_sim()
    for r in range(2):
        assert got[r][0] != "no error" and got[r][1] < 30, got


@pytest.mark.gpu
def test_sharded_simulation_failure_with_generator_one_rank_only():
    """The same with rank 0 on the GPU and rank 1 on a device that does not exist: rank 0 learns of the failure in the step's
    exchange and both return an error within seconds."""
    got = _failing_sim()
    assert "rank 1" in got[0][0] and got[0][1] < 30 and got[1][0] != "no error" and got[1][1] < 30, got


def _cli(args, out, extra=()):
    exe = os.path.join(ROOT, "libldpc_amd", "ldpcsim")
    subprocess.check_call([exe, orc.H_TXT, str(out)] + list(args) + list(extra), stdout=subprocess.DEVNULL)
    return [ln.split()[:5] for ln in open(out).read().splitlines()] if os.path.exists(out) else []


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["awgn_ms_sweep", "bsc", "awgn_bp", "bec_G", "awgn_bp_noearly_i10"])
def test_ldpcsim_devices_equals_one_device(golden_sim, tmp_path, name):
    """`ldpcsim --devices 0,0 --comm shm` (two rank processes sharing the GPU) and `--devices 0,0,0`: the result
    file equals the one-process run's, line for line — counters, FER, BER and average iterations included."""
    args = [orc.G_TXT if a == "<G>" else a for a in golden_sim["cli"][name]["args"]] + ["--bec-compat"]
    one = _cli(args, tmp_path / "one.txt")
    assert len(one) >= 2
    for devs in ("0,0", "0,0,0"):
        many = _cli(args, tmp_path / f"many{len(devs)}.txt", ("--devices", devs, "--comm", "shm"))
        assert many == one, (name, devs)


@pytest.mark.gpu
def test_rccl_communicator_single_rank_and_sharded_step():
    """The RCCL transport on the one GPU there is: librccl.so loads, a communicator of one rank initialises,
    ncclAllGather carries the payload; and a sharded step over that communicator decodes the frames a plain
    stream_decode decodes (the code path the multi-GPU runs take, with world = 1)."""
    import libldpc_amd
    comm = libldpc_amd.Comm(0, 1, device=0, unique_id=libldpc_amd.Comm.unique_id())
    v = np.array([7, 2**63 + 5, 0, 123456789], np.uint64)
    assert np.array_equal(comm.all_gather(v), v[None, :])
    dec = libldpc_amd.HipDecoder(orc.H_TXT)
    dec.stream_begin("AWGN", 0, -4.0)
    got_it, got_be = [], []
    for _ in range(3):
        bufs, step = dec.stream_decode_sharded(comm, 5000)
        assert step[0] == sum(len(x) for x in got_it) and step[2] == step[0] and step[3] == step[1]
        got_it.append(bufs["iters"][:step[3]].copy()), got_be.append(bufs["bit_errors"][:step[3]].copy())
    n = sum(len(x) for x in got_it)
    dec.stream_begin("AWGN", 0, -4.0)
    ref = dec.stream_decode(n)
    assert np.array_equal(np.concatenate(got_it), ref["iters"]) and np.array_equal(np.concatenate(got_be), ref["bit_errors"])
    comm.close()


def _rccl_worker(rank, world, id_q, case, q, code_file):
    sys.path.insert(0, ROOT)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.setdefault("LDPC_AMD_COMM_TIMEOUT_S", "120")
    import libldpc_amd
    chan, x, seed, target, steps, decoding = case
    try:
        if rank == 0:
            uid = libldpc_amd.Comm.unique_id()
            for _ in range(world - 1):
                id_q.put(uid)
        else:
            uid = id_q.get(timeout=120)
        dec = libldpc_amd.HipDecoder(code_file, device=rank)
        comm = libldpc_amd.Comm(rank, world, device=rank, unique_id=uid)  # ncclCommInitRank: one rank per GPU
        dec.stream_begin(chan, seed, x)
        out = []
        for _ in range(steps):
            bufs, step = dec.stream_decode_sharded(comm, target, decoding=decoding)
            out.append((step, bufs["iters"][:step[3]].copy(), bufs["bit_errors"][:step[3]].copy()))
        q.put((rank, out, comm.describe(), comm.exchange_stats()))
        comm.close()
    except Exception as e:  # (the parent must not wait for a rank that failed)
        q.put((rank, repr(e), "", {}))


@pytest.mark.gpu
@pytest.mark.parametrize("code", ["h", "8k"])
@pytest.mark.parametrize("world", [2, 4, 8])
def test_sharded_frames_keep_their_results_rccl(world, code, h8k_file):
    """The real thing, the day a node with several GPUs runs the suite: one process per GPU, the sharded step over RCCL
    (ncclCommInitRank with world > 1, ncclAllGather between processes over xGMI), on the n = 1024 code and on the code
    BASELINE.json shards — every frame has the iteration count and bit-error count of the one-rank run, whichever GPU decoded
    it.  Skipped where fewer than `world` GPUs are visible (the one-GPU rehearsals above cover the same code over shm)."""
    import torch
    if torch.cuda.device_count() < world:  # (counting devices does not initialise the GPU)
        pytest.skip(f"needs {world} GPUs, {torch.cuda.device_count()} visible")
    import libldpc_amd
    code_file = orc.H_TXT if code == "h" else h8k_file
    case = ("AWGN", -4.0, 0, 3000 * world, 3, "BP") if code == "h" else ("AWGN", 2.0, 0, 300 * world, 3, "BP")
    ctx = mp.get_context("spawn")  # fresh processes: none inherits an initialised HIP runtime
    q, id_q = ctx.Queue(), ctx.Queue()
    ps = [ctx.Process(target=_rccl_worker, args=(r, world, id_q, case, q, code_file)) for r in range(world)]
    [p.start() for p in ps]
    got = {}
    for _ in range(world):
        rank, out, desc, stats = q.get(timeout=600)
        assert not isinstance(out, str), f"rank {rank}: {out}"
        got[rank] = out
        assert desc.startswith("rccl") and stats["calls"] == case[4]
    [p.join(timeout=120) for p in ps]
    total = got[0][-1][0][0] + got[0][-1][0][1]
    dec = libldpc_amd.HipDecoder(code_file)
    dec.stream_begin(case[0], case[2], case[1])
    ref = dec.stream_decode(total, decoding=case[5])
    pos = 0
    for s in range(case[4]):
        step0 = got[0][s][0]
        assert step0[0] == pos
        nxt = step0[0]
        for r in range(world):
            step, it, be = got[r][s]
            assert step[:2] == step0[:2] and step[2] == nxt
            assert np.array_equal(it, ref["iters"][step[2]:step[2] + step[3]]), (s, r)
            assert np.array_equal(be, ref["bit_errors"][step[2]:step[2] + step[3]]), (s, r)
            nxt += step[3]
        assert nxt == step0[0] + step0[1]
        pos = nxt


def _timeout_worker(rank, name, q):
    sys.path.insert(0, ROOT)
    os.environ["LDPC_AMD_COMM_TIMEOUT_S"] = "2"
    import time
    import libldpc_amd
    comm = libldpc_amd.Comm(rank, 2, shm_name=name)
    comm.all_gather(np.array([rank], np.uint64))
    if rank == 1:
        q.put((rank, "left"))  # this rank leaves the job: it never joins the second exchange
        return
    t0 = time.time()
    try:
        comm.all_gather(np.array([rank], np.uint64))
        q.put((rank, "no error"))
    except RuntimeError as e:
        q.put((rank, f"{time.time() - t0:.1f}s {e}"))


def test_exchange_gives_up_on_a_rank_that_left():
    """No wait in the exchange is unbounded: a rank whose peer has left the job gets an error after the deadline
    (LDPC_AMD_COMM_TIMEOUT_S; RCCL: the same deadline around ncclCommInitRank and the all-gather's stream) instead of
    spinning for ever.  CPU, shared-memory transport."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    name = f"/ldpc_amd_test_{os.getpid()}_to"
    ps = [ctx.Process(target=_timeout_worker, args=(r, name, q)) for r in range(2)]
    [p.start() for p in ps]
    got = dict(q.get(timeout=120) for _ in range(2))
    [p.join(timeout=60) for p in ps]
    assert got[1] == "left" and "did not arrive" in got[0] and float(got[0].split("s ")[0]) < 30, got
