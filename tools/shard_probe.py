"""The sharded step with ONE rank (RCCL communicator of size 1, or host shared memory) against the plain step, on one
GPU: what the N > 1 code path costs before any second GPU is involved (the driver's scaling run measures the rest).
usage: python tools/shard_probe.py [rccl|shm]"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import libldpc_amd
from libldpc_amd import workloads

kind = sys.argv[1] if len(sys.argv) > 1 else "rccl"
w = workloads.get("2")
B, K, W = w["batch"], 60, 10
dev = torch.device("cuda", 0)


def run(sharded):
    dec = libldpc_amd.HipDecoder(workloads.code_path(w))
    comm = None
    if sharded:
        comm = (libldpc_amd.Comm(0, 1, device=0, unique_id=libldpc_amd.Comm.unique_id()) if kind == "rccl"
                else libldpc_amd.Comm(0, 1, shm_name="/ldpc_shard_probe"))
    cap = dec.shard_capacity(B, 1) if sharded else B
    out = {"iters": torch.zeros(cap, dtype=torch.int32, device=dev), "bit_errors": torch.zeros(cap, dtype=torch.int32, device=dev)}
    stream = torch.cuda.current_stream().cuda_stream
    dec.stream_begin(w["channel"], 0, w["x"])
    frames = 0
    for i in range(W + K):
        if i == W:
            torch.cuda.synchronize(); t0 = time.perf_counter(); frames = 0
        if sharded:
            _, s4 = dec.stream_decode_sharded(comm, B, early_term=True, iterations=50, decoding="BP", want=(), out=out, stream=stream)
            frames += s4[3]
        else:
            dec.stream_decode(B, early_term=True, iterations=50, decoding="BP", want=(), out=out, stream=stream)
            frames += B
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return dt / K * 1e3, frames / dt


for rep in range(2):
    a = run(False)
    b = run(True)
    print("plain step %.3f ms (%.4g frames/s)   sharded step, one rank over %s: %.3f ms (%.4g frames/s)" % (a[0], a[1], kind, b[0], b[1]))
