"""What ONE rank of a sharded job costs as a function of the world size, measured on one GPU.

    python tools/shard_probe.py [--config 2|4] [--worlds 1,2,4,8] [--rank last] [--steps 60] [--out profiles/r3_shard_cost.jsonl]

For every world size W the process plays rank r of W with the echo communicator (include/ldpc_amd.h,
ldpc_hip_comm_create_echo: every rank's slot of the step's all-gather is answered with this rank's own payload, i.e. every
piece holds as many accepted pairs as this one).  Everything a real rank does per step is done — seek to its piece, the one
jump-ahead launch that moves its chunk start states to the next step, the generator over its piece + margin, the slab
table, the decode of its frames, the five counters — except the wire.  Reported per W: ms per step (wall), decode kernel ms
and noise-stream ms (HIP events), host ms waiting for the noise stream's result, jump-ahead tasks per step, frames per step.
Per-rank work that does not depend on W is the claim being checked (round-2 VERDICT: the old state table cost O(W))."""
import argparse, json, os, sys, time
sys.path.insert(0, os.getcwd())
import torch
import libldpc_amd
from libldpc_amd import workloads

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="2")
ap.add_argument("--worlds", default="1,2,4,8")
ap.add_argument("--rank", default="last")
ap.add_argument("--steps", type=int, default=60)
ap.add_argument("--warmup", type=int, default=10)
ap.add_argument("--out", default="")
ap.add_argument("--gen", action="store_true", help="with the generator matrix (tests/golden/g.txt; h.txt only): the encoder's share of a step")
args = ap.parse_args()
w = workloads.get(args.config)
B = w["batch"]
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
lines = []
for W in [int(x) for x in args.worlds.split(",")]:
    r = W - 1 if args.rank == "last" else min(int(args.rank), W - 1)
    dec = libldpc_amd.HipDecoder(workloads.code_path(w), os.path.join("tests", "golden", "g.txt") if args.gen else "")
    dec.set_profiling(True)
    comm = libldpc_amd.Comm(r, W, echo=True)
    cap = dec.shard_capacity(B * W, W)
    out = {"iters": torch.zeros(cap, dtype=torch.int32, device=dev), "bit_errors": torch.zeros(cap, dtype=torch.int32, device=dev)}
    c = torch.zeros(5, dtype=torch.int64, device=dev)
    dec.stream_begin(w["channel"], 0, w["x"])
    frames = 0
    for i in range(args.warmup + args.steps):
        if i == args.warmup:
            torch.cuda.synchronize()
            for k in range(4):
                dec.last_ms(k)
            j0, t0, frames = dec.jump_tasks, time.perf_counter(), 0
        _, s4 = dec.stream_decode_sharded(comm, B * W, early_term=w["early_term"], iterations=w["iterations"], decoding=w["decoding"],
                                          want=(), out=out, stream=stream)
        dec.batch_counters(out["iters"].data_ptr(), out["bit_errors"].data_ptr(), s4[3], w["iterations"], w["early_term"], c.data_ptr(), stream)
        frames += s4[3]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    line = {"config": args.config, "world": W, "rank": r, "steps": args.steps, "ms_per_step": dt / args.steps * 1e3,
            "decode_kernel_ms": dec.last_ms(0), "noise_stream_ms": dec.last_ms(1), "host_wait_noise_ms": dec.last_ms(3),
            "host_in_exchange_ms": dec.last_ms(2), "jump_tasks_per_step": (dec.jump_tasks - j0) / args.steps,
            "frames_per_step_this_rank": frames / args.steps, "frames_per_s_this_rank": frames / dt,
            "generator_matrix": bool(args.gen), "exchange": "echo (one process standing in for the rank; no wire)"}
    print(json.dumps(line), flush=True)
    lines.append(line)
    comm.close()
    del dec
if args.out:
    with open(args.out, "w") as f:
        for l in lines:
            f.write(json.dumps(l) + "\n")
