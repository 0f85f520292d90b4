"""Lean workload driver for the rocprofv3 passes (no torch: the profiler sees only the library's kernels).

usage: python3 tools/pmc_probe.py --config {2,2n,3,4,4n,5,5bec} [--steps 3] [--warmup 2] [--batch B] [--x POINT]
Runs warmup + steps batches of the BASELINE.json configuration through ldpc_hip_stream_decode (outputs: per-frame
iteration counts and bit errors to host buffers) and prints one JSON line {config, frames, iterations_executed,
edge_updates, kernel_ms}."""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
from libldpc_amd import workloads
from libldpc_amd.binding import HipDecoder

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="2")
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--warmup", type=int, default=2)
ap.add_argument("--batch", type=int, default=0)
ap.add_argument("--x", type=float, default=None, help="channel point instead of the configuration's (tools/valu_fit.sh)")
args = ap.parse_args()
w = dict(workloads.get(args.config))
if args.x is not None:
    w["x"] = args.x
B = args.batch or w["batch"]
d = HipDecoder(workloads.code_path(w))
d.set_profiling(True)
d.set_bec_compat(w.get("bec_compat", False))
d.set_fast_mode(int(w.get("fast", 0)))
d.stream_begin(w["channel"], 0, w["x"])
it = np.zeros(B, np.uint32); be = np.zeros(B, np.uint32)
kw = dict(early_term=w["early_term"], iterations=w["iterations"], decoding=w["decoding"], want=(), out={"iters": it, "bit_errors": be})
for _ in range(args.warmup):
    d.stream_decode(B, **kw)
d.last_ms(0)
its = conv = 0
for _ in range(args.steps):
    d.stream_decode(B, **kw)
    its += int(it.sum(dtype=np.uint64)); conv += int((it < w["iterations"]).sum()) if w["early_term"] else 0
print(json.dumps({"config": args.config, "workload": w["name"], "frames": B * args.steps, "iterations_executed": its + conv,
                  "edge_updates": (its + conv) * d.nnz, "kernel_ms": d.last_ms(0), "steps": args.steps}))
