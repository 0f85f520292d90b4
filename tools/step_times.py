"""Per-step wall time of a configuration's stream_decode loop, with the jump-ahead tasks each step launched (one GPU):
    python tools/step_times.py [--config 2] [--steps 80]
Where the step-time outliers of bench.py's `step_ms` come from."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import libldpc_amd
from libldpc_amd import workloads

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="2")
ap.add_argument("--steps", type=int, default=80)
args = ap.parse_args()
w = workloads.get(args.config)
B = w["batch"]
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
dec = libldpc_amd.HipDecoder(workloads.code_path(w))
out = {"iters": torch.zeros(B, dtype=torch.int32, device=dev), "bit_errors": torch.zeros(B, dtype=torch.int32, device=dev)}
dec.stream_begin(w["channel"], 0, w["x"])
ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
jt = []
ev[0].record()
for i in range(args.steps):
    dec.stream_decode(B, early_term=w["early_term"], iterations=w["iterations"], decoding=w["decoding"], want=(), out=out, stream=stream)
    ev[i + 1].record()
    jt.append(int(dec.jump_tasks))
torch.cuda.synchronize()
prev = 0
for i in range(args.steps):
    print("step %3d  %.3f ms  jump tasks %d" % (i, ev[i].elapsed_time(ev[i + 1]), jt[i] - prev))
    prev = jt[i]
