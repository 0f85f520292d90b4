"""Config 1 (BASELINE.json configs[0]): one frame at a time.  Latency of a single-frame decode through the batch
entry (host buffers in and out, as the reference's C-ABI decode() hands them over) and through the stream entry."""
import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import numpy as np, libldpc_amd
dec = libldpc_amd.HipDecoder("tests/golden/h.txt")
dec.stream_begin("AWGN", 0, -4.0)
llr = dec.stream_decode(64, want=("llr_in",))["llr_in"]
for _ in range(20):
    dec.decode_batch(llr[:1], want=("iters", "hard", "llr_out"))
ts = []
for f in range(64):
    t0 = time.perf_counter(); r = dec.decode_batch(llr[f:f + 1], want=("iters", "hard", "llr_out")); ts.append(time.perf_counter() - t0)
ts = np.array(ts) * 1e6
print(f"decode_batch(n=1, host buffers): median {np.median(ts):.0f} us, min {ts.min():.0f} us, max {ts.max():.0f} us")
dec.stream_begin("AWGN", 0, -4.0)
for _ in range(5):
    dec.stream_decode(1)
ts = []
for f in range(64):
    t0 = time.perf_counter(); dec.stream_decode(1); ts.append(time.perf_counter() - t0)
ts = np.array(ts) * 1e6
print(f"stream_decode(n=1): median {np.median(ts):.0f} us, min {ts.min():.0f} us")
