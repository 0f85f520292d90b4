"""One-off validation of the rank-7 start of an 8-GPU bench run: seeking to frame 7*110*65536 with the RNG-only
stream_skip must land on the same frames as decoding all the way there."""
import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import numpy as np, torch, libldpc_amd
B = 65536; first = 7 * 110 * B; tail = 2048
dev = torch.device("cuda", 0)
it = torch.zeros(2 * B, dtype=torch.int32, device=dev); be = torch.zeros(2 * B, dtype=torch.int32, device=dev)
a = libldpc_amd.HipDecoder("tests/golden/h.txt")
t0 = time.time()
a.stream_begin("AWGN", 0, -4.0); a.stream_skip(first)
ra = a.stream_decode(tail, want=("iters", "bit_errors"))
t1 = time.time()
b = libldpc_amd.HipDecoder("tests/golden/h.txt")
b.stream_begin("AWGN", 0, -4.0)
done = 0
while done < first:
    n = min(2 * B, first - done)
    b.stream_decode(n, want=(), out={"iters": it, "bit_errors": be})
    done += n
rb = b.stream_decode(tail, want=("iters", "bit_errors"))
t2 = time.time()
print(f"skip path {t1 - t0:.2f} s, decode-through path {t2 - t1:.2f} s, raw draws {a.stream_raw_draws} vs {b.stream_raw_draws}")
assert a.stream_raw_draws == b.stream_raw_draws
assert np.array_equal(ra["iters"], rb["iters"]) and np.array_equal(ra["bit_errors"], rb["bit_errors"])
print("far skip OK:", ra["iters"][:8])
