"""Kernel time of the n = 8192 code's fixed-iteration decode (BASELINE config 4 with --no-early-term) against the iteration
count: what an iteration costs in each regime (first iterations, saturated, far beyond).  usage: python tools/iter_sweep8k.py"""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests"); sys.path.insert(0, "tools")
import torch, libldpc_amd, gen_regular_code
p = "/tmp/h8k.txt"
open(p, "w").write(gen_regular_code.generate(8192, 3, 6, 1))
dec = libldpc_amd.HipDecoder(p); dec.set_profiling(True)
B = 8192; dev = torch.device("cuda", 0)
it = torch.zeros(B, dtype=torch.int32, device=dev); be = torch.zeros(B, dtype=torch.int32, device=dev)
prev = None
for iters in (0, 5, 10, 15, 20, 25, 30, 40, 50):
    ms = []
    for rep in range(3):
        dec.stream_begin("AWGN", 0, 2.0)
        dec.stream_decode(B, early_term=False, iterations=iters, decoding="BP", want=(), out={"iters": it, "bit_errors": be})
        torch.cuda.synchronize(); ms.append(dec.last_ms(0))
    m = min(ms)
    print(f"iters={iters}: kernel {m:.3f} ms" + (f"  -> {(m - prev[1]) / (iters - prev[0]):.3f} ms per iteration since {prev[0]}" if prev else ""))
    prev = (iters, m)
