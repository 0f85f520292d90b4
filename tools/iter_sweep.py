"""Kernel time vs iteration count: separates the per-frame fixed cost (channel + LLR init + v2c init + outputs)
from the per-iteration cost.  usage: python tools/iter_sweep.py [BP|BP_MS]"""
import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import torch, libldpc_amd
dec = libldpc_amd.HipDecoder("tests/golden/h.txt"); dec.set_profiling(True)
B = 65536; dev = torch.device("cuda", 0)
it = torch.zeros(B, dtype=torch.int32, device=dev); be = torch.zeros(B, dtype=torch.int32, device=dev)
mode = sys.argv[1] if len(sys.argv) > 1 else "BP"
for early in (False, True):
    for iters in (0, 1, 2, 4, 8, 16):
        ms = []
        for rep in range(3):
            dec.stream_begin("AWGN", 0, -4.0)
            dec.stream_decode(B, early_term=early, iterations=iters, decoding=mode, want=(), out={"iters": it, "bit_errors": be})
            torch.cuda.synchronize(); ms.append(dec.last_ms(0))
        print(f"{mode} early={early} iters={iters}: kernel {min(ms):.3f} ms  -> {min(ms)*1e6/B:.1f} ns/frame")
