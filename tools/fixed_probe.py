import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import torch, libldpc_amd
dec = libldpc_amd.HipDecoder("tests/golden/h.txt"); dec.set_profiling(True)
B = 65536; dev = torch.device("cuda", 0)
it = torch.zeros(B, dtype=torch.int32, device=dev); be = torch.zeros(B, dtype=torch.int32, device=dev)
for name, out in (("iters+be", {"iters": it, "bit_errors": be}), ("iters only", {"iters": it}), ("none", {})):
    for iters in (0, 1, 2):
        ms = []
        for rep in range(3):
            dec.stream_begin("AWGN", 0, -4.0)
            dec.stream_decode(B, early_term=True, iterations=iters, decoding="BP", want=(), out=out)
            torch.cuda.synchronize(); ms.append(dec.last_ms(0))
        print(f"{name} iters={iters}: kernel {min(ms):.3f} ms -> {min(ms)*1e6/B:.1f} ns/frame")
