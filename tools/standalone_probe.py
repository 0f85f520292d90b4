"""Decode kernel alone (no noise-stream kernels running underneath) vs the pipelined step: how much the overlap costs."""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import torch, libldpc_amd
dec = libldpc_amd.HipDecoder("tests/golden/h.txt"); dec.set_profiling(True)
B = 65536; dev = torch.device("cuda", 0)
it = torch.zeros(B, dtype=torch.int32, device=dev); be = torch.zeros(B, dtype=torch.int32, device=dev)
out = {"iters": it, "bit_errors": be}
ms = []
for rep in range(6):
    dec.stream_begin("AWGN", 0, -4.0)
    dec.stream_decode(B, early_term=True, iterations=50, decoding="BP", want=(), out=out)
    torch.cuda.synchronize(); ms.append(dec.last_ms(0))
print("standalone decode kernel ms:", [round(m, 3) for m in ms])
dec.stream_begin("AWGN", 0, -4.0)
for rep in range(30):
    dec.stream_decode(B, early_term=True, iterations=50, decoding="BP", want=(), out=out)
torch.cuda.synchronize()
print("pipelined mean decode kernel ms:", round(dec.last_ms(0), 3), " noise stream ms:", round(dec.last_ms(1), 3))
