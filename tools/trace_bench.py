import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import numpy as np, torch, libldpc_amd
dec = libldpc_amd.HipDecoder("tests/golden/h.txt")
B=65536
dev=torch.device("cuda",0)
it=torch.zeros(B,dtype=torch.int32,device=dev); be=torch.zeros(B,dtype=torch.int32,device=dev)
dec.stream_begin("AWGN",0,-4.0)
for s in range(8):
    torch.cuda.synchronize(); t0=time.perf_counter()
    dec.stream_decode(B, want=(), out={"iters":it,"bit_errors":be})
    t1=time.perf_counter(); torch.cuda.synchronize(); t2=time.perf_counter()
    print(f"step {s}: call {1e3*(t1-t0):.2f} ms, +sync {1e3*(t2-t1):.2f} ms", file=sys.stderr)
