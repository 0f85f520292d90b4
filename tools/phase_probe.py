"""Per-wave phase timers of the likelihood-ratio decode loop (needs a debug build:
LDPC_AMD_PHASE_TRACE_BUILD=1 python -m libldpc_amd.build).  Prints mean cycles per loop pass and wave for: CN pass,
wait at the vote barrier, VN pass, wait at the second barrier."""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import numpy as np, torch, libldpc_amd
os.environ["LDPC_AMD_PHASE_TRACE"] = "gpurun_out/phase.bin"
dec = libldpc_amd.HipDecoder("tests/golden/h.txt")
B = 65536; dev = torch.device("cuda", 0)
it = torch.zeros(B, dtype=torch.int32, device=dev)
dec.stream_begin("AWGN", 0, -4.0)
dec.stream_decode(B, early_term=True, iterations=50, decoding="BP", want=(), out={"iters": it})
torch.cuda.synchronize()
t = np.fromfile("gpurun_out/phase.bin", np.uint64).reshape(2048, 4, 4).astype(np.float64)
itc = it[:2048].cpu().numpy().astype(np.float64) + 2  # loop passes
per = t / itc[:, None, None]
print("mean cycles per loop pass, per wave: [cn, wait1, vn, wait2]")
for w in range(4):
    print(w, np.round(per[:, w, :].mean(axis=0), 0))
print("total per pass (wave 0):", per[:, 0, :].sum(axis=1).mean())
