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
t = np.fromfile("gpurun_out/phase.bin", np.uint64).reshape(2048, 4, 8).astype(np.float64)
itc = it[30000:32048].cpu().numpy().astype(np.float64) + 2  # loop passes (frames 30000.. of the launch: steady state)
per = t[:, :, :4] / itc[:, None, None]
print("mean cycles per loop pass, per wave: [cn, wait1, vn, wait2]")
for w in range(4):
    print(w, np.round(per[:, w, :].mean(axis=0), 0))
print("total per pass (wave 0):", per[:, 0, :].sum(axis=1).mean())
raw = np.fromfile("gpurun_out/phase.bin", np.uint64).reshape(2048, 4, 8)
pro, loop = t[:, 0, 4], t[:, 0, 5]
chan = (raw[:, :, 6] >> np.uint64(32)).astype(np.float64); chan_wait = (raw[:, :, 6] & np.uint64(0xFFFFFFFF)).astype(np.float64)
print("per frame (wave 0), cycles: prologue %.0f (channel init %.0f + barrier wait %.0f, the rest: LLR / slot-index pick-up and v2c"
      " init)  loop %.0f  (mean loop passes %.1f)" % (pro.mean(), chan[:, 0].mean(), chan_wait[:, 0].mean(), loop.mean(), itc.mean()))
print("channel init per wave:", np.round(chan.mean(axis=0)), " barrier wait per wave:", np.round(chan_wait.mean(axis=0)))
