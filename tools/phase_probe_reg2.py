"""Per-wave phase timers of the register-resident kernel (totals form) on BASELINE config 4 (needs the debug build:
LDPC_AMD_PHASE_TRACE_BUILD=1 python -m libldpc_amd.build, then LDPC_AMD_LIB=libldpc_amd/libldpc_trace.so).  Prints mean
cycles per loop pass: work and barrier wait of each of the five phases."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
os.environ["LDPC_AMD_PHASE_TRACE"] = "gpurun_out/phase_reg2.bin"
from libldpc_amd import workloads
from libldpc_amd.binding import HipDecoder
w = workloads.get("4")
d = HipDecoder(workloads.code_path(w))
d.stream_begin("AWGN", 0, 2.0)
r = d.stream_decode(8192)
t = np.fromfile("gpurun_out/phase_reg2.bin", np.uint64)[:256 * 16 * 16].reshape(256, 16, 16).astype(np.float64)
passes = t[:, :, 12] + 2
names = ["gather", "wait(vote)", "CN+scatter0", "wait", "VN0", "wait", "scatter1", "wait", "VN1", "wait"]
per = t[:, :, :10] / passes[:, :, None]
m = per.mean(axis=(0, 1))
for n, v in zip(names, m):
    print("%-12s %8.0f cycles per pass" % (n, v))
print("sum per pass %.0f   whole frame %.0f cycles, passes %.1f, loop share %.2f" % (
    m.sum(), t[:, :, 11].mean(), passes.mean(), (t[:, :, :10].sum(axis=2) / t[:, :, 11]).mean()))
print("slowest / fastest wave per phase (work phases):", [round(float(per[:, :, k].mean(axis=0).max() / per[:, :, k].mean(axis=0).min()), 2) for k in (0, 2, 4, 6, 8)])
