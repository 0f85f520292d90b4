"""The five BASELINE.json configurations as data, shared by bench.py, tools/ and the tests.

Keys follow the reference CLI's flags (src/sim_cpu.cpp:7-22): channel, x (SNR in dB / crossover / erasure
probability), decoding ("BP" | "BP_MS"), iterations, early_term.  `ref_args(w, out, frames, threads)` spells the
same point as an `ldpcsim` command line, for the CPU baseline leg.
"""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
H_TXT = os.path.join(ROOT, "tests", "golden", "h.txt")

_H = "tests/code h.txt (nc=1152,nct=1024,nnz=3456)"
_8K = "(3,6)-regular nc=8192 mc=4096 nnz=24576 (tools/gen_regular_code.py seed 1)"

WORKLOADS = {
    # configs[0]: one frame at a time through the C-ABI decode()/stream path: latency, see bench.py --config 1
    "1": dict(name=f"{_H} AWGN -4 dB BP 50 iters early-term, 1 frame per call, seed 0", code="h", channel="AWGN", x=-4.0,
              decoding="BP", iterations=50, early_term=True, batch=1),
    # configs[1]: the headline
    "2": dict(name=f"{_H} AWGN -4 dB BP 50 iters early-term batch=65536/GPU all-zero codeword seed 0", code="h",
              channel="AWGN", x=-4.0, decoding="BP", iterations=50, early_term=True, batch=65536),
    # configs[1] through the opt-in NON-PARITY fast mode (binary32 messages): never the headline
    "2f": dict(name=f"{_H} AWGN -4 dB BP 50 iters early-term batch=65536/GPU seed 0, NON-PARITY fast mode (binary32 messages)",
               code="h", channel="AWGN", x=-4.0, decoding="BP", iterations=50, early_term=True, batch=65536, fast=1),
    # configs[1] through the opt-in NON-PARITY layered schedule (one wavefront per frame), binary32 / binary16 messages
    "2l": dict(name=f"{_H} AWGN -4 dB BP 50 iters early-term batch=65536/GPU seed 0, NON-PARITY layered schedule (binary32 messages)",
               code="h", channel="AWGN", x=-4.0, decoding="BP", iterations=50, early_term=True, batch=65536, fast=2),
    "2h": dict(name=f"{_H} AWGN -4 dB BP 50 iters early-term batch=65536/GPU seed 0, NON-PARITY layered schedule (binary16 messages)",
               code="h", channel="AWGN", x=-4.0, decoding="BP", iterations=50, early_term=True, batch=65536, fast=3),
    "2n": dict(name=f"{_H} AWGN -4 dB BP 50 iters --no-early-term batch=65536/GPU seed 0", code="h", channel="AWGN",
               x=-4.0, decoding="BP", iterations=50, early_term=False, batch=65536),
    # configs[2]
    "3": dict(name=f"{_H} AWGN -4 dB BP_MS 50 iters --no-early-term batch=65536/GPU seed 0", code="h", channel="AWGN",
              x=-4.0, decoding="BP_MS", iterations=50, early_term=False, batch=65536),
    # configs[3]
    "4": dict(name=f"{_8K} AWGN 2.0 dB BP 50 iters early-term batch=8192/GPU seed 0", code="8k", channel="AWGN", x=2.0,
              decoding="BP", iterations=50, early_term=True, batch=8192),
    "4n": dict(name=f"{_8K} AWGN 2.0 dB BP 50 iters --no-early-term batch=8192/GPU seed 0", code="8k", channel="AWGN",
               x=2.0, decoding="BP", iterations=50, early_term=False, batch=8192),
    # configs[4]: BSC and BEC on h.txt with its 128-bit puncture header
    "5": dict(name=f"{_H} BSC eps=0.24 BP 50 iters early-term batch=65536/GPU seed 0", code="h", channel="BSC", x=0.24,
              decoding="BP", iterations=50, early_term=True, batch=65536),
    "5bec": dict(name=f"{_H} BEC eps=0.7 50 iters early-term batch=65536/GPU seed 0 (reference-compatible degree-1 rule)",
                 code="h", channel="BEC", x=0.7, decoding="BP", iterations=50, early_term=True, batch=65536,
                 bec_compat=True),
}


def get(key):
    key = str(key)
    if key not in WORKLOADS:
        raise KeyError(f"unknown configuration {key!r}: one of {sorted(WORKLOADS)}")
    return dict(WORKLOADS[key], key=key)


def code_path(w):
    """Path of the workload's parity-check file; the n=8192 code is generated on first use (deterministic)."""
    if w["code"] == "h":
        return H_TXT
    path = os.path.join(tempfile.gettempdir(), f"ldpc_amd_h8k_{os.getuid()}.txt")
    if not os.path.exists(path):
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import gen_regular_code
        tmp = f"{path}.{os.getpid()}"
        with open(tmp, "w") as f:
            f.write(gen_regular_code.generate(8192, 3, 6, 1))
        os.replace(tmp, path)
    return path


def code_dims(w):
    return dict(nc=1152, nnz=3456, nct=1024) if w["code"] == "h" else dict(nc=8192, nnz=24576, nct=8192)


def algorithmic_bytes_per_edge_update(w):
    """the reference dataflow moves 32*nnz + 17*nc (+ nnz syndrome reads) fp64-message bytes per
    frame-iteration; both codes have nnz = 3*nc."""
    d = code_dims(w)
    return (32 * d["nnz"] + 17 * d["nc"] + (d["nnz"] if w["early_term"] else 0)) / d["nnz"]


def ref_args(w, out_file, frames, threads):
    """The reference CLI's command line for this channel point (MAX is exclusive: one point)."""
    x = w["x"]
    step = 1.0
    a = [code_path(w), out_file, repr(x), repr(x + (0.01 if w["channel"] == "AWGN" else 1e-4)), repr(step), "-i", str(w["iterations"]),
         "-s", "0", "-t", str(threads), "--channel", w["channel"], "--decoding", w["decoding"], "--max-frames", str(frames),
         "--frame-error-count", str(10**9)]
    if not w["early_term"]:
        a.append("--no-early-term")
    return a
