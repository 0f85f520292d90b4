"""Throughput of the BASELINE.json configurations other than the headline one (which bench.py measures).
Prints one JSON line per configuration.  usage: python tools/bench_configs.py [--steps 4]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch, libldpc_amd, gen_regular_code

ap = argparse.ArgumentParser(); ap.add_argument("--steps", type=int, default=4); args = ap.parse_args()
H = os.path.join(ROOT, "tests", "golden", "h.txt")
h8k = "/tmp/h8k_bench.txt"; open(h8k, "w").write(gen_regular_code.generate(8192, 3, 6, 1))
dev = torch.device("cuda", 0)

def run(name, pc, chan, x, decoding, early, B, iters=50, compat=False):
    d = libldpc_amd.HipDecoder(pc); d.set_profiling(True); d.set_bec_compat(compat)
    it = torch.zeros(B, dtype=torch.int32, device=dev); be = torch.zeros(B, dtype=torch.int32, device=dev)
    d.stream_begin(chan, 0, x)
    for _ in range(2):
        d.stream_decode(B, early_term=early, iterations=iters, decoding=decoding, want=(), out={"iters": it, "bit_errors": be})
    torch.cuda.synchronize(); t0 = time.perf_counter(); kms = []; its = 0; conv = 0; fe = 0
    for _ in range(args.steps):
        d.stream_decode(B, early_term=early, iterations=iters, decoding=decoding, want=(), out={"iters": it, "bit_errors": be})
        kms.append(d.last_ms(0)); its += int(it.sum()); conv += int((it < iters).sum()) if early else 0; fe += int((be > 0).sum())
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    n = B * args.steps; eu = (its + conv) * d.nnz
    bpe = (32 * d.nnz + 17 * d.nc + (d.nnz if early else 0)) / d.nnz
    k_s = sum(kms) * 1e-3
    print(json.dumps({"config": name, "frames_per_s": n / dt, "edge_updates_per_s": eu / dt, "ms_per_step": dt / args.steps * 1e3,
                      "kernel_ms": sum(kms) / len(kms), "avg_iter": its / n, "fer": fe / n,
                      "algorithmic_GBs_kernel": eu * bpe / k_s / 1e9 if chan != "BEC" else None,
                      "lds_resident": d.lds_resident}))

run("cfg2 h.txt AWGN -4dB BP early-term B=65536", H, "AWGN", -4.0, "BP", True, 65536)
run("cfg2' h.txt AWGN -4dB BP --no-early-term B=65536", H, "AWGN", -4.0, "BP", False, 65536)
run("cfg3 h.txt AWGN -4dB BP_MS --no-early-term B=65536", H, "AWGN", -4.0, "BP_MS", False, 65536)
run("cfg4 (3,6) n=8192 AWGN 2.0dB BP early-term B=8192", h8k, "AWGN", 2.0, "BP", True, 8192)
run("cfg4' (3,6) n=8192 AWGN 2.0dB BP --no-early-term B=8192", h8k, "AWGN", 2.0, "BP", False, 8192)
run("cfg5 h.txt BSC eps=0.24 BP early-term B=65536", H, "BSC", 0.24, "BP", True, 65536)
run("cfg5 h.txt BEC eps=0.7 early-term B=65536 (compat)", H, "BEC", 0.7, "BP", True, 65536, compat=True)
