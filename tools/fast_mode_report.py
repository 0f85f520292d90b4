"""What the opt-in NON-PARITY modes cost in error rate and buy in time: the same >= 10^6 frames of the headline stream
(h.txt, AWGN, BP, 50 iterations, early termination, seed 0) through the binary64 path and through mode 1 (flooding,
binary32 messages: kernels_fast.hip), mode 2 (layered schedule, binary32 messages) and mode 3 (layered schedule,
binary16 messages: kernels_layered.hip), at several SNR points.  Prints one JSON line per point: FER / BER / average
iterations (sweeps) of each, the frames on which the verdict differs from the binary64 path's, kernel ms per 65 536-frame batch.
usage: python tools/fast_mode_report.py [--frames 1048576] [--snr -4.5 -4.0 -3.5]"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from libldpc_amd.binding import HipDecoder

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=1 << 20)
ap.add_argument("--snr", type=float, nargs="+", default=[-4.5, -4.0, -3.5])
args = ap.parse_args()
H = os.path.join(ROOT, "tests", "golden", "h.txt")
B = 65536
d = HipDecoder(H)
d.set_profiling(True)
for snr in args.snr:
    res = {}
    be_all = {}
    for mode, code in (("f64", 0), ("fast", 1), ("layered32", 2), ("layered16", 3)):
        d.set_fast_mode(code)
        d.stream_begin("AWGN", 0, snr)
        fe = be = it = 0
        bes = []
        d.last_ms(0)
        for _ in range(args.frames // B):
            r = d.stream_decode(B)
            fe += int((r["bit_errors"] > 0).sum()); be += int(r["bit_errors"].sum()); it += int(r["iters"].sum())
            bes.append(r["bit_errors"] > 0)
        n = args.frames // B * B
        res[mode] = {"fer": fe / n, "ber": be / (n * d.nc), "avg_iter": it / n, "kernel_ms_per_batch": d.last_ms(0)}
        be_all[mode] = np.concatenate(bes)
    print(json.dumps({"snr_dB": snr, "frames": n, **{f"{k}_{m}": v for m, r in res.items() for k, v in r.items()},
                      **{f"frames_with_another_verdict_{m}": int((be_all["f64"] != be_all[m]).sum()) for m in res if m != "f64"},
                      **{f"fer_ratio_{m}_over_f64": res[m]["fer"] / max(res["f64"]["fer"], 1e-30) for m in res if m != "f64"},
                      **{f"kernel_time_ratio_{m}_over_f64": res[m]["kernel_ms_per_batch"] / max(res["f64"]["kernel_ms_per_batch"], 1e-30) for m in res if m != "f64"}}), flush=True)
