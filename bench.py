#!/usr/bin/env python3
"""bench.py — the LDPC BP hot path on MI355X, one BASELINE.json configuration per run.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config {1,2,2n,2f,3,4,4n,5,5bec}]

One "step" = one pass of the hot path over one batch of synthetic frames of the reference's noise stream
mt19937_64(seed 0): noise generation, acceptance scan, fused channel + LLR init + flooding BP decode with syndrome
early termination, per-frame iteration / bit-error outputs, the five counters of the batch.  Default = configs[1]
(the headline): tests/code h.txt, AWGN -4 dB, BP, 50 iterations, 65 536 frames per GPU and step.  --config 1 is the
single-frame case: one C-ABI decode() call (src/shared.cpp:47-65) per step, latency reported beside the rate.

N GPUs: one process per GPU.  Under `python -m torch.distributed.run` the ranks come from the environment; a plain
`python bench.py --gpus N` starts the N rank processes itself (before anything in this process touches a GPU) and
relays rank 0's line.  Every step is one range of the single noise stream shared out over the ranks (weak scaling:
N x 65 536 frames per step): each rank generates and scans only its own piece of the raw mt19937_64 stream, one RCCL
all-gather of 24 bytes per rank places it in the pair sequence, and the counters {frames, fec, bec, iters, converged}
are summed on each rank's device and reduced ONCE after the last step.  At every N the line carries, per rank and as
min / max over ranks: decode-kernel and noise-stream milliseconds, host time inside the exchange and waiting for the noise
stream, frames per step, communicator set-up seconds (outside the timed region), and the min / median / max step time.

Prints ONE JSON line on rank 0 (contract in the task statement), with
  roofline      of the dominant kernel, from SQ / TCC counters collected by rocprofv3 in separate passes of the same
                workload inside this run (tools/pmc_probe.py).  Sum-product configurations: WORK-NORMALISED — binary64
                operations retired per second against the 78.6 TFLOP/s vector peak (bound "valu-fp64"), with the
                instruction mix per edge-update and the SIMD-busy share (valu_busy) beside it; the other ceilings (LDS
                array, HBM bytes) in `ceilings`.  The survey's algorithmic-bytes figure is reported as
                algorithmic_equiv_GBs and is not a physical bandwidth
  cpu_baseline  the reference CLI (oracle/_ref/ldpcsim_ref) on this box's host cores, bounded sample, with the
                one-thread rate beside it
"""
import argparse
import csv
import glob
import json
import os
import shutil
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from libldpc_amd import workloads  # noqa: E402  (host-only module: no GPU, no torch)

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: 8 TB/s spec
CLOCK_HZ = 2.4e9            # max shader clock
N_SIMD, N_CU = 1024, 256
FP64_PEAK_TFLOPS = N_SIMD * 16 * 2 * CLOCK_HZ / 1e12  # 78.6: 16 fp64 FMA lanes per SIMD and clock (MI355X_MICROARCH.md)
SQ_PASS = ["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_LDS", "SQ_LDS_BANK_CONFLICT",
           "SQ_LDS_IDX_ACTIVE", "SQ_WAVE_CYCLES", "GRBM_GUI_ACTIVE"]
# the instruction mix of the same kernel (its own pass): binary64 arithmetic by class, integer, conversions, scalar
MIX_PASS = ["SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64",
            "SQ_INSTS_VALU_INT32", "SQ_INSTS_VALU_INT64", "SQ_INSTS_VALU_CVT", "SQ_INSTS_SALU"]


# ------------------------------------------------------------------------------------------------------------
# legs that run in child processes before this process touches the GPU
# ------------------------------------------------------------------------------------------------------------
def pmc_passes(cfg, batch, steps=2, warmup=1, passes=("sq", "mix", "fetch", "write")):
    """Per-launch counter averages of the dominant decode kernel of this workload: three separate rocprofv3 --pmc
    passes (SQ set; FETCH_SIZE; WRITE_SIZE — MI355X_MICROARCH.md §rocprofv3 PMC slots) of tools/pmc_probe.py.
    Returns None when rocprofv3 is not usable here."""
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return None
    tmp = tempfile.mkdtemp(prefix="ldpc_pmc_")
    env = dict(os.environ, TMPDIR=tempfile.gettempdir())
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    probe = [sys.executable, os.path.join(ROOT, "tools", "pmc_probe.py"), "--config", cfg, "--steps", str(steps), "--warmup",
             str(warmup), "--batch", str(batch)]
    agg, meta = {}, {}
    try:
        for name, counters in (("sq", SQ_PASS), ("mix", MIX_PASS), ("fetch", ["FETCH_SIZE"]), ("write", ["WRITE_SIZE"])):
            if name not in passes:
                continue
            out = os.path.join(tmp, name)
            p = subprocess.run([rocprof, "--pmc", *counters, "-d", out, "-o", "run", "--output-format", "csv", "--", *probe],
                               cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
            if p.returncode != 0:
                if name == "mix":  # (the mix is additional evidence: without it the line still carries the other ceilings)
                    meta["mix_error"] = f"rocprofv3 pass mix failed (rc {p.returncode}): {p.stderr[-200:]}"
                    continue
                return {"error": f"rocprofv3 pass {name} failed (rc {p.returncode}): {p.stderr[-300:]}"}
            rows = []
            for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                rows += [r for r in csv.DictReader(open(f)) if "ldpc_amd" in r["Kernel_Name"]]
            per = {}
            for r in rows:  # group by (kernel, dispatch): one row per counter
                d = per.setdefault((r["Kernel_Name"], r["Dispatch_Id"]), {"ns": int(r["End_Timestamp"]) - int(r["Start_Timestamp"]),
                                                                          "id": int(r["Dispatch_Id"])})
                d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            kernels = {}
            for (k, _), d in per.items():
                if "decode" in k or "bec_kernel" in k or "bec_sliced_kernel" in k:
                    kernels.setdefault(k, []).append(d)
            if not kernels:
                return {"error": "no decode kernel in the counter output"}
            dom = max(kernels, key=lambda k: sum(d["ns"] for d in kernels[k]))
            timed = sorted(kernels[dom], key=lambda d: d["id"])[-steps:]  # the last `steps` launches (after the warm-up)
            for c in counters:
                agg[c] = sum(d.get(c, 0.0) for d in timed) / len(timed)
            if name == "sq":
                agg["ns"] = sum(d["ns"] for d in timed) / len(timed)
                meta.update({"kernel": dom.replace("ldpc_amd::(anonymous namespace)::", "").replace("void ", "").split("(")[0],
                             "launches_averaged": len(timed)})
            line = [l for l in p.stdout.splitlines() if l.startswith("{")]
            if line and name == "sq":
                meta["probe"] = json.loads(line[-1])
    except Exception as e:  # a profiler problem must not take the benchmark down
        return {"error": f"{type(e).__name__}: {e}"}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return {"counters": agg, **meta}


def roofline_from(pmc, kernel_ms, eu_per_launch, w):
    """Ceilings of the dominant kernel.  kernel_ms = average launch duration from HIP events inside the timed region."""
    bpe = workloads.algorithmic_bytes_per_edge_update(w)
    equiv = eu_per_launch * bpe / (kernel_ms * 1e-3) / 1e9 if kernel_ms else None
    base = {"kernel_ms_avg": kernel_ms, "algorithmic_bytes_per_edge_update": bpe,
            "algorithmic_bytes_per_launch": eu_per_launch * bpe, "algorithmic_equiv_GBs": equiv,
            "algorithmic_equiv_note": "SURVEY §8d reference-dataflow bytes / kernel time; the messages live in LDS / registers, so "
                                      "this is NOT a physical bandwidth and carries no roofline fraction"}
    if not pmc or "counters" not in pmc:
        return {"bound": None, "achieved": None, "peak": None, "unit": None, "frac": None, "traffic": None,
                "counters_error": (pmc or {}).get("error", "rocprofv3 not available"), **base}
    c = pmc["counters"]
    t = c["ns"] * 1e-9                                  # launch duration in the profiled pass
    cycles = c["GRBM_GUI_ACTIVE"] / 8.0                 # rocprofv3 sums the 8 XCDs
    clk = cycles / t
    valu_busy = c["SQ_ACTIVE_INST_VALU"] * 4.0          # quad-cycles -> SIMD-cycles in which a VALU instruction executes
    valu_issue = c["SQ_INSTS_VALU"] * 4.0               # wave-instructions x 4 cycles (16 fp64 lanes/clk/SIMD)
    lds_busy = c["SQ_LDS_IDX_ACTIVE"]                   # LDS-array cycles, conflicts included
    # gfx950: FETCH_SIZE counts 128-B requests at 64 B (None: the byte passes were not run, other_configs)
    hbm_bytes = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0 if "FETCH_SIZE" in c and "WRITE_SIZE" in c else None
    ceilings = {
        "valu": {"achieved": valu_busy / t / 1e9, "peak": N_SIMD * CLOCK_HZ / 1e9, "unit": "G SIMD-busy-cycles/s",
                 "busy_frac_of_kernel_cycles": valu_busy / (N_SIMD * cycles), "issue_frac_of_kernel_cycles": valu_issue / (N_SIMD * cycles),
                 "valu_wave_instructions_per_launch": c["SQ_INSTS_VALU"],
                 "valu_lane_instructions_per_edge_update": c["SQ_INSTS_VALU"] * 64.0 / max(pmc.get("probe", {}).get("edge_updates", 0) / max(pmc.get("probe", {}).get("steps", 1), 1), 1)},
        "lds": {"achieved": lds_busy / t / 1e9, "peak": N_CU * CLOCK_HZ / 1e9, "unit": "G LDS-array-cycles/s",
                "busy_frac_of_kernel_cycles": lds_busy / (N_CU * cycles), "bank_conflict_frac": c["SQ_LDS_BANK_CONFLICT"] / max(lds_busy, 1.0),
                "lds_wave_instructions_per_launch": c["SQ_INSTS_LDS"]},
    }
    if hbm_bytes is not None:
        ceilings["hbm"] = {"achieved": hbm_bytes / t / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s"}
    for v in ceilings.values():
        v["frac"] = v["achieved"] / v["peak"]
    eu = max(pmc.get("probe", {}).get("edge_updates", 0) / max(pmc.get("probe", {}).get("steps", 1), 1), 1)
    mix = None
    if "SQ_INSTS_VALU_FMA_F64" in c:
        # Work-normalised: binary64 operations the kernel retires (an FMA counts two, everything else one; a wave instruction
        # counts 64 lanes) against 78.6 TFLOP/s.  Unlike the busy share above this does NOT rise when the kernel executes
        # redundant instructions.  v_rcp_f64 (TRANS) occupies its SIMD 16 cycles, the others 4: fp64_pipe_frac prices that.
        add, mul, fma, trans = (c[k] for k in MIX_PASS[:4])
        f64 = add + mul + fma + trans
        flops = 64.0 * (add + mul + 2.0 * fma + trans)
        mix = {"achieved": flops / t / 1e12, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": flops / t / 1e12 / FP64_PEAK_TFLOPS,
               "fp64_share_of_valu_instructions": f64 / max(c["SQ_INSTS_VALU"], 1.0),
               "fp64_pipe_frac_of_kernel_cycles": (4.0 * (add + mul + fma) + 16.0 * trans) / (N_SIMD * cycles),
               "lane_instructions_per_edge_update": {"fp64_add": add * 64 / eu, "fp64_mul": mul * 64 / eu, "fp64_fma": fma * 64 / eu,
                                                     "fp64_reciprocal": trans * 64 / eu, "int32": c[MIX_PASS[4]] * 64 / eu,
                                                     "int64": c[MIX_PASS[5]] * 64 / eu, "convert": c[MIX_PASS[6]] * 64 / eu,
                                                     "all_valu": c["SQ_INSTS_VALU"] * 64 / eu, "salu_per_valu": c[MIX_PASS[7]] / max(c["SQ_INSTS_VALU"], 1.0)}}
        ceilings["fp64"] = mix
    sum_product = w["decoding"] == "BP" and w["channel"] != "BEC" and not w.get("fast")
    if mix and sum_product:
        bound, b = "valu-fp64", mix  # sum-product: binary64 arithmetic is the work; the other ceilings stay in `ceilings`
    else:
        bound = max((k for k in ceilings if k != "fp64"), key=lambda k: ceilings[k]["frac"])
        b = ceilings[bound]
    return {"bound": bound, "achieved": b["achieved"], "peak": b["peak"], "unit": b["unit"], "frac": b["frac"],
            "valu_busy": ceilings["valu"]["busy_frac_of_kernel_cycles"],
            "traffic": hbm_bytes, "traffic_source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this run: (2*FETCH_SIZE + WRITE_SIZE) KB",
            "kernel": pmc.get("kernel"), "profiled_kernel_ms": c["ns"] * 1e-6, "shader_clock_GHz": clk / 1e9,
            "mix_error": pmc.get("mix_error"), "ceilings": ceilings, **base}


def cpu_baseline(w, budget_s=18.0):
    """The reference CPU path (src/sim_cpu.cpp) on this box: one thread, then all host cores, bounded samples."""
    cores = os.cpu_count() or 1
    ref = os.path.join(ROOT, "oracle", "_ref", "ldpcsim_ref")
    d = workloads.code_dims(w)
    out = os.path.join(tempfile.gettempdir(), f"ldpc_ref_{os.getpid()}.txt")

    def run_ref(frames, threads):
        t0 = time.time()
        subprocess.run([ref] + workloads.ref_args(w, out, frames, threads), stdout=subprocess.DEVNULL, check=True)
        dt = time.time() - t0
        avg_it = None
        try:  # a result line exists only if a frame error occurred
            line = open(out).read().splitlines()[1].split()
            frames, avg_it = int(line[3]), float(line[4])
        except Exception:
            pass
        return frames / dt, dt, frames, avg_it

    if os.path.exists(ref):
        r0, _, _, _ = run_ref(8, 1)                        # calibration: eight frames on one thread (a fraction of a second)
        r1, dt1, n1, _ = run_ref(max(8, int(r0 * 3.0)), 1)  # one thread, about three seconds
        # the reference's shared counters (omp atomic/critical, ldpcsim.cpp:175-252) stop scaling at about eight threads'
        # worth of work (measured on the 256-core GPU box): size the all-core sample for that
        rN, dtN, nN, avg_it = run_ref(max(64, int(r1 * min(cores, 8) * budget_s)), cores)
        if w["early_term"]:
            eups = rN * (avg_it + 1) * d["nnz"] if avg_it else None
        else:
            eups = rN * w["iterations"] * d["nnz"]
        return {"value": rN, "unit": "frames/s", "edge_updates_per_s": eups, "cores": cores, "kind": "reference",
                "one_thread_frames_per_s": r1,
                "sample": f"oracle/_ref/ldpcsim_ref {' '.join(workloads.ref_args(w, 'OUT', nN, cores)[2:])}: {nN} frames in {dtN:.1f}s "
                          f"on {cores} threads; one thread: {n1} frames in {dt1:.1f}s"}
    import orc  # the C restatement, OpenMP over frames (checked bit for bit against the reference in tests/test_oracle_golden.py)
    code = orc.Code(workloads.code_path(w))
    t0 = time.time()
    code.simulate(w["channel"], [w["x"], w["x"] + 1e-4, 1], threads=1, max_frames=8, min_fec=10**9, iters=w["iterations"],
                  early_term=w["early_term"], min_sum=w["decoding"] == "BP_MS")  # calibration: eight frames on one thread
    guess = 8 / max(time.time() - t0, 1e-3)
    frames = max(64, int(guess * min(cores, 32) * budget_s * 0.5))
    t0 = time.time()
    res = code.simulate(w["channel"], [w["x"], w["x"] + 1e-4, 1], threads=cores, max_frames=frames, min_fec=10**9,
                        iters=w["iterations"], early_term=w["early_term"], min_sum=w["decoding"] == "BP_MS")
    dt = time.time() - t0
    n_frames, _, _, it = (int(v) for v in res["totals"][0])
    return {"value": n_frames / dt, "unit": "frames/s", "edge_updates_per_s": (it + n_frames) * d["nnz"] / dt, "cores": cores,
            "kind": "port", "sample": f"oracle port, {cores} OpenMP threads, {n_frames} frames, wall {dt:.1f}s"}


def other_configs(budget_s=75.0):
    """The other BASELINE.json configurations under the same clock as the headline: each a short child run
    of this script (its own process, before this one touches the GPU): value, ms_per_step, kernel_ms_avg, edge-updates/s
    and roofline bound / frac (counter passes for configuration 4 only: the instruction mix of the register-resident
    kernel; the others report the on-chip ceiling without a fraction).  Keys: BASELINE.json configs[0], [2], [3], [4]
    (BSC and BEC)."""
    t0 = time.time()
    out = {}
    plan = [("1", ["--steps", "400", "--warmup", "50"], False), ("3", ["--steps", "10", "--warmup", "3"], False),
            ("4", ["--steps", "10", "--warmup", "3"], True), ("5", ["--steps", "10", "--warmup", "3"], False),
            ("5bec", ["--steps", "10", "--warmup", "3"], False)]
    env = dict(os.environ, LDPC_BENCH_OTHER="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LDPC_BENCH_CHILD"):
        env.pop(k, None)
    for key, extra, with_mix in plan:
        if time.time() - t0 > budget_s:
            out[key] = {"error": "skipped: time budget of the default run spent"}
            continue
        cmd = [sys.executable, os.path.abspath(__file__), "--config", key, "--no-cpu-baseline", "--no-other-configs", *extra]
        cmd.append("--pmc-passes=sq,mix" if with_mix else "--no-pmc")
        try:
            p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
            j = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
            r = j.get("roofline") or {}
            out[key] = {"workload": j["config"]["workload"], "value": j["value"], "unit": j["unit"], "ms_per_step": j["ms_per_step"],
                        "steps": j["steps"], "kernel_ms_avg": r.get("kernel_ms_avg"), "edge_updates_per_s": j.get("edge_updates_per_s"),
                        "roofline": {"bound": r.get("bound"), "frac": r.get("frac"), "valu_busy": r.get("valu_busy")},
                        "dtype": j["dtype"]}
            if "latency_us" in j:
                out[key]["latency_us"] = j["latency_us"]
        except Exception as e:  # (a failing side measurement must not take the headline line down)
            out[key] = {"error": f"{type(e).__name__}: {str(e)[:200]}"}
    out["seconds"] = time.time() - t0
    return out


# ------------------------------------------------------------------------------------------------------------
# the rank process
# ------------------------------------------------------------------------------------------------------------
def run_latency(args, w):
    """configs[0]: one frame per call through the reference's C ABI (ldpc_setup + decode, host buffers in and out)."""
    import ctypes as ct
    import numpy as np
    import libldpc_amd
    from libldpc_amd.binding import decoder_param
    K, W = args.steps, args.warmup
    dec = libldpc_amd.HipDecoder(workloads.code_path(w))
    dec.stream_begin(w["channel"], 0, w["x"])
    llr = dec.stream_decode(K + W, want=("llr_in", "iters"))  # the frames the reference's channel would hand to decode()
    tx = ~(llr["llr_in"] == 0.0).all(axis=0)                  # transmitted positions (punctured inputs are exactly 0.0)
    lib = ct.CDLL(libldpc_amd.LIB_PATH)
    n, m, nct, mct = (ct.c_int(0) for _ in range(4))
    lib.ldpc_setup(workloads.code_path(w).encode(), b"", ct.byref(n), ct.byref(m), ct.byref(nct), ct.byref(mct))
    assert int(tx.sum()) == nct.value
    frames = np.ascontiguousarray(llr["llr_in"][:, tx])
    p = decoder_param(w["early_term"], w["iterations"], w["decoding"].encode())
    vout = (ct.c_double * nct.value)()
    lib.decode.restype = ct.c_int
    lib.decode.argtypes = [decoder_param, ct.c_void_p, ct.c_void_p]
    its = []
    for f in range(W):
        lib.decode(p, frames[f].ctypes.data, vout)
    lat = np.zeros(K)
    t0 = time.perf_counter()
    for f in range(K):
        ts = time.perf_counter()
        its.append(lib.decode(p, frames[W + f].ctypes.data, vout))
        lat[f] = time.perf_counter() - ts
    dt = time.perf_counter() - t0
    its = np.array(its)
    assert (its == llr["iters"][W:]).all(), "C-ABI decode() and the batch path disagree on iteration counts"
    executed = int((its + (its < w["iterations"])).sum())
    return {"frames": K, "dt": dt, "edge_updates": executed * dec.nnz, "kernel_ms": None,
            "extra": {"latency_us": {"median": float(np.median(lat) * 1e6), "p99": float(np.percentile(lat, 99) * 1e6),
                                     "min": float(lat.min() * 1e6), "mean": float(lat.mean() * 1e6)},
                      "avg_iter": float(its.mean()), "call": "ldpc_setup + decode(decoder_param, llr[nct], llrOut[nct]) of include/ldpc_amd.h"}}


def run_rank(args, w):
    import torch
    import libldpc_amd
    from libldpc_amd import shard

    rank, local_rank, world = shard.rank_world()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # rehearsal on a one-GPU box (tests only): LDPC_BENCH_ONE_GPU=1 puts every rank on cuda:0 and
    # LDPC_BENCH_BACKEND=gloo carries the counter reduce over gloo; the driver's runs use neither
    if os.environ.get("LDPC_BENCH_ONE_GPU"):
        local_rank = 0
    backend = os.environ.get("LDPC_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    if w["key"] == "1":
        if world > 1:
            raise SystemExit("--config 1 is the single-frame latency case: one GPU")
        return run_latency(args, w), rank, world, dist

    early, iters = w["early_term"], w["iterations"]
    B, K, W = args.batch or w["batch"], args.steps, args.warmup
    dec = libldpc_amd.HipDecoder(workloads.code_path(w), device=local_rank)
    dec.set_profiling(True)
    dec.set_bec_compat(w.get("bec_compat", False))
    dec.set_fast_mode(int(w.get("fast", 0)))
    stream = torch.cuda.current_stream().cuda_stream
    dev = torch.device("cuda", local_rank)
    cap = dec.shard_capacity(B * world, world) if world > 1 else B
    iters_d = torch.zeros(cap, dtype=torch.int32, device=dev)
    be_d = torch.zeros(cap, dtype=torch.int32, device=dev)
    out = {"iters": iters_d, "bit_errors": be_d}
    c = torch.zeros(5, dtype=torch.int64, device=dev)

    # N > 1: every step is one range of the single noise stream (seed 0) shared out over the ranks by the library
    # (ldpc_hip_stream_decode_sharded): each rank generates and scans only its own piece of the raw stream, one RCCL
    # all-gather of the accepted-pair counts places it, and each rank decodes about B frames.  Weak scaling: the step
    # grows with N.  The library's communicator is RCCL (its unique id travels over torch.distributed); the rehearsal
    # on one GPU uses the host shared-memory transport, RCCL refuses two ranks on one device.
    comm = None
    comm_init_s = 0.0
    if world > 1:
        t_init = time.perf_counter()
        if os.environ.get("LDPC_BENCH_ONE_GPU"):
            comm = libldpc_amd.Comm(rank, world, shm_name=f"/ldpc_bench_{os.environ.get('MASTER_PORT', '0')}")
        else:
            box = [libldpc_amd.Comm.unique_id() if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            comm = libldpc_amd.Comm(rank, world, device=local_rank, unique_id=box[0])  # ncclCommInitRank: outside the timed region
        comm_init_s = time.perf_counter() - t_init
    dec.stream_begin(w["channel"], 0, w["x"])
    span = [None, 0]  # frames covered by the timed steps: [first, end)
    own = []          # frames this rank decoded per timed step
    tot = torch.zeros(5, dtype=torch.int64, device=dev)

    def step():
        if comm is None:
            dec.stream_decode(B, early_term=early, iterations=iters, decoding=w["decoding"], want=(), out=out, stream=stream)
            n_own, st = B, (dec.stream_frame - B, B)
        else:
            _, s4 = dec.stream_decode_sharded(comm, B * world, early_term=early, iterations=iters, decoding=w["decoding"], want=(),
                                              out=out, stream=stream)
            n_own, st = s4[3], s4[:2]
        if span[0] is None:
            span[0] = st[0]
        span[1] = st[0] + st[1]
        own.append(n_own)
        # {frames, fec, bec, iters, converged} of the rank's frames, summed by the library in one launch on the same stream and
        # added to the rank's running totals on the device: nothing crosses ranks per step but the library's own 24-byte
        # all-gather.  The counters are reduced ONCE, after the last step (north_star: "a single RCCL reduce of FER/BER
        # counters") — and no second communicator is in use while the library's collectives run.
        dec.batch_counters(iters_d.data_ptr(), be_d.data_ptr(), n_own, iters, early, c.data_ptr(), stream)
        tot.add_(c)

    for _ in range(W):  # same ops as the timed loop, so every kernel's code object is loaded beforehand
        step()
    torch.cuda.synchronize()
    for k in range(4):
        dec.last_ms(k)  # drop the warm-up launches' events / timers
    tot.zero_()
    span[0] = None
    own.clear()
    if comm is not None:
        comm.exchange_stats(reset=True)  # (host time inside the all-gathers: the timed steps only)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)]  # per-step spread without a host sync per step
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    marks[0].record()
    for k in range(K):
        step()
        marks[k + 1].record()
    torch.cuda.synchronize()
    exch = comm.exchange_stats() if comm is not None else {"calls": 0, "min": 0.0, "median": 0.0, "max": 0.0}
    transport = comm.describe() if comm is not None else "none (one rank)"
    if comm is not None:  # the one reduce of the counters: 5 x int64 per rank through the library's communicator
        tot = torch.from_numpy(comm.all_gather(tot.cpu().numpy().astype("uint64")).astype("int64").sum(axis=0))
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # HIP events around every decode launch (on the launch stream) and every noise-stream refill (on the engine's
    # internal stream), queued during the loop and read back here: mean over the K timed steps
    kernel_ms, rng_ms = dec.last_ms(0), dec.last_ms(1)
    exch_ms, wait_ms = dec.last_ms(2), dec.last_ms(3)
    step_ms = sorted(marks[k].elapsed_time(marks[k + 1]) for k in range(K))
    mine = {"rank": rank, "kernel_ms_avg": kernel_ms, "rng_ms_avg": rng_ms, "host_in_exchange_ms_avg": exch_ms,
            "host_wait_noise_ms_avg": wait_ms, "comm_init_s": comm_init_s, "frames": int(sum(own)),
            "frames_per_step_min": int(min(own)), "frames_per_step_max": int(max(own)),
            "step_ms": {"min": step_ms[0], "median": step_ms[len(step_ms) // 2], "max": step_ms[-1]},
            "jump_tasks": int(dec.jump_tasks),
            # host microseconds inside Comm::all_gather per timed step (copy in, ncclAllGather, copy out, bounded wait)
            "exchange_us": {"min": exch["min"], "median": exch["median"], "max": exch["max"]}, "transport": transport}
    per_rank = [mine]
    t = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        box = [None] * world
        dist.all_gather_object(box, mine)
        per_rank = box
    dt = float(t.item())
    frames, fec, bec, it_sum, conv = (int(v) for v in tot.tolist())
    d = workloads.code_dims(w)
    # iterations executed = ret+1 for converged frames, ret otherwise
    edge_updates = (it_sum + conv) * d["nnz"]
    if comm is not None:
        comm.close()

    def over(key):
        vals = [r[key] for r in per_rank]
        return {"min": min(vals), "max": max(vals)}
    ranks = {"kernel_ms_avg": over("kernel_ms_avg"), "rng_ms_avg": over("rng_ms_avg"),
             "host_in_exchange_ms_avg": over("host_in_exchange_ms_avg"), "host_wait_noise_ms_avg": over("host_wait_noise_ms_avg"),
             "comm_init_s": over("comm_init_s"), "frames": over("frames"),
             "frames_per_step": {"min": min(r["frames_per_step_min"] for r in per_rank), "max": max(r["frames_per_step_max"] for r in per_rank)},
             "step_ms": {"min": min(r["step_ms"]["min"] for r in per_rank), "median_max_over_ranks": max(r["step_ms"]["median"] for r in per_rank),
                         "max": max(r["step_ms"]["max"] for r in per_rank)},
             "exchange_us": {"min": min(r["exchange_us"]["min"] for r in per_rank), "median_max_over_ranks": max(r["exchange_us"]["median"] for r in per_rank),
                             "max": max(r["exchange_us"]["max"] for r in per_rank)},
             "transport": per_rank[0]["transport"],
             "per_rank": per_rank}
    return ({"frames": frames, "dt": dt, "edge_updates": edge_updates, "kernel_ms": max(r["kernel_ms_avg"] for r in per_rank),
             "extra": {"fer": fec / frames, "ber": bec / (frames * d["nc"]), "avg_iter": it_sum / frames, "rng_ms_avg": rng_ms,
                       "step_ms": ranks["step_ms"], "ranks": ranks,
                       "counters": {"frames": frames, "fec": fec, "bec": bec, "iters": it_sum, "converged": conv},
                       "timed_frame_span": span, "residency": dec.residency,
                       "exchange": "none (one rank)" if world == 1 else ("host shared memory (one-GPU rehearsal)" if os.environ.get("LDPC_BENCH_ONE_GPU")
                                                                        else "RCCL: 24-byte all-gather per step (accepted-pair counts + status), one 5 x int64 gather of the counters after the last step")}},
            rank, world, dist)


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", default="2", choices=sorted(workloads.WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="frames per GPU and step (default: the configuration's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 counter passes (roofline fields become null)")
    ap.add_argument("--pmc-passes", default="sq,mix,fetch,write", help="which counter passes to run (comma-separated)")
    ap.add_argument("--no-other-configs", action="store_true", help="default run only: skip the short runs of the other BASELINE configurations")
    args = ap.parse_args()
    w = workloads.get(args.config)
    if args.steps is None:
        args.steps = 2000 if args.config == "1" else 100
    if args.warmup is None:
        args.warmup = 100 if args.config == "1" else 10

    world_env = os.environ.get("WORLD_SIZE")
    if world_env is not None and int(world_env) != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world_env}; launch with python -m torch.distributed.run "
                         f"--nnodes=1 --nproc-per-node {args.gpus} --master-addr 127.0.0.1 bench.py --gpus {args.gpus} …, or "
                         f"run plain `python bench.py --gpus {args.gpus}` (it starts the ranks itself)")
    rank = int(os.environ.get("RANK", "0"))
    leader = rank == 0 and not os.environ.get("LDPC_BENCH_CHILD")

    # ---- legs that need child processes: before this process (or its rank children) touches a GPU ----
    # (N = 1 only: the counters describe one GPU's kernel, the CPU baseline is a per-box figure; at N > 1 the other ranks
    # would sit in the rendezvous while rank 0 profiles)
    legs = {}
    if leader and args.gpus == 1:
        if not args.no_pmc and args.config != "1":
            legs["pmc"] = pmc_passes(args.config, args.batch or w["batch"], passes=tuple(args.pmc_passes.split(",")))
        if not args.no_cpu_baseline:
            legs["cpu"] = cpu_baseline(w)
        if args.config == "2" and not args.no_other_configs and not args.batch:
            legs["other"] = other_configs()

    if world_env is None and args.gpus > 1:
        # plain `python bench.py --gpus N`: one fresh process per GPU; this parent never initialises the GPU
        port = free_port()
        cmd = [sys.executable, os.path.abspath(__file__), "--gpus", str(args.gpus), "--steps", str(args.steps), "--warmup",
               str(args.warmup), "--config", args.config, "--batch", str(args.batch), "--no-cpu-baseline", "--no-pmc"]
        procs = []
        for r in range(args.gpus):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                       MASTER_PORT=str(port), LDPC_BENCH_CHILD="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
            procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
        out0 = procs[0].communicate()[0]
        rcs = [p.wait() for p in procs]
        if any(rcs):
            sys.stderr.write(out0)
            raise SystemExit(f"bench.py: rank processes exited with {rcs}")
        res = json.loads([l for l in out0.splitlines() if l.startswith("{")][-1])
        finish(res, legs, w)
        print(json.dumps(res))
        return

    m, rank, world, dist = run_rank(args, w)
    if rank == 0:
        d = workloads.code_dims(w)
        K = args.steps
        res = {
            "metric": "decoded frames/s (+ edge-updates/s), 50 BP iters" if args.config != "2" else
                      "decoded frames/s (n=1024 code, 50 BP iters, AWGN)",
            "value": m["frames"] / m["dt"],
            "unit": "frames/s",
            "edge_updates_per_s": m["edge_updates"] / m["dt"],
            "n_gpus": world,
            "steps": K,
            "warmup": args.warmup,
            "ms_per_step": m["dt"] / K * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8" if w["channel"] == "BEC" else ({1: "f32 (non-parity fast mode)", 2: "f32 (non-parity layered mode)", 3: "f16 messages / f32 totals (non-parity layered mode)"}[int(w["fast"])] if w.get("fast") else "f64"),
            "data": "synthetic",
            "config": {"workload": w["name"], "baseline_config": args.config, "frames_per_step": m["frames"] // K,
                       "parallelism": f"frame-shard x{world}", "code": d},
            **m["extra"],
            "_kernel_ms": m["kernel_ms"], "_eu_per_launch_rank": m["edge_updates"] / world / K,
        }
        if not os.environ.get("LDPC_BENCH_CHILD"):
            finish(res, legs, w)
        print(json.dumps(res))
    if dist is not None:
        dist.destroy_process_group()


def finish(res, legs, w):
    """Attach roofline and cpu_baseline to rank 0's line."""
    kernel_ms, eu = res.pop("_kernel_ms", None), res.pop("_eu_per_launch_rank", None)
    if w["key"] == "1":
        res["roofline"] = {"bound": "latency", "achieved": None, "peak": None, "unit": None, "frac": None, "traffic": None,
                           "note": "one frame = one workgroup on one of 256 CUs: a latency measurement, not a throughput roofline"}
    else:
        res["roofline"] = roofline_from(legs.get("pmc"), kernel_ms, eu, w)
    if "cpu" in legs:
        res["cpu_baseline"] = legs["cpu"]
    if "other" in legs:
        res["other_configs"] = legs["other"]


if __name__ == "__main__":
    main()
