"""bench.py's launch paths and the shape of its line.  CPU tests cover the argument handling that needs no GPU; the
`-m gpu` tests run the real thing on one MI355X, including a two-rank rehearsal (both ranks on cuda:0, counters
reduced over gloo) that checks the sharded counters against single-process decodes of the same frame ranges."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def run_bench(*args, env=None, check=True):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    p = subprocess.run([sys.executable, BENCH, *args], env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    if check:
        assert p.returncode == 0, p.stderr[-2000:]
    return p


def last_json(p):
    return json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])


def test_gpus_flag_must_match_world_size():
    p = run_bench("--gpus", "2", env={"WORLD_SIZE": "4", "RANK": "0"}, check=False)
    assert p.returncode != 0 and "torch.distributed.run" in p.stderr


def test_every_baseline_config_is_selectable():
    from libldpc_amd import workloads
    assert {"1", "2", "3", "4", "5"} <= set(workloads.WORKLOADS)
    for k in workloads.WORKLOADS:
        w = workloads.get(k)
        args = workloads.ref_args(w, "/tmp/out.txt", 100, 8)
        assert "--channel" in args and args[args.index("--decoding") + 1] == w["decoding"]
        assert ("--no-early-term" in args) == (not w["early_term"])
        assert 37.6 < workloads.algorithmic_bytes_per_edge_update(w) < 38.7  # 37.667 / 38.667


def test_roofline_arithmetic_is_a_fraction():
    """roofline_from on the round-1 headline counters (profiles/r1_pmc.json values): VALU-bound, frac <= 1."""
    sys.path.insert(0, ROOT)
    import bench
    from libldpc_amd import workloads
    c = {"ns": 4.409e6, "GRBM_GUI_ACTIVE": 8 * 8.97e6, "SQ_ACTIVE_INST_VALU": 1.714e9, "SQ_INSTS_VALU": 1.50e9, "SQ_INSTS_LDS": 1.78e8,
         "SQ_ACTIVE_INST_LDS": 3e8, "SQ_LDS_BANK_CONFLICT": 4e7, "SQ_LDS_IDX_ACTIVE": 4e8, "SQ_WAVE_CYCLES": 1e10,
         "FETCH_SIZE": 262144.0, "WRITE_SIZE": 512.0}
    r = bench.roofline_from({"counters": c, "kernel": "k", "probe": {"edge_updates": 3.15e9, "steps": 1}}, 4.4, 3.15e9, workloads.get("2"))
    assert r["bound"] == "valu" and 0.6 < r["frac"] <= 1.0
    assert all(v["frac"] <= 1.0 for v in r["ceilings"].values())
    assert abs(r["ceilings"]["valu"]["busy_frac_of_kernel_cycles"] - 0.746) < 0.01
    assert r["traffic"] == (2 * 262144.0 + 512.0) * 1024 and r["algorithmic_equiv_GBs"] > 8000  # reported, not a fraction
    c.update({"SQ_INSTS_VALU_ADD_F64": 1e8, "SQ_INSTS_VALU_MUL_F64": 3e8, "SQ_INSTS_VALU_FMA_F64": 5e8, "SQ_INSTS_VALU_TRANS_F64": 6e7,
              "SQ_INSTS_VALU_INT32": 4e8, "SQ_INSTS_VALU_INT64": 0, "SQ_INSTS_VALU_CVT": 1e6, "SQ_INSTS_SALU": 1e9})
    r = bench.roofline_from({"counters": c, "kernel": "k", "probe": {"edge_updates": 3.15e9, "steps": 1}}, 4.4, 3.15e9, workloads.get("2"))
    flops = 64 * (1e8 + 3e8 + 2 * 5e8 + 6e7)  # an FMA counts two
    assert r["bound"] == "valu-fp64" and abs(r["achieved"] - flops / 4.409e-3 / 1e12) < 1e-9 and abs(r["peak"] - 78.6432) < 1e-3
    assert r["frac"] < r["valu_busy"] and r["ceilings"]["valu"]["frac"] > 0.6  # the busy figure stays, no longer THE fraction
    r3 = bench.roofline_from({"counters": c, "kernel": "k", "probe": {"edge_updates": 3.15e9, "steps": 1}}, 4.4, 3.15e9, workloads.get("3"))
    assert r3["bound"] in ("valu", "lds", "hbm")  # min-sum retires hardly any binary64 arithmetic: the on-chip ceilings decide
    r0 = bench.roofline_from(None, 4.4, 3.15e9, workloads.get("2"))
    assert r0["frac"] is None and r0["traffic"] is None


@pytest.mark.gpu
def test_bench_line_headline():
    j = last_json(run_bench("--steps", "3", "--warmup", "2", "--no-cpu-baseline"))
    assert j["n_gpus"] == 1 and j["unit"] == "frames/s" and j["config"]["baseline_config"] == "2"
    assert j["config"]["frames_per_step"] == 65536 and j["dtype"] == "f64" and j["vs_baseline"] is None
    r = j["roofline"]
    # work-normalised: binary64 operations retired against the 78.6 TFLOP/s vector peak, the busy share beside it
    assert r["bound"] == "valu-fp64" and r["unit"] == "TFLOP/s" and 0 < r["frac"] < r["valu_busy"] <= 1.0, r
    mix = r["ceilings"]["fp64"]["lane_instructions_per_edge_update"]
    assert 0.3 < r["ceilings"]["fp64"]["fp64_share_of_valu_instructions"] < 0.95
    assert abs(mix["all_valu"] - r["ceilings"]["valu"]["valu_lane_instructions_per_edge_update"]) / mix["all_valu"] < 0.05
    assert r["traffic"] and r["traffic"] > 65536 * 8192  # at least the 8 KB of normals per frame cross HBM
    assert abs(r["profiled_kernel_ms"] - r["kernel_ms_avg"]) / r["kernel_ms_avg"] < 0.25
    assert 1e-4 < j["fer"] < 1e-2 and 12 < j["avg_iter"] < 14


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", ["1", "3", "4", "5", "5bec"])
def test_bench_line_other_configs(cfg):
    extra = ["--steps", "200", "--warmup", "20"] if cfg == "1" else ["--steps", "2", "--warmup", "1"]
    j = last_json(run_bench("--config", cfg, "--no-cpu-baseline", *extra))
    assert j["config"]["baseline_config"] == cfg and j["value"] > 0
    if cfg == "1":
        assert j["latency_us"]["median"] < 2000 and j["config"]["frames_per_step"] == 1
    else:
        assert 0 < j["roofline"]["frac"] <= 1.0 and 0 < j["roofline"]["valu_busy"] <= 1.0, j["roofline"]


@pytest.mark.gpu
def test_bench_two_ranks_rehearsal():
    """`python bench.py --gpus 2` starts two rank processes; both use cuda:0 here (counters reduced over gloo, the
    library's exchange over host shared memory).  The summed counters must equal a single-process decode of the
    frames the timed steps covered."""
    import libldpc_amd
    B, K, W = 4096, 2, 1
    j = last_json(run_bench("--gpus", "2", "--steps", str(K), "--warmup", str(W), "--batch", str(B), "--no-cpu-baseline", "--no-pmc",
                            env={"LDPC_BENCH_ONE_GPU": "1", "LDPC_BENCH_BACKEND": "gloo"}))
    # (a rank's piece is a whole number of 5.6 MB generator chunks, about 536 frames of this code each)
    assert j["n_gpus"] == 2 and abs(j["config"]["frames_per_step"] - 2 * B) < 0.15 * B
    first, end = j["timed_frame_span"]
    assert first >= W * 2 * B * 0.85 and end - first == j["counters"]["frames"]
    dec = libldpc_amd.HipDecoder(os.path.join(ROOT, "tests", "golden", "h.txt"))
    dec.stream_begin("AWGN", 0, -4.0)
    dec.stream_skip(first)
    r = dec.stream_decode(end - first)
    it, be = r["iters"].astype(np.int64), r["bit_errors"].astype(np.int64)
    tot = [end - first, int((be > 0).sum()), int(be.sum()), int(it.sum()), int((it < 50).sum())]
    c = j["counters"]
    assert [c["frames"], c["fec"], c["bec"], c["iters"], c["converged"]] == tot
    # the N-rank line explains itself (round-2 VERDICT): per rank and min / max over ranks
    rk = j["ranks"]
    for key in ("kernel_ms_avg", "rng_ms_avg", "host_in_exchange_ms_avg", "host_wait_noise_ms_avg", "comm_init_s", "frames", "frames_per_step"):
        assert set(rk[key]) == {"min", "max"} and rk[key]["min"] <= rk[key]["max"], key
    assert len(rk["per_rank"]) == 2 and sorted(r["rank"] for r in rk["per_rank"]) == [0, 1]
    assert sum(r["frames"] for r in rk["per_rank"]) == c["frames"]
    assert 0 < j["step_ms"]["min"] <= j["step_ms"]["median_max_over_ranks"] <= j["step_ms"]["max"]
    assert all(r["kernel_ms_avg"] > 0 and r["step_ms"]["median"] > 0 for r in rk["per_rank"])
