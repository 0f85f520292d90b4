#!/usr/bin/env python3
"""Distil what tools/collect_round.sh left under gpurun_out/<dir> into the committed summaries under profiles/.

usage: python tools/summarize_profiles.py gpurun_out/r2 r2
writes profiles/<tag>_configs.jsonl      the bench.py line of every BASELINE configuration (roofline + cpu_baseline inside)
       profiles/<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary of the default bench.py command
       profiles/<tag>_pmc.json           per configuration: per-launch averages of the counter passes of the dominant decode
                                         kernel (SQ sets, FETCH_SIZE, WRITE_SIZE) and the ceilings derived from them, the
                                         same arithmetic as bench.py's roofline (so the line can be checked by hand)
       profiles/<tag>_config_stats.csv   per configuration: kernel-trace stats of the probe run (all kernels above 1 %)
       profiles/<tag>_fast_modes.jsonl   tools/fast_mode_report.py
       profiles/<tag>_shard_cost_cfg{2,4}.jsonl   tools/shard_probe.py
"""
import csv, glob, json, os, shutil, sys

src, tag = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(ROOT, "profiles")
os.makedirs(out, exist_ok=True)
CLOCK, N_SIMD, N_CU, HBM = 2.4e9, 1024, 256, 8000.0


def short(name):
    n = name.replace("(anonymous namespace)::", "").split("(")[0]
    for p in ("void ", "ldpc_amd::"):
        n = n.replace(p, "")
    return n[:110]


# bench lines
if os.path.exists(os.path.join(src, "configs.jsonl")):
    lines = [l for l in open(os.path.join(src, "configs.jsonl")) if l.startswith("{")]
    open(os.path.join(out, f"{tag}_configs.jsonl"), "w").writelines(lines)
    for l in lines:
        j = json.loads(l)
        r = j["roofline"]
        print("%-5s %11.4g frames/s %10.3e eu/s  %7.3f ms/step  kernel %s ms  bound %s frac %s" % (
            j["config"]["baseline_config"], j["value"], j["edge_updates_per_s"], j["ms_per_step"],
            None if r.get("kernel_ms_avg") is None else round(r["kernel_ms_avg"], 3), r["bound"], r["frac"] and round(r["frac"], 3)))

# kernel stats of the default bench command
stats = glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    rows = list(csv.DictReader(open(stats[0])))
    with open(os.path.join(out, f"{tag}_kernel_stats.csv"), "w") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows[:16]:
            w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])

# per-configuration counters
pmc, cstats = {}, []
for d in sorted(glob.glob(os.path.join(src, "cfg*"))):
    if not os.path.isdir(d):
        continue
    cfg = os.path.basename(d)[3:]
    per = {}
    for sub in ("sq", "sq2", "mix", "fetch", "write"):
        for f in glob.glob(os.path.join(d, sub, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"]
                if "ldpc_amd" not in k or not ("decode" in k or "bec_kernel" in k or "bec_sliced_kernel" in k):
                    continue
                e = per.setdefault(short(k), {}).setdefault((sub, r["Dispatch_Id"]), {"ns": int(r["End_Timestamp"]) - int(r["Start_Timestamp"])})
                e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    if not per:
        continue
    dom = max(per, key=lambda k: sum(v["ns"] for (s, _), v in per[k].items() if s == "sq"))
    agg = {}
    for (sub, _), v in per[dom].items():
        for c, x in v.items():
            agg.setdefault((sub, c), []).append(x)
    c = {name: sum(v) / len(v) for (sub, name), v in agg.items() if name != "ns"}
    c["ns"] = sum(agg[("sq", "ns")]) / len(agg[("sq", "ns")])
    t, cyc = c["ns"] * 1e-9, c["GRBM_GUI_ACTIVE"] / 8
    probe = None
    for l in open(os.path.join(src, f"cfg{cfg}.sq1.log"), errors="ignore"):
        if l.startswith("{"):
            probe = json.loads(l)
    eu = probe["edge_updates"] / probe["steps"] if probe else None
    hbm = (2 * c.get("FETCH_SIZE", 0) + c.get("WRITE_SIZE", 0)) * 1024
    pmc[cfg] = {"kernel": dom, "launches_averaged": len(agg[("sq", "ns")]), "counters_per_launch": c,
                "shader_clock_GHz": cyc / t / 1e9,
                "valu_busy_frac_of_kernel_cycles": c["SQ_ACTIVE_INST_VALU"] * 4 / (N_SIMD * cyc),
                "valu_frac_of_peak_2.4GHz": c["SQ_ACTIVE_INST_VALU"] * 4 / t / (N_SIMD * CLOCK),
                "valu_lane_instructions_per_edge_update": c["SQ_INSTS_VALU"] * 64 / eu if eu else None,
                "lds_busy_frac_of_kernel_cycles": c["SQ_LDS_IDX_ACTIVE"] / (N_CU * cyc) if "SQ_LDS_IDX_ACTIVE" in c else None,
                "lds_bank_conflict_frac": c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"] if c.get("SQ_LDS_IDX_ACTIVE") else None,
                "wave_wait_frac": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"] if "SQ_WAIT_ANY" in c else None,
                "hbm_bytes_per_launch": hbm, "hbm_GBs": hbm / t / 1e9, "hbm_frac_of_8TBs": hbm / t / 1e9 / HBM,
                "edge_updates_per_launch": eu, "kernel_ms": c["ns"] * 1e-6}
    if "SQ_INSTS_VALU_FMA_F64" in c:  # work-normalised: binary64 operations retired against 78.6 TFLOP/s (bench.py roofline_from)
        add, mul, fma, trans = (c[k] for k in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64"))
        flops = 64.0 * (add + mul + 2 * fma + trans)
        pmc[cfg].update({"fp64_TFLOPs": flops / t / 1e12, "fp64_frac_of_78.6_TFLOPs": flops / t / 1e12 / (N_SIMD * 16 * 2 * CLOCK / 1e12),
                         "fp64_share_of_valu_instructions": (add + mul + fma + trans) / c["SQ_INSTS_VALU"],
                         "lane_instructions_per_edge_update": None if not eu else {
                             "fp64_add": add * 64 / eu, "fp64_mul": mul * 64 / eu, "fp64_fma": fma * 64 / eu, "fp64_reciprocal": trans * 64 / eu,
                             "int32": c["SQ_INSTS_VALU_INT32"] * 64 / eu, "int64": c["SQ_INSTS_VALU_INT64"] * 64 / eu,
                             "convert": c["SQ_INSTS_VALU_CVT"] * 64 / eu, "all_valu": c["SQ_INSTS_VALU"] * 64 / eu}})
    for f in glob.glob(os.path.join(d, "stats", "**", "*kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if float(r["Percentage"]) >= 1.0:
                cstats.append([cfg, short(r["Name"]), r["Calls"], r["AverageNs"], r["Percentage"]])
json.dump(pmc, open(os.path.join(out, f"{tag}_pmc.json"), "w"), indent=1)
with open(os.path.join(out, f"{tag}_config_stats.csv"), "w") as f:
    w = csv.writer(f)
    w.writerow(["config", "kernel", "calls", "AverageNs", "Percentage"])
    w.writerows(cstats)
for cfg, p in pmc.items():
    print("cfg %-5s %-70s %.3f ms  VALU busy %.2f  LDS busy %s (conflicts %s)  wait %s  HBM %.0f GB/s  VALU/edge %s" % (
        cfg, p["kernel"][:70], p["kernel_ms"], p["valu_busy_frac_of_kernel_cycles"],
        p["lds_busy_frac_of_kernel_cycles"] and round(p["lds_busy_frac_of_kernel_cycles"], 2),
        p["lds_bank_conflict_frac"] and round(p["lds_bank_conflict_frac"], 2), p["wave_wait_frac"] and round(p["wave_wait_frac"], 2),
        p["hbm_GBs"], p["valu_lane_instructions_per_edge_update"] and round(p["valu_lane_instructions_per_edge_update"], 1)))

fm = os.path.join(src, "fast_mode_report.jsonl")
if os.path.exists(fm):
    shutil.copy(fm, os.path.join(out, f"{tag}_fast_modes.jsonl"))
for name in ("shard_cost_cfg2.jsonl", "shard_cost_cfg4.jsonl", "shard_cost_cfg2_gen.jsonl"):
    if os.path.exists(os.path.join(src, name)):
        shutil.copy(os.path.join(src, name), os.path.join(out, f"{tag}_{name}"))
