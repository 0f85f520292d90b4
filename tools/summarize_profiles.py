#!/usr/bin/env python3
"""Distil rocprofv3 outputs (gpurun_out/<dir>) into the committed summaries under profiles/.

usage: python tools/summarize_profiles.py gpurun_out/prof2 r1
writes profiles/<tag>_kernel_stats.csv   (rocprofv3 --kernel-trace --stats summary, our kernels + top others)
       profiles/<tag>_pmc.json           per-kernel averages of the PMC passes (FETCH_SIZE, WRITE_SIZE, SQ_*)
       profiles/<tag>_traffic.json       HBM bytes per launch of the dominant kernel, corrected as
                                         MI355X_MICROARCH.md prescribes (FETCH_SIZE x2 for wide coalesced reads)
       profiles/<tag>_bench.json         the bench.py JSON line of the same build
"""
import csv, glob, json, os, sys

src, tag = sys.argv[1], sys.argv[2]
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
os.makedirs(out, exist_ok=True)


def short(name):
    n = name.replace("(anonymous namespace)::", "")
    n = n.split("(")[0]
    for p in ("void ", "ldpc_amd::"):
        n = n.replace(p, "")
    return n[:90]


# kernel stats
stats = glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    rows = list(csv.DictReader(open(stats[0])))
    with open(os.path.join(out, f"{tag}_kernel_stats.csv"), "w") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows[:16]:
            w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])

pmc = {}
for d in ("fetch", "write", "sq"):
    for f in glob.glob(os.path.join(src, d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "ldpc_amd" not in r["Kernel_Name"]:
                continue
            k = short(r["Kernel_Name"])
            pmc.setdefault(k, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
            pmc[k].setdefault("_ms", []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
summary = {k: {c: sum(v) / len(v) for c, v in cs.items()} | {"_launches": len(cs.get("_ms", []))} for k, cs in pmc.items()}
json.dump(summary, open(os.path.join(out, f"{tag}_pmc.json"), "w"), indent=1)

# dominant kernel = the decode instantiation with the largest summed duration (the ratio-form launch; the
# LLR-domain instantiation that re-decodes escaped frames runs right after it and is nearly empty on this workload)
dec = [k for k in summary if k.startswith("decode_kernel") and "_ms" in summary[k]]
dom = max(dec, key=lambda k: summary[k]["_ms"] * summary[k]["_launches"], default=None)
if dom and "FETCH_SIZE" in summary[dom]:
    fetch_kb, write_kb = summary[dom]["FETCH_SIZE"], summary[dom].get("WRITE_SIZE", 0.0)
    traffic = {"kernel": dom, "FETCH_SIZE_KB": fetch_kb, "WRITE_SIZE_KB": write_kb,
               "hbm_bytes_per_launch": (2 * fetch_kb + write_kb) * 1024,
               "correction": "FETCH_SIZE x2 (gfx950 tallies 128-B requests of a wide coalesced read at 64 B), WRITE_SIZE as is"}
    json.dump(traffic, open(os.path.join(out, f"{tag}_traffic.json"), "w"), indent=1)
    print(traffic)

bj = os.path.join(src, "bench_full.json")
if os.path.exists(bj):
    line = [l for l in open(bj).read().splitlines() if l.startswith("{")][-1]
    json.dump(json.loads(line), open(os.path.join(out, f"{tag}_bench.json"), "w"), indent=1)
print(open(os.path.join(out, f"{tag}_kernel_stats.csv")).read())
