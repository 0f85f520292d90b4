#!/bin/bash
# VALU wave-instructions and duration of EVERY kernel of a step (decode and noise stream), per launch: rocprofv3 --pmc pass of
# tools/pmc_probe.py (through gpurun).   tools/kernel_valu.sh [config] [outfile]
set -o pipefail
c=${1:-2}; out=${2:-gpurun_out/kernel_valu_cfg$c.txt}
export TMPDIR=/tmp
d=$(mktemp -d /tmp/kvalu.XXXX)
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES -d $d -o run --output-format csv -- python3 tools/pmc_probe.py --config $c --steps 4 --warmup 2 > /dev/null 2>&1
python3 - "$d" <<'PY' | tee "$out"
import csv, glob, sys, collections
rows = []
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
per = {}
for r in rows:
    d = per.setdefault((r["Kernel_Name"], r["Dispatch_Id"]), {"ns": int(r["End_Timestamp"]) - int(r["Start_Timestamp"])})
    d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for (k, _), d in per.items():
    name = k.replace("ldpc_amd::(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]
    agg[name]["n"] += 1
    for c, v in d.items():
        agg[name][c] += v
print(f"{'kernel':60s} {'launches':>8s} {'ms/launch':>10s} {'VALU Mwave-instr/launch':>24s} {'SALU':>10s} {'busy share of wave cycles':>10s}")
for name, a in sorted(agg.items(), key=lambda kv: -kv[1]["SQ_INSTS_VALU"]):
    n = a["n"]
    print(f"{name:60s} {int(n):8d} {a['ns'] / n * 1e-6:10.3f} {a['SQ_INSTS_VALU'] / n * 1e-6:24.2f} {a['SQ_INSTS_SALU'] / n * 1e-6:10.2f} {a['SQ_ACTIVE_INST_VALU'] * 4 / max(a['SQ_WAVE_CYCLES'], 1):10.3f}")
PY
rm -rf $d
