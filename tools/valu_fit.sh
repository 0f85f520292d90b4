#!/bin/bash
# The headline kernel's VALU instructions per frame as a + b * (loop passes): SQ_INSTS_VALU of the same kernel at several
# channel points (the frames of a point run different numbers of passes), through gpurun:
#   tools/valu_fit.sh gpurun_out/valu_fit "-5.0 -4.0 -3.0 -1.0"
# prints per point: frames, executed passes per frame, VALU lane-instructions per frame; then the least-squares a, b over the
# first and last point and the prediction error at the points in between (profiles/r3_isa_budget.md holds b against the routines).
set -o pipefail
out=${1:?output directory}; points=${2:-"-5.0 -4.0 -3.0 -1.0"}
mkdir -p "$out"; export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-$PWD}
: > "$out/points.jsonl"
for x in $points; do
    (cd /tmp && timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU -d "$root/$out/x$x" -o run --output-format csv -- python3 "$root/tools/pmc_probe.py" --config 2 --x "$x" --steps 3 --warmup 2 > "$root/$out/x$x.log" 2>&1)
    python3 - "$out" "$x" <<'PY' >> "$out/points.jsonl"
import csv, glob, json, sys
out, x = sys.argv[1], sys.argv[2]
line = json.loads([l for l in open(f"{out}/x{x}.log") if l.startswith("{")][-1])
f = glob.glob(f"{out}/x{x}/**/*counter_collection.csv", recursive=True)[0]
per = {}
for r in csv.DictReader(open(f)):
    if ("decode_fused_small" in r["Kernel_Name"] or "decode_kernel_w5" in r["Kernel_Name"]) and r["Counter_Name"] == "SQ_INSTS_VALU":
        per[r["Dispatch_Id"]] = per.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
launches = sorted(per.items(), key=lambda kv: int(kv[0]))[-line["steps"]:]
valu = sum(v for _, v in launches) * 64
frames = line["frames"]
print(json.dumps({"x": float(x), "frames": frames, "passes_per_frame": line["iterations_executed"] / frames,
                  "valu_lane_instructions_per_frame": valu / frames}))
PY
    rm -rf "$out/x$x"
done
python3 - "$out" <<'PY'
import json, sys
pts = [json.loads(l) for l in open(sys.argv[1] + "/points.jsonl")]
for p in pts:
    print("x %+.1f dB: %.2f passes per frame, %.0f VALU lane-instructions per frame" % (p["x"], p["passes_per_frame"], p["valu_lane_instructions_per_frame"]))
p0, p1 = pts[0], pts[-1]
b = (p0["valu_lane_instructions_per_frame"] - p1["valu_lane_instructions_per_frame"]) / (p0["passes_per_frame"] - p1["passes_per_frame"])
a = p0["valu_lane_instructions_per_frame"] - b * p0["passes_per_frame"]
print("fit over the outer points: a = %.0f lane-instructions per frame (prologue + epilogue), b = %.0f per loop pass = %.2f per edge of h.txt" % (a, b, b / 3456))
for p in pts[1:-1]:
    pred = a + b * p["passes_per_frame"]
    print("  x %+.1f dB: predicted %.0f, measured %.0f (%+.2f %%)" % (p["x"], pred, p["valu_lane_instructions_per_frame"], 100 * (pred / p["valu_lane_instructions_per_frame"] - 1)))
PY
