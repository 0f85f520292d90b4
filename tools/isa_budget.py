#!/usr/bin/env python3
"""Instruction budget of the headline kernel's loop from the disassembly (no GPU needed).

    python tools/isa_budget.py [--out profiles/r3_isa_budget.md] [--measured-all-valu X --measured-mix '{"fp64_fma": ..}']

tools/micro/isa_probe.hip wraps each per-node routine the headline kernel (h.txt, sum-product with early termination,
likelihood-ratio form) runs in its loop into a probe kernel of its own; this script compiles it to gfx950 assembly with the
product's flags, counts the instructions between the probe markers by class, and multiplies by how often h.txt's graph
calls each routine per iteration (640 check nodes of degree 3 and 384 of degree 4 in block pairs; 512 leaves and 512
degree-2 variable nodes in block pairs; 128 punctured variable nodes of degree 15 with their slot indices in registers).
What the routines do not account for — loop control, block descriptors, the syndrome / escape vote, and the per-frame
prologue and epilogue spread over the frame's iterations — is the difference to the measured total (rocprofv3
SQ_INSTS_VALU of bench.py's roofline, passed in by the caller)."""
import argparse, json, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++20", "-ffp-contract=off", "-fno-fast-math", "--cuda-device-only", "-S"]

CLASSES = [("fp64_fma", r"v_fma_f64|v_fmac_f64"), ("fp64_mul", r"v_mul_f64"), ("fp64_add", r"v_add_f64"),
           ("fp64_reciprocal", r"v_rcp_f64|v_rsq_f64|v_sqrt_f64"), ("fp64_other", r"v_\w+_f64"),
           ("int_bit", r"v_(and|or|xor|bfe|bfi|lshl|lshr|ashr|perm|alignbit|lshlrev|lshrrev|not|and_or|or3|xor3|lshl_or|lshl_add)\w*"),
           ("int_arith", r"v_(add|sub|subrev|mad|mul|max|min|max3|min3|med3)\w*_(u|i)(16|32|64)\w*|v_add_co\w*|v_sub_co\w*|v_addc\w*"),
           ("compare_select", r"v_cmp\w*|v_cndmask\w*"), ("move", r"v_mov\w*|v_readfirstlane\w*|v_accvgpr\w*"), ("valu_other", r"v_\w+")]


def classify(op):
    op = re.sub(r"_(e32|e64|dpp|sdwa)$", "", op)
    for name, pat in CLASSES:
        if re.fullmatch(pat, op):
            return name
    return None


def probe_counts(asm):
    out, cur, name = {}, None, None
    for line in asm.splitlines():
        m = re.match(r"^(_ZN8ldpc_amd12_GLOBAL__N_1\d+(probe_\w+?)E\w*):", line)
        if m:
            name = m.group(2)
        t = line.strip()
        if "PROBE_BEGIN" in t:
            cur = {"lds": 0, "salu": 0}
            continue
        if "PROBE_END" in t and cur is not None:
            out[name] = cur
            cur = None
            continue
        if cur is None or not t or t.startswith((";", ".", "/")):
            continue
        op = t.split()[0]
        if op.startswith("ds_"):
            cur["lds"] += 1
        elif op.startswith("s_") and not op.startswith(("s_waitcnt", "s_nop")):
            cur["salu"] += 1
        else:
            c = classify(op)
            if c:
                cur[c] = cur.get(c, 0) + 1
    return out


# calls per iteration of one h.txt frame (each probe = one wave-wide call covering 2 x 64 nodes, except vn15: 64 nodes)
H_TXT = {"probe_cn33": (640 - 128) / 128, "probe_cn44": (384 - 128) / 128, "probe_cn43": 128 / 64 / 1,  # see note in the output
         "probe_vn1x2": 512 / 128, "probe_vn2x2": 512 / 128, "probe_vn15": 128 / 64}
NNZ = 3456


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="")
    ap.add_argument("--measured-all-valu", type=float, default=None, help="lane-instructions per edge-update (roofline.ceilings.fp64...all_valu)")
    ap.add_argument("--measured-mix", default="", help="JSON of roofline.ceilings.fp64.lane_instructions_per_edge_update")
    ap.add_argument("--avg-iterations", type=float, default=13.93, help="executed iterations per frame of the workload")
    args = ap.parse_args()
    src = os.path.join(ROOT, "tools", "micro", "isa_probe.hip")
    with tempfile.TemporaryDirectory() as tmp:
        s = os.path.join(tmp, "probe.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", *FLAGS, "-o", s, src], stderr=subprocess.DEVNULL)
        asm = open(s).read()
    counts = probe_counts(asm)
    valu_classes = [c for c, _ in CLASSES]
    lines = []
    w = lines.append
    w("# Headline loop: instructions per node routine (from the gfx950 disassembly) and per edge-update")
    w("")
    w("`python tools/isa_budget.py` — every routine of the likelihood-ratio loop wrapped in a probe kernel")
    w("(`tools/micro/isa_probe.hip`), compiled with the product's flags, counted between markers.  One call = one wave-wide")
    w("instruction stream covering two blocks of 64 nodes (one block for the degree-15 nodes).")
    w("")
    w("| routine | nodes per call | " + " | ".join(valu_classes) + " | VALU total | LDS | SALU |")
    w("|---|---|" + "---|" * (len(valu_classes) + 3))
    nodes = {"probe_cn33": "2x64 check nodes, degree 3 (shared reciprocal)", "probe_cn44": "2x64 check nodes, degree 4 (shared reciprocal)",
             "probe_cn43": "64 + 64 check nodes, degrees 4 and 3", "probe_cn33_separate": "round 2: degree 3, one quotient per output",
             "probe_cn44_separate": "round 2: degree 4, one quotient per output", "probe_vn1x2": "2x64 leaves (degree-1 variable nodes)",
             "probe_vn2x2": "2x64 variable nodes of degree 2", "probe_vn15": "64 variable nodes of degree 15 (indices in registers)"}
    tot = {}
    for name in nodes:
        c = counts.get(name)
        if not c:
            continue
        t = sum(c.get(k, 0) for k in valu_classes)
        tot[name] = t
        w(f"| `{name[6:]}` | {nodes[name]} | " + " | ".join(str(c.get(k, 0)) for k in valu_classes) + f" | {t} | {c['lds']} | {c['salu']} |")
    # per-iteration budget for h.txt: 10 CN block pairs (640 + 384 = 16 blocks: 8 pairs... the plan pairs equal degrees first)
    calls = {"probe_cn33": 5.0, "probe_cn44": 3.0, "probe_vn1x2": 4.0, "probe_vn2x2": 4.0, "probe_vn15": 2.0}
    w("")
    w("h.txt per iteration and frame: 10 blocks of degree-3 check nodes (5 pair calls), 6 of degree 4 (3 pair calls), 8 blocks of")
    w("leaves and 8 of degree-2 variable nodes (4 pair calls each), 2 blocks of degree-15 variable nodes; 3 456 edges.")
    w("")
    w("| class | lane-instructions per edge-update in the routines |" + (" measured (rocprofv3 mix pass) |" if args.measured_mix else ""))
    w("|---|---|" + ("---|" if args.measured_mix else ""))
    meas = json.loads(args.measured_mix) if args.measured_mix else {}
    per_class = {}
    for k in valu_classes:
        per_class[k] = sum(calls[n] * counts[n].get(k, 0) for n in calls if n in counts) * 64 / NNZ
    merged = {"fp64_fma": per_class["fp64_fma"], "fp64_mul": per_class["fp64_mul"], "fp64_add": per_class["fp64_add"] + per_class["fp64_other"],
              "fp64_reciprocal": per_class["fp64_reciprocal"],
              "int32": per_class["int_bit"] + per_class["int_arith"] + per_class["compare_select"] + per_class["move"] + per_class["valu_other"]}
    for k, v in merged.items():
        w(f"| {k} | {v:.2f} |" + (f" {meas.get(k, float('nan')):.2f} |" if meas else ""))
    routines = sum(merged.values())
    w(f"| **all VALU, routines** | **{routines:.2f}** |" + (f" {meas.get('all_valu', args.measured_all_valu or float('nan')):.2f} (whole kernel) |" if meas else ""))
    if args.measured_all_valu:
        rest = args.measured_all_valu - routines
        w("")
        w(f"Measured whole-kernel figure: {args.measured_all_valu:.2f} VALU lane-instructions per edge-update.  The routines above account for")
        w(f"{routines:.2f}; the remaining {rest:.2f} ({100 * rest / args.measured_all_valu:.0f} %) are loop control and block descriptors, the syndrome /")
        w(f"escape vote, and the per-frame prologue (channel + LLR initialisation, lambda(L_ch), slot indices, v2c initialisation) and")
        w(f"epilogue spread over the frame's {args.avg_iterations:.1f} executed iterations.")
    if "probe_cn33_separate" in tot:
        w("")
        w(f"Shared reciprocal vs separate quotients (round 2): degree-3 pair {tot['probe_cn33']} vs {tot['probe_cn33_separate']} instructions, degree-4 pair "
          f"{tot['probe_cn44']} vs {tot['probe_cn44_separate']}; reciprocals per pair {counts['probe_cn33'].get('fp64_reciprocal', 0)} vs "
          f"{counts['probe_cn33_separate'].get('fp64_reciprocal', 0)} and {counts['probe_cn44'].get('fp64_reciprocal', 0)} vs {counts['probe_cn44_separate'].get('fp64_reciprocal', 0)}.")
    text = "\n".join(lines) + "\n"
    if args.out:
        open(args.out, "w").write(text)
    print(text)


if __name__ == "__main__":
    main()
