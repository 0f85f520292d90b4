"""Build the native library and CLI in-tree with hipcc for gfx950.

    python -m libldpc_amd.build            # libldpc_amd/libldpc.so + libldpc_amd/ldpcsim

Objects are cached under libldpc_amd/csrc/build/ and rebuilt when a source or header changes.
The shared object keeps the reference's name (libldpc.so, CMakeLists.txt:21) so that
pyLDPC's `LDPC(..., lib=<path>)` can be pointed at it unchanged.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(PKG, "libldpc.so")
CLI = os.path.join(PKG, "ldpcsim")

LIB_SOURCES = ["kernels.hip", "kernels_reg.hip", "kernels_reg2.hip", "kernels_reg2u.hip", "kernels_fast.hip", "kernels_layered.hip", "kernels_fused.hip", "kernels_bec.hip", "rng_kernels.hip", "selftest.hip", "engine.cpp", "comm.cpp", "api.cpp", "sim.cpp", "code.cpp", "plan.cpp", "mt64.cpp", "mtstates.cpp"]
CLI_SOURCES = ["ldpcsim_main.cpp"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -ffp-contract=off: every fused multiply-add is written explicitly (detmath.h); results must not
# depend on the compiler's contraction choices.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++20", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fvisibility=hidden",
         "-Wall", "-Wno-unused-function", "-Wno-unused-result"]
if os.environ.get("LDPC_AMD_PHASE_TRACE_BUILD"):  # debug build with per-wave phase timers (tools/phase_probe.py)
    FLAGS.append("-DLDPC_AMD_PHASE_TRACE")
# objects are cached per flag set, so a debug build never mixes with normal objects
import hashlib
OBJ = os.path.join(OBJ, hashlib.sha256(" ".join(FLAGS).encode()).hexdigest()[:10])
if os.environ.get("LDPC_AMD_PHASE_TRACE_BUILD"):
    LIB = os.path.join(PKG, "libldpc_trace.so")  # load it with LDPC_AMD_LIB; the product library stays as built
    CLI = os.path.join(OBJ, "ldpcsim_trace")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _headers():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hpp", ".h"))]
    hs.append(os.path.join(os.path.dirname(PKG), "include", "ldpc_amd.h"))
    return hs


def _compile(src):
    obj = os.path.join(OBJ, src + ".o")
    path = os.path.join(CSRC, src)
    if _stale(obj, [path] + _headers()):
        cmd = [HIPCC] + FLAGS + ["-c", path, "-o", obj]
        if src.endswith(".hip"):
            # keep the compiler's per-kernel resource report next to the object: the headline kernel lives exactly at
            # the 96-VGPR boundary of five frames per CU, and tests/test_host.py checks that it still does
            cmd.append("-Rpass-analysis=kernel-resource-usage")
            p = subprocess.run(cmd, stderr=subprocess.PIPE, text=True)
            remarks = [l for l in p.stderr.splitlines() if "-Rpass-analysis=kernel-resource-usage" in l]
            lines = p.stderr.splitlines()
            if p.returncode:
                sys.stderr.write(p.stderr)
            else:  # diagnostics other than the remarks (each remark is followed by a source line and a caret line)
                skip, held = 0, []
                for l in lines:
                    if l.startswith("In file included from "):
                        held.append(l)  # the include trail of the next diagnostic: shown unless that is a remark
                        continue
                    if l in remarks:
                        skip, held = 2, []
                    elif skip and ("|" in l[:8] or not l.strip()):
                        skip -= 1
                    elif "remark generated" not in l and "remarks generated" not in l:
                        for h in held:
                            sys.stderr.write(h + "\n")
                        held = []
                        sys.stderr.write(l + "\n")
            if p.returncode:
                raise subprocess.CalledProcessError(p.returncode, cmd)
            with open(obj + ".resources.txt", "w") as f:
                f.write("\n".join(l.split("remark: ", 1)[1].replace(" [-Rpass-analysis=kernel-resource-usage]", "")
                                  for l in remarks if "remark: " in l) + "\n")
        else:
            subprocess.check_call(cmd)
    return obj


def kernel_resources(src="kernels.hip"):
    """{mangled kernel name: {"VGPRs": n, "ScratchSize [bytes/lane]": n, "Occupancy [waves/SIMD]": n, ...}} from the last
    compile of `src` (None if the report is missing, e.g. objects built by an older build.py)."""
    path = os.path.join(OBJ, src + ".o.resources.txt")
    if not os.path.exists(path):
        return None
    out, cur = {}, None
    for line in open(path):
        line = line.strip()
        if line.startswith("Function Name: "):
            cur = out.setdefault(line[len("Function Name: "):], {})
        elif cur is not None and ": " in line:
            k, v = line.rsplit(": ", 1)
            try:
                cur[k.strip()] = int(v)
            except ValueError:
                pass
    return out


def build(verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    srcs = LIB_SOURCES + [s for s in CLI_SOURCES if os.path.exists(os.path.join(CSRC, s))]
    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = dict(zip(srcs, ex.map(_compile, srcs)))
    lib_objs = [objs[s] for s in LIB_SOURCES]
    if _stale(LIB, lib_objs):
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-o", LIB] + lib_objs)
    cli_objs = [objs[s] for s in CLI_SOURCES if s in objs]
    if cli_objs and _stale(CLI, cli_objs + [LIB]):
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-o", CLI] + cli_objs +
                              ["-L" + PKG, "-l:" + os.path.basename(LIB), "-Wl,-rpath,$ORIGIN"])
    if verbose:
        print("built", LIB, "and", CLI if cli_objs else "(no CLI)")
    return LIB


if __name__ == "__main__":
    build(verbose=True)
    sys.exit(0)
