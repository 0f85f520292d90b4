"""ctypes binding of the CPU oracle (oracle/liboracle.so) — test infrastructure only.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
"""
import ctypes as ct
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "liboracle.so")
GOLDEN = os.path.join(ROOT, "tests", "golden")
H_TXT = os.path.join(GOLDEN, "h.txt")
G_TXT = os.path.join(GOLDEN, "g.txt")

MATH_LIBM, MATH_DET = 0, 1
AWGN, BSC, BEC = 1, 2, 3
CHAN = {"AWGN": AWGN, "BSC": BSC, "BEC": BEC}
ERASURE = ord("E")

_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, os.path.join(ORACLE_DIR, "liboracle.so")])


def lib():
    global _lib
    if _lib is not None:
        return _lib
    src = os.path.join(ORACLE_DIR, "ldpc_oracle.c")
    if not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(src):
        build()
    L = ct.CDLL(LIB_PATH)
    vp, i32, u32, u64, dbl = ct.c_void_p, ct.c_int, ct.c_uint, ct.c_uint64, ct.c_double
    L.orc_code_load.restype = vp
    L.orc_code_load.argtypes = [ct.c_char_p, ct.c_char_p]
    L.orc_code_free.argtypes = [vp]
    for name in ("nc", "mc", "kc", "nnz", "nct", "mct", "kct", "max_degree", "num_puncture",
                 "num_shorten", "has_G", "g_rows", "g_cols", "g_nnz"):
        f = getattr(L, "orc_code_" + name)
        f.restype, f.argtypes = i32, [vp]
    L.orc_code_edges.argtypes = [vp, vp, vp]
    for name in ("bit_pos", "puncture", "shorten"):
        getattr(L, "orc_code_" + name).argtypes = [vp, vp]
    L.orc_syndrome.argtypes = [vp, vp, vp]
    L.orc_encode_accumulate.argtypes = [vp, vp, vp]
    L.orc_rank.restype, L.orc_rank.argtypes = i32, [vp]
    L.orc_decode.restype = i32
    L.orc_decode.argtypes = [vp, i32, i32, u32, i32, vp, vp, vp]
    L.orc_decode_fast_batch.restype = None
    L.orc_decode_fast_batch.argtypes = [vp, i32, i32, u32, u64, vp, vp, vp, vp]
    L.orc_layer_steps.restype = i32
    L.orc_layer_steps.argtypes = [vp, vp]
    L.orc_decode_bec.restype = i32
    L.orc_decode_bec.argtypes = [vp, i32, u32, i32, vp, vp, vp, vp]
    L.orc_chan_new.restype = vp
    L.orc_chan_new.argtypes = [vp, i32, u64, i32, i32, i32, u32, i32]
    L.orc_chan_free.argtypes = [vp]
    L.orc_chan_set_param.argtypes = [vp, dbl]
    L.orc_chan_raw_draws.restype, L.orc_chan_raw_draws.argtypes = u64, [vp]
    L.orc_chan_run_frames.argtypes = [vp, u64, u64, vp, vp, vp, vp, vp, vp]
    L.orc_simulate.restype = i32
    L.orc_simulate.argtypes = [vp, i32, u64, vp, i32, i32, u32, i32, i32, u32, u64, u64, vp, vp, vp]
    L.orc_mt64_stream.argtypes = [u64, u64, vp]
    L.orc_ratio_stats.argtypes = [vp, vp, i32]
    L.orc_exp.restype, L.orc_exp.argtypes = dbl, [i32, dbl]
    L.orc_log.restype, L.orc_log.argtypes = dbl, [i32, dbl]
    for f in (L.orc_math_ref, L.orc_math_det):
        f.restype, f.argtypes = i32, [i32, u64, vp, vp, vp]
    _lib = L
    return L


def _p(a):
    return None if a is None else a.ctypes.data_as(ct.c_void_p)


class Results(ct.Structure):
    _fields_ = [("fer", ct.c_void_p), ("ber", ct.c_void_p), ("avg_iter", ct.c_void_p),
                ("time", ct.c_void_p), ("fec", ct.c_void_p), ("frames", ct.c_void_p)]


class Code:
    def __init__(self, pc_file, gen_file=""):
        L = lib()
        self.h = L.orc_code_load(pc_file.encode(), gen_file.encode())
        if not self.h:
            raise OSError("oracle: cannot open " + pc_file)
        for name in ("nc", "mc", "kc", "nnz", "nct", "mct", "kct", "max_degree", "num_puncture",
                     "num_shorten", "has_G", "g_rows", "g_cols", "g_nnz"):
            setattr(self, name, getattr(L, "orc_code_" + name)(self.h))
        self.edge_row = np.zeros(self.nnz, np.int32)
        self.edge_col = np.zeros(self.nnz, np.int32)
        L.orc_code_edges(self.h, _p(self.edge_row), _p(self.edge_col))
        self.bit_pos = np.zeros(self.nct, np.int32)
        L.orc_code_bit_pos(self.h, _p(self.bit_pos))
        self.puncture = np.zeros(self.num_puncture, np.int32)
        L.orc_code_puncture(self.h, _p(self.puncture))
        self.shorten = np.zeros(self.num_shorten, np.int32)
        L.orc_code_shorten(self.h, _p(self.shorten))

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_code_free(self.h)
            self.h = None

    def rank(self):
        return lib().orc_rank(self.h)

    def syndrome(self, word):
        word = np.ascontiguousarray(word, np.uint8)
        s = np.zeros(self.mc, np.uint8)
        lib().orc_syndrome(self.h, _p(word), _p(s))
        return s

    def encode(self, info):
        """u*G on all nc columns (fresh codeword)."""
        info = np.ascontiguousarray(info, np.uint8)
        cw = np.zeros(self.g_cols, np.uint8)
        lib().orc_encode_accumulate(self.h, _p(info), _p(cw))
        return cw

    def decode(self, llr_in, min_sum=False, early_term=True, iters=50, math=MATH_LIBM):
        llr_in = np.ascontiguousarray(llr_in, np.float64)
        assert llr_in.shape == (self.nc,)
        out = np.zeros(self.nc, np.float64)
        hard = np.zeros(self.nc, np.uint8)
        it = lib().orc_decode(self.h, int(min_sum), int(early_term), iters, math, _p(llr_in), _p(out), _p(hard))
        return it, out, hard

    def decode_fast(self, mode, llr_in, early_term=True, iters=50):
        """Mirror of the HIP library's NON-PARITY mode `mode` (1, 2, 3) on frames llr_in[n][nc]: (iters, llr_out, hard)."""
        x = np.ascontiguousarray(llr_in, np.float64).reshape(-1, self.nc)
        n = x.shape[0]
        it, out, hard = np.zeros(n, np.uint32), np.zeros((n, self.nc), np.float64), np.zeros((n, self.nc), np.uint8)
        lib().orc_decode_fast_batch(self.h, int(mode), int(early_term), iters, n, _p(x), _p(it), _p(out), _p(hard))
        return it, out, hard

    def layer_steps(self):
        step_of = np.zeros(self.mc, np.int32)
        n = lib().orc_layer_steps(self.h, _p(step_of))
        return n, step_of

    def decode_bec(self, llr_in, codeword, early_term=True, iters=50, compat=False):
        llr_in = np.ascontiguousarray(llr_in, np.uint8)
        codeword = np.ascontiguousarray(codeword, np.uint8)
        out = np.zeros(self.nc, np.uint8)
        hard = np.zeros(self.nc, np.uint8)
        it = lib().orc_decode_bec(self.h, int(early_term), iters, int(compat), _p(llr_in), _p(codeword), _p(out), _p(hard))
        return it, out, hard

    def run_frames(self, chan, x, seed=0, skip=0, count=1, min_sum=False, early_term=True, iters=50,
                   math=MATH_LIBM, bec_compat=False, want_vectors=True):
        """Frames [skip, skip+count) of stream mt19937_64(seed) at channel point x."""
        L = lib()
        ch = L.orc_chan_new(self.h, CHAN[chan] if isinstance(chan, str) else chan, seed, math,
                            int(min_sum), int(early_term), iters, int(bec_compat))
        L.orc_chan_set_param(ch, float(x))
        r = {"iters": np.zeros(count, np.uint32), "bit_errors": np.zeros(count, np.uint32)}
        if want_vectors:
            r["hard"] = np.zeros((count, self.nc), np.uint8)
            r["llr_in"] = np.zeros((count, self.nc), np.float64)
            r["llr_out"] = np.zeros((count, self.nc), np.float64)
            r["codeword"] = np.zeros((count, self.nc), np.uint8)
        L.orc_chan_run_frames(ch, skip, count, _p(r["iters"]), _p(r["bit_errors"]), _p(r.get("hard")),
                              _p(r.get("llr_in")), _p(r.get("llr_out")), _p(r.get("codeword")))
        r["raw_draws"] = L.orc_chan_raw_draws(ch)
        L.orc_chan_free(ch)
        return r

    def simulate(self, chan, x_range, seed=0, min_sum=False, early_term=True, iters=50, math=MATH_LIBM,
                 bec_compat=False, threads=1, max_frames=10**10, min_fec=50):
        L = lib()
        xr = np.asarray(x_range, np.float64)
        n_max = max(1, int(np.ceil((xr[1] - xr[0]) / xr[2])) + 2)
        arrs = {k: np.zeros(n_max, np.float64) for k in ("fer", "ber", "avg_iter", "time")}
        arrs["fec"] = np.zeros(n_max, np.uint64)
        arrs["frames"] = np.zeros(n_max, np.uint64)
        res = Results(*[arrs[k].ctypes.data for k in ("fer", "ber", "avg_iter", "time", "fec", "frames")])
        totals = np.zeros(4 * n_max, np.uint64)
        nx = L.orc_simulate(self.h, CHAN[chan], seed, _p(xr), int(min_sum), int(early_term), iters, math,
                            int(bec_compat), threads, max_frames, min_fec, ct.byref(res), _p(totals), None)
        out = {k: v[:nx] for k, v in arrs.items()}
        out["totals"] = totals[:4 * nx].reshape(nx, 4)  # frames, fec, bec, iters
        return out


def shortened_variant(path):
    """h.txt with `shorten [4]: 130 290 0 1` (130 and 290 are two of the three neighbours of check node 0)."""
    txt = open(H_TXT).read()
    assert "shorten [0]: " in txt
    open(path, "w").write(txt.replace("shorten [0]: ", "shorten [4]: 130 290 0 1"))
    return path


def ratio_stats(reset=False):
    """(frames finished by the likelihood-ratio form, frames handed back to the LLR-domain form) since reset."""
    done, esc = ct.c_uint64(0), ct.c_uint64(0)
    lib().orc_ratio_stats(ct.byref(done), ct.byref(esc), int(reset))
    return done.value, esc.value


def ratio_second():
    """Frames whose first (shared-reciprocal) attempt was given up since the last reset of ratio_stats."""
    f = lib().orc_ratio_second
    f.restype = ct.c_uint64
    return int(f())


def craft_degree6_overflow(code, rng, big=78.0, odd=True, base=2.0, flip=0.01):
    """Input LLRs that make a degree-6 check node's shared reciprocals overflow INSIDE the box of the ratio form (detmath.h,
    dm_cn6_shared: the product of three denominators reaches 2^897 when the node's inputs are near |L| = 156, while every
    message and total stays below 166).  A check node B is chosen; each of its six variable nodes v gets |L_ch| = big and a
    helper check node whose other five neighbours have |L_ch| = big too, so that after one iteration v sends 2 * big to B.
    odd: one of the six is negative — B is unsatisfied, its messages cancel the totals, the frame stays in the box and the
    second launch (outputs divided separately) finishes it; even: the totals double and the frame goes on to the LLR domain."""
    rows = [code.edge_col[code.edge_row == r] for r in range(code.mc)] if not hasattr(code, "_rows") else code._rows
    cols = [code.edge_row[code.edge_col == v] for v in range(code.nc)] if not hasattr(code, "_cols") else code._cols
    code._rows, code._cols = rows, cols
    llr = np.full(code.nc, base)
    llr[rng.random(code.nc) < flip] *= -1
    for _ in range(1000):
        b = int(rng.integers(code.mc))
        if len(rows[b]) != 6:
            continue
        vs, helpers, used, ok = list(rows[b]), [], set(rows[b]), True
        for v in vs:
            cand = [a for a in cols[v] if a != b and len(rows[a]) == 6 and not (set(rows[a]) - {v}) & used]
            if not cand:
                ok = False
                break
            helpers.append(cand[0])
            used |= set(rows[cand[0]])
        if ok:
            break
    else:
        raise AssertionError("no check node of degree 6 with six disjoint helpers")
    for i, (v, a) in enumerate(zip(vs, helpers)):
        s = -1.0 if (odd and i == 5) else 1.0
        others = [u for u in rows[a] if u != v]
        llr[v] = s * big
        llr[others] = big
        if s < 0:
            llr[others[0]] = -big
    return llr


def mt64_stream(seed, n):
    out = np.zeros(n, np.uint64)
    lib().orc_mt64_stream(seed, n, _p(out))
    return out


# ---- reference binaries built into oracle/_ref (present only where /root/reference is) ----
REF_DIR = os.path.join(ORACLE_DIR, "_ref")
REF_DUMP = os.path.join(REF_DIR, "ref_dump")
REF_SIM = os.path.join(REF_DIR, "ldpcsim_ref")
REF_LIB = os.path.join(REF_DIR, "libldpc_ref.so")


def have_ref():
    return os.path.exists(REF_DUMP)


def ref_dump(h, g, chan, dec, iters, early, seed, x, skip, count, tmp):
    """Run the reference through oracle/_ref/ref_dump; returns the same dict as Code.run_frames."""
    subprocess.check_call([REF_DUMP, h, g or "-", chan, dec, str(iters), str(int(early)), str(seed),
                           repr(float(x)), str(skip), str(count), tmp], stdout=subprocess.DEVNULL)
    raw = open(tmp, "rb").read()
    nc, cnt = np.frombuffer(raw, np.int32, 2)
    rec = np.dtype([("iters", "<i4"), ("bit_errors", "<i4"), ("hard", "u1", (nc,)), ("llr_in", "<f8", (nc,)),
                    ("llr_out", "<f8", (nc,)), ("codeword", "u1", (nc,))])
    a = np.frombuffer(raw, rec, cnt, 8)
    return {k: np.ascontiguousarray(a[k]) for k in rec.names}


MATH_FNS = ("exp", "log", "boxplus", "ratio_div", "ratio_rho", "ratio_lambda", "e_combine", "exp_clamped", "boxplus_exp",
            "boxplus_log", "cn_ratio3", "cn_ratio4", "cn_ratio5", "cn_ratio6", "cn_ratio8", "cn_llr4", "cn_llr6", "cn_ratio3s", "cn_ratio4s", "cn_ratio6s")


def math_eval(fn, a, b=None, det=False):
    """out = fn(a[, b]) element by element: det=False -> what the function means (glibc libm / long double, no
    detmath.h), det=True -> detmath.h compiled for the host."""
    a = np.ascontiguousarray(a, np.float64)
    bb = None if b is None else np.ascontiguousarray(b, np.float64)
    out = np.empty_like(a)
    f = lib().orc_math_det if det else lib().orc_math_ref
    assert f(MATH_FNS.index(fn), a.shape[0], _p(a), _p(bb), _p(out)) == 0
    return out


def math_points(fn, n, seed):
    """Deterministic operands for `fn`: (a, b or None).  Covers the whole domain the kernels use each function on,
    with the edges of the representable boxes (|L| = 166, 600, 700, 709) and special operands mixed in."""
    rng = np.random.default_rng([seed, MATH_FNS.index(fn)])
    def mix(parts):
        x = np.concatenate(parts)
        rng.shuffle(x)
        return np.resize(x, n)
    edge = np.array([0.0, -0.0, 1.0, -1.0, 166.0, -166.0, 166.3, -166.3, 600.0, -600.0, 700.0, -700.0, 709.0, -709.0, 709.78,
                     -745.13, 1e-300, -1e-300, 5e-324, 0.5, 2.0, np.log(2.0), 1e-17, -1e-17])
    k = max(n // 4, 1)
    if fn == "exp":
        return mix([rng.uniform(-745.2, 709.78, 2 * k), rng.uniform(-1, 1, k), rng.normal(0, 40, k), edge]), None
    if fn == "log":
        return mix([np.ldexp(rng.uniform(0.5, 1, 2 * k), rng.integers(-1073, 1024, 2 * k)), 1 + rng.normal(0, 1e-3, k),
                    np.exp(rng.uniform(-170, 170, k)), np.abs(edge[edge != 0])]), None
    if fn == "boxplus":
        a = mix([rng.normal(0, 10, k), rng.uniform(-50, 50, k), rng.uniform(-700, 700, k), rng.normal(0, 0.01, k), edge])
        b = mix([rng.normal(0, 10, k), rng.uniform(-50, 50, k), rng.uniform(-700, 700, k), a[:k] * rng.choice([-1, 1], k), edge[::-1]])
        # saturated operands (the device's short cut, device_math.hpp box_jacobian): magnitudes up to 1e15, gaps on both
        # sides of 37, signed zeros and infinities against large values
        q = max(n // 8, 8)
        big = np.ldexp(rng.uniform(0.5, 1, q), rng.integers(-30, 50, q)) * rng.choice([-1, 1], q)
        gap = np.concatenate([rng.uniform(36.5, 37.5, q // 2), np.ldexp(rng.uniform(0.5, 1, q - q // 2), rng.integers(-20, 45, q - q // 2))])
        other = (np.abs(big) + gap) * rng.choice([-1, 1], q)
        special_a = np.array([0.0, -0.0, 0.0, -0.0, np.inf, -np.inf, 7.0, 1e300, -1e300, 37.0, -37.0, 1e-310, 50.0, 0.0])
        special_b = np.array([40.0, 40.0, -1e9, np.inf, 5.0, -1e9, np.inf, -1e300, 1.0, 0.0, -0.0, -80.0, 13.0, 37.0])
        a[:q], b[:q] = big, other
        a[q:2 * q], b[q:2 * q] = other, big
        a[2 * q:2 * q + special_a.size], b[2 * q:2 * q + special_a.size] = special_a, special_b
        return a, b
    if fn == "ratio_div":
        a = np.ldexp(rng.uniform(0.5, 1, n), rng.integers(-499, 500, n))
        q = np.ldexp(rng.uniform(0.5, 1, n), rng.integers(-499, 500, n))
        b = a / q
        ok = (b > 2.0**-999) & (b < 2.0**999)
        b[~ok] = 1.0
        return a, b
    if fn in ("ratio_rho", "ratio_lambda"):
        lim = 240 * np.log(2.0) * (1 - 1e-12)
        a = np.exp(mix([rng.uniform(-lim, lim, 2 * k), rng.normal(0, 5, 2 * k), np.array([lim, -lim, 0.0])]))
        b = np.exp(mix([rng.uniform(-lim, lim, 2 * k), rng.normal(0, 5, 2 * k), np.array([-lim, lim, 0.0])]))
        return a, b
    if fn == "e_combine":
        return np.exp(-mix([rng.uniform(0, 600, 3 * k), rng.exponential(3, k), np.array([0.0, 600.0])])), \
               np.exp(-mix([rng.uniform(0, 600, 3 * k), rng.exponential(3, k), np.array([600.0, 0.0])]))
    if fn == "exp_clamped":
        return mix([rng.uniform(-705, 705, 3 * k), rng.normal(0, 20, k), edge[np.abs(edge) <= 709]]), None
    if fn == "boxplus_exp":
        return mix([rng.uniform(0, 710, 2 * k), rng.exponential(5, 2 * k), np.abs(edge)]), None
    if fn == "boxplus_log":
        return mix([rng.uniform(0.5, 2, 3 * k), 1 + rng.normal(0, 1e-6, k), np.array([0.5, 1.0, 2.0])]), None
    if fn in ("cn_ratio3s", "cn_ratio4s", "cn_ratio6s"):
        # the shared-reciprocal forms: inputs kept where the product of the node's denominators stays below 2^897
        # (degree 3: |L| <= 100, degree 4: |L| <= 50, degree 6: |L| <= 30); beyond that the kernels hand the frame back,
        # covered by whole decodes
        d, lim = {"cn_ratio3s": (3, 100.0), "cn_ratio4s": (4, 50.0), "cn_ratio6s": (6, 30.0)}[fn]
        L = np.concatenate([rng.uniform(-lim, lim, (n // 2, d)), rng.normal(0, 6, (n - n // 2 - 2, d)).clip(-lim, lim),
                            np.full((1, d), lim), np.full((1, d), -lim)])
        return np.exp(L), None
    d = int(fn[-1])
    if fn.startswith("cn_ratio"):
        lim = 240 * np.log(2.0) * (1 - 1e-12)
        L = np.concatenate([rng.uniform(-lim, lim, (n // 2, d)), rng.normal(0, 6, (n - n // 2 - 2, d)),
                            np.full((1, d), lim), np.full((1, d), -lim)])
        return np.exp(L), None
    L = np.concatenate([rng.normal(0, 10, (n // 2, d)), rng.uniform(-100, 100, (n - n // 2 - 2, d)),
                        np.full((1, d), 600.0), np.zeros((1, d))])
    # saturated nodes (detmath.h, dm_sat_*): a large common magnitude plus offsets of the order of channel LLRs, the two
    # edges of the rule (smallest input around 40, spread around 600), and nodes that must take the box-plus chain
    q = n // 4
    big = 10.0 ** rng.uniform(1.5, 13, (q, 1)) + rng.normal(0, 20, (q, d))
    L[:q] = np.abs(big) * rng.choice([-1, 1], (q, d))
    e = q // 4
    L[q:q + e] = (rng.uniform(39.5, 40.5, (e, 1)) + rng.exponential(30, (e, d))) * rng.choice([-1, 1], (e, d))
    L[q + e:q + 2 * e] = (rng.uniform(50, 1e4, (e, 1)) + rng.uniform(0, 1, (e, d)) * rng.uniform(590, 610, (e, 1))) * rng.choice([-1, 1], (e, d))
    L[q + 2 * e:q + 3 * e] = rng.uniform(-2000, 2000, (e, d))
    return L, None
