"""ctypes binding of the batch interface (Part 2 of include/ldpc_amd.h)."""
import ctypes as ct
import os

import numpy as np

PKG = os.path.dirname(os.path.abspath(__file__))
# LDPC_AMD_LIB selects another build of the library (same-box A/B runs, tools/ab.sh); the default is the in-tree product
LIB_PATH = os.environ.get("LDPC_AMD_LIB") or os.path.join(PKG, "libldpc.so")

AWGN, BSC, BEC = 1, 2, 3
CHANNELS = {"AWGN": AWGN, "BSC": BSC, "BEC": BEC}


class decoder_param(ct.Structure):
    _fields_ = [("earlyTerm", ct.c_bool), ("iterations", ct.c_uint32), ("type", ct.c_char_p)]


class channel_param(ct.Structure):
    _fields_ = [("seed", ct.c_uint64), ("xRange", ct.c_double * 3), ("type", ct.c_char_p)]


class simulation_param(ct.Structure):
    _fields_ = [("threads", ct.c_uint32), ("maxFrames", ct.c_uint64), ("fec", ct.c_uint64),
                ("resultFile", ct.c_char_p)]


class sim_results_t(ct.Structure):
    _fields_ = [("fer", ct.POINTER(ct.c_double)), ("ber", ct.POINTER(ct.c_double)),
                ("avg_iter", ct.POINTER(ct.c_double)), ("time", ct.POINTER(ct.c_double)),
                ("fec", ct.POINTER(ct.c_uint64)), ("frames", ct.POINTER(ct.c_uint64))]


class ldpc_hip_out(ct.Structure):
    _fields_ = [("iters", ct.c_void_p), ("bit_errors", ct.c_void_p), ("hard", ct.c_void_p),
                ("llr_out", ct.c_void_p), ("llr_in", ct.c_void_p), ("codeword", ct.c_void_p)]


_lib = None


def load_library(path=LIB_PATH):
    """Load libldpc.so; raises if it has not been built (python -m libldpc_amd.build)."""
    global _lib
    if _lib is not None and path == LIB_PATH:
        return _lib
    if not os.path.exists(path):
        raise OSError(f"{path} is missing: build the HIP extension first (python -m libldpc_amd.build); "
                      "there is no CPU fallback")
    L = ct.CDLL(path)
    vp, u64, i32 = ct.c_void_p, ct.c_uint64, ct.c_int
    L.ldpc_hip_device_count.restype = i32
    L.ldpc_hip_last_error.restype = ct.c_char_p
    L.ldpc_hip_create.restype = vp
    L.ldpc_hip_create.argtypes = [ct.c_char_p, ct.c_char_p, i32]
    L.ldpc_hip_destroy.argtypes = [vp]
    L.ldpc_hip_code_info.argtypes = [vp, ct.POINTER(ct.c_int64)]
    L.ldpc_hip_set_bec_compat.argtypes = [vp, i32]
    L.ldpc_hip_set_fast_mode.argtypes = [vp, i32]
    L.ldpc_hip_decode_batch.restype = i32
    L.ldpc_hip_decode_batch.argtypes = [vp, decoder_param, u64, vp, ct.POINTER(ldpc_hip_out), vp]
    L.ldpc_hip_stream_begin.restype = i32
    L.ldpc_hip_stream_begin.argtypes = [vp, i32, u64, ct.c_double]
    L.ldpc_hip_stream_skip.restype = i32
    L.ldpc_hip_stream_skip.argtypes = [vp, u64, vp]
    L.ldpc_hip_stream_decode.restype = i32
    L.ldpc_hip_stream_decode.argtypes = [vp, decoder_param, u64, ct.POINTER(ldpc_hip_out), vp]
    L.ldpc_hip_stream_frame.restype = u64
    L.ldpc_hip_stream_frame.argtypes = [vp]
    L.ldpc_hip_stream_raw_draws.restype = u64
    L.ldpc_hip_stream_raw_draws.argtypes = [vp]
    L.ldpc_hip_synchronize.restype = i32
    L.ldpc_hip_synchronize.argtypes = [vp, vp]
    L.ldpc_hip_batch_counters.restype = i32
    L.ldpc_hip_batch_counters.argtypes = [vp, vp, vp, u64, ct.c_uint32, i32, vp, vp]
    L.ldpc_hip_selftest_division.restype = i32
    L.ldpc_hip_selftest_division.argtypes = [vp, u64, u64, vp]
    L.ldpc_hip_mt64.restype = i32
    L.ldpc_hip_mt64.argtypes = [vp, u64, u64, u64, vp, vp]
    L.ldpc_hip_set_profiling.argtypes = [vp, i32]
    L.ldpc_hip_last_ms.restype = ct.c_float
    L.ldpc_hip_last_ms.argtypes = [vp, i32]
    L.ldpc_hip_comm_unique_id.restype = i32
    L.ldpc_hip_comm_unique_id.argtypes = [vp]
    L.ldpc_hip_comm_create.restype = vp
    L.ldpc_hip_comm_create.argtypes = [i32, i32, i32, vp]
    L.ldpc_hip_comm_create_shm.restype = vp
    L.ldpc_hip_comm_create_shm.argtypes = [i32, i32, ct.c_char_p]
    L.ldpc_hip_comm_create_echo.restype = vp
    L.ldpc_hip_comm_create_echo.argtypes = [i32, i32]
    L.ldpc_hip_comm_destroy.argtypes = [vp]
    L.ldpc_hip_comm_allgather.restype = i32
    L.ldpc_hip_comm_allgather.argtypes = [vp, vp, vp, u64]
    L.ldpc_hip_fused_plan_info.restype = None
    L.ldpc_hip_fused_plan_info.argtypes = [vp, vp]
    L.ldpc_hip_selftest_layer_plan.restype = i32
    L.ldpc_hip_selftest_layer_plan.argtypes = [vp, vp]
    L.ldpc_hip_selftest_place.restype = i32
    L.ldpc_hip_selftest_place.argtypes = [vp, u64, u64, u64, u64, u64, u64, u64, vp]
    L.ldpc_hip_comm_stats.restype = None
    L.ldpc_hip_comm_stats.argtypes = [vp, vp, i32]
    L.ldpc_hip_comm_describe.restype = ct.c_char_p
    L.ldpc_hip_comm_describe.argtypes = [vp]
    L.ldpc_hip_shard_capacity.restype = u64
    L.ldpc_hip_shard_capacity.argtypes = [vp, u64, i32]
    L.ldpc_hip_stream_decode_sharded.restype = i32
    L.ldpc_hip_stream_decode_sharded.argtypes = [vp, vp, decoder_param, u64, ct.POINTER(ldpc_hip_out), vp, vp]
    L.ldpc_hip_simulate_sharded.restype = i32
    L.ldpc_hip_simulate_sharded.argtypes = [vp, vp, decoder_param, channel_param, simulation_param, ct.POINTER(sim_results_t),
                                            vp, ct.POINTER(ct.c_bool), i32]
    L.ldpc_hip_selftest_math.restype = i32
    L.ldpc_hip_selftest_math.argtypes = [vp, i32, u64, vp, vp, vp]
    L.ldpc_hip_selftest_chunk_table.restype = u64
    L.ldpc_hip_selftest_chunk_table.argtypes = [u64, u64, u64, u64]
    L.ldpc_hip_selftest_shard_table.restype = u64
    L.ldpc_hip_selftest_shard_table.argtypes = [i32, i32, ct.c_uint32, u64, vp]
    L.ldpc_hip_jump_tasks.restype = u64
    L.ldpc_hip_jump_tasks.argtypes = [vp]
    L.ldpc_hip_simulate.restype = i32
    L.ldpc_hip_simulate.argtypes = [vp, decoder_param, channel_param, simulation_param, ct.POINTER(sim_results_t),
                                    vp, ct.POINTER(ct.c_bool), i32]
    if path == LIB_PATH:
        _lib = L
    return L


def _ptr(buf):
    """Address of a numpy array or of a torch tensor (host or device)."""
    if buf is None:
        return None
    if isinstance(buf, np.ndarray):
        assert buf.flags["C_CONTIGUOUS"]
        return buf.ctypes.data
    if hasattr(buf, "data_ptr"):
        assert buf.is_contiguous()
        return buf.data_ptr()
    raise TypeError(f"unsupported buffer type {type(buf)}")


def _dec(early_term, iterations, decoding):
    return decoder_param(bool(early_term), int(iterations), decoding.encode())


class Comm:
    """The exchange between the ranks of a sharded simulation (include/ldpc_amd.h part 3): RCCL when `unique_id` (128
    bytes from Comm.unique_id() on rank 0) is given, host shared memory when `shm_name` is."""

    def __init__(self, rank, world, device=0, unique_id=None, shm_name=None, echo=False, lib=LIB_PATH):
        self.lib = load_library(lib)
        self.rank, self.world = int(rank), int(world)
        if echo:  # one process standing in for one rank of `world` (cost probes): no transport
            self.handle = self.lib.ldpc_hip_comm_create_echo(self.rank, self.world)
        elif shm_name is not None:
            self.handle = self.lib.ldpc_hip_comm_create_shm(self.rank, self.world, shm_name.encode())
        else:
            buf = (ct.c_uint8 * 128).from_buffer_copy(bytes(unique_id))
            self.handle = self.lib.ldpc_hip_comm_create(self.rank, self.world, int(device), buf)
        if not self.handle:
            raise RuntimeError("communicator: " + self.lib.ldpc_hip_last_error().decode())

    @staticmethod
    def unique_id(lib=LIB_PATH):
        L = load_library(lib)
        buf = (ct.c_uint8 * 128)()
        if L.ldpc_hip_comm_unique_id(buf) != 0:
            raise RuntimeError("ldpc_hip_comm_unique_id: " + L.ldpc_hip_last_error().decode())
        return bytes(buf)

    def all_gather(self, values):
        """values: 1-D uint64 array (at most 32 words) -> [world][len] array"""
        v = np.ascontiguousarray(values, np.uint64)
        out = np.zeros((self.world, v.size), np.uint64)
        if self.lib.ldpc_hip_comm_allgather(self.handle, v.ctypes.data, out.ctypes.data, v.nbytes) != 0:
            raise RuntimeError("ldpc_hip_comm_allgather: " + self.lib.ldpc_hip_last_error().decode())
        return out

    def place(self, nct, pairs_before, frame_pos, cap, piece_pairs, pairs_with_margin, status=0):
        """The placement step of a sharded AWGN step over this communicator (no GPU): (first frame, frames of this rank,
        frames of the step, first pair of the piece, first pair after the step)."""
        out = (ct.c_uint64 * 5)()
        if self.lib.ldpc_hip_selftest_place(self.handle, nct, pairs_before, frame_pos, cap, piece_pairs, pairs_with_margin, status, out) != 0:
            raise RuntimeError("ldpc_hip_selftest_place: " + self.lib.ldpc_hip_last_error().decode())
        return tuple(int(v) for v in out)

    def exchange_stats(self, reset=False):
        """host microseconds inside the all-gathers since the last reset: {"calls", "min", "median", "max"}"""
        out = (ct.c_double * 4)()
        self.lib.ldpc_hip_comm_stats(self.handle, out, int(reset))
        return {"calls": int(out[0]), "min": out[1], "median": out[2], "max": out[3]}

    def describe(self):
        return self.lib.ldpc_hip_comm_describe(self.handle).decode()

    def close(self):
        if getattr(self, "handle", None):
            self.lib.ldpc_hip_comm_destroy(self.handle)
            self.handle = None

    __del__ = close


class HipDecoder:
    """One code on one GPU.  All batch calls take numpy arrays or torch tensors as buffers."""

    OUT_SPEC = {"iters": (np.uint32, False), "bit_errors": (np.uint32, False), "hard": (np.uint8, True),
                "llr_out": (np.float64, True), "llr_in": (np.float64, True), "codeword": (np.uint8, True)}

    def __init__(self, pc_file, gen_file="", device=0, lib=LIB_PATH):
        self.lib = load_library(lib)
        self.ctx = self.lib.ldpc_hip_create(pc_file.encode(), gen_file.encode(), device)
        if not self.ctx:
            raise RuntimeError("ldpc_hip_create: " + self.lib.ldpc_hip_last_error().decode())
        info = (ct.c_int64 * 10)()
        self.lib.ldpc_hip_code_info(self.ctx, info)
        (self.nc, self.mc, self.nnz, self.nct, self.mct, self.kct, self.kc, self.max_degree, lds,
         self.lds_bytes) = list(info)
        self.lds_resident = lds == 1
        self.residency = {0: "memory", 1: "lds", 2: "registers", 3: "registers"}[int(lds)]
        self.register_form = {2: "messages", 3: "totals"}.get(int(lds))  # which register-resident kernel

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.ldpc_hip_destroy(self.ctx)
            self.ctx = None

    __del__ = close

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what}: " + self.lib.ldpc_hip_last_error().decode())

    def _outs(self, n, want, out):
        """Build the output struct; `want` names host (numpy) outputs to allocate, `out` passes buffers."""
        bufs = dict(out or {})
        for name in want:
            if name not in bufs:
                dt, per_bit = self.OUT_SPEC[name]
                bufs[name] = np.zeros((n, self.nc) if per_bit else (n,), dt)
        s = ldpc_hip_out(*[_ptr(bufs.get(k)) for k in ("iters", "bit_errors", "hard", "llr_out", "llr_in", "codeword")])
        return s, bufs

    def set_bec_compat(self, on):
        self.lib.ldpc_hip_set_bec_compat(self.ctx, int(on))

    def fused_plan(self):
        """Summary of the plan of the fused form (include/ldpc_amd.h, ldpc_hip_fused_plan_info)."""
        info = (ct.c_int64 * 8)()
        self.lib.ldpc_hip_fused_plan_info(self.ctx, info)
        return dict(zip(("ok", "n_slots", "vnb", "cnl", "small", "calls_stride", "has_shortened", "table_entries"), [int(v) for v in info]))

    def set_fast_mode(self, on):
        """Opt-in NON-PARITY mode: sum-product with binary32 messages (include/ldpc_amd.h)."""
        self.lib.ldpc_hip_set_fast_mode(self.ctx, int(on))

    def decode_batch(self, llr_in, early_term=True, iterations=50, decoding="BP",
                     want=("iters", "hard", "llr_out"), out=None, stream=None):
        """Decode llr_in[n][nc] (column order)."""
        n = int(llr_in.shape[0])
        s, bufs = self._outs(n, want, out)
        self._check(self.lib.ldpc_hip_decode_batch(self.ctx, _dec(early_term, iterations, decoding), n, _ptr(llr_in),
                                                   ct.byref(s), stream), "ldpc_hip_decode_batch")
        return bufs

    def stream_begin(self, channel, seed, x):
        """Channel point x of mt19937_64(seed); also resets the encoder (a fresh channel object)."""
        ch = CHANNELS[channel] if isinstance(channel, str) else int(channel)
        self._check(self.lib.ldpc_hip_stream_begin(self.ctx, ch, int(seed), float(x)), "ldpc_hip_stream_begin")

    def stream_skip(self, n, stream=None):
        self._check(self.lib.ldpc_hip_stream_skip(self.ctx, int(n), stream), "ldpc_hip_stream_skip")

    def stream_decode(self, n, early_term=True, iterations=50, decoding="BP", want=("iters", "bit_errors"),
                      out=None, stream=None):
        s, bufs = self._outs(int(n), want, out)
        self._check(self.lib.ldpc_hip_stream_decode(self.ctx, _dec(early_term, iterations, decoding), int(n),
                                                    ct.byref(s), stream), "ldpc_hip_stream_decode")
        return bufs

    def shard_capacity(self, target_frames, world):
        return int(self.lib.ldpc_hip_shard_capacity(self.ctx, int(target_frames), int(world)))

    def stream_decode_sharded(self, comm, target_frames, early_term=True, iterations=50, decoding="BP",
                              want=("iters", "bit_errors"), out=None, stream=None):
        """This rank's share of the next global step of about target_frames frames (every rank calls it).  Returns
        (buffers, step) with step = (step_first, step_frames, first, n); only the first n entries of the buffers are
        this rank's frames."""
        cap = self.shard_capacity(target_frames, comm.world)
        s, bufs = self._outs(cap, want, out)
        step = (ct.c_uint64 * 4)()
        self._check(self.lib.ldpc_hip_stream_decode_sharded(self.ctx, comm.handle, _dec(early_term, iterations, decoding),
                                                            int(target_frames), ct.byref(s), step, stream),
                    "ldpc_hip_stream_decode_sharded")
        return bufs, tuple(int(v) for v in step)

    @property
    def stream_frame(self):
        return self.lib.ldpc_hip_stream_frame(self.ctx)

    @property
    def stream_raw_draws(self):
        n = self.lib.ldpc_hip_stream_raw_draws(self.ctx)
        if n == 2**64 - 1:
            raise RuntimeError("ldpc_hip_stream_raw_draws: " + self.lib.ldpc_hip_last_error().decode())
        return n

    @property
    def jump_tasks(self):
        """jump-ahead tasks the context's noise stream has launched so far"""
        return self.lib.ldpc_hip_jump_tasks(self.ctx)

    def synchronize(self, stream=None):
        self._check(self.lib.ldpc_hip_synchronize(self.ctx, stream), "ldpc_hip_synchronize")

    def set_profiling(self, on=True):
        self.lib.ldpc_hip_set_profiling(self.ctx, int(on))

    def last_ms(self, which=0):
        return float(self.lib.ldpc_hip_last_ms(self.ctx, which))

    def batch_counters(self, iters_ptr, bit_errors_ptr, n, max_iters, early_term, counters_ptr, stream=None):
        """{frames, frame errors, bit errors, iterations, early stops} of a batch, summed on the device (all three
        arguments are device pointers, e.g. tensor.data_ptr()); one launch on `stream`."""
        self._check(self.lib.ldpc_hip_batch_counters(self.ctx, iters_ptr, bit_errors_ptr, int(n), int(max_iters),
                                                     int(bool(early_term)), counters_ptr, stream), "ldpc_hip_batch_counters")

    def selftest_division(self, n, seed=1):
        """Pairs (of n) on which the kernels' division sequence and the IEEE division disagree (expected 0)."""
        bad = ct.c_uint64(0)
        self._check(self.lib.ldpc_hip_selftest_division(self.ctx, int(n), int(seed), ct.byref(bad)),
                    "ldpc_hip_selftest_division")
        return bad.value

    MATH_FNS = ("exp", "log", "boxplus", "ratio_div", "ratio_rho", "ratio_lambda", "e_combine", "exp_clamped", "boxplus_exp",
                "boxplus_log", "cn_ratio3", "cn_ratio4", "cn_ratio5", "cn_ratio6", "cn_ratio8", "cn_llr4", "cn_llr6", "cn_ratio3s", "cn_ratio4s", "cn_ratio6s")

    def selftest_math(self, fn, a, b=None):
        """The device arithmetic of detmath.h / device_cn.hpp on host arrays: fn is a name of MATH_FNS."""
        a = np.ascontiguousarray(a, np.float64)
        out = np.empty_like(a)
        n = a.shape[0]
        bb = None if b is None else np.ascontiguousarray(b, np.float64)
        self._check(self.lib.ldpc_hip_selftest_math(self.ctx, self.MATH_FNS.index(fn), n, _ptr(a), _ptr(bb), _ptr(out)),
                    "ldpc_hip_selftest_math")
        return out

    def mt64(self, seed, first, n):
        out = np.zeros(int(n), np.uint64)
        self._check(self.lib.ldpc_hip_mt64(self.ctx, int(seed), int(first), int(n), _ptr(out), None), "ldpc_hip_mt64")
        return out

    def simulate(self, channel, x_range, seed=0, early_term=True, iterations=50, decoding="BP",
                 max_frames=10**10, fec=50, result_file="", cli_output=False, comm=None):
        n_max = max(1, int(np.ceil((x_range[1] - x_range[0]) / x_range[2])) + 2)
        arrs = {k: np.zeros(n_max, np.float64) for k in ("fer", "ber", "avg_iter", "time")}
        arrs["fec"] = np.zeros(n_max, np.uint64)
        arrs["frames"] = np.zeros(n_max, np.uint64)
        res = sim_results_t(*[arrs[k].ctypes.data_as(ct.POINTER(ct.c_double if arrs[k].dtype == np.float64
                                                                 else ct.c_uint64))
                              for k in ("fer", "ber", "avg_iter", "time", "fec", "frames")])
        totals = np.zeros(4 * n_max, np.uint64)
        stop = ct.c_bool(False)
        ch = channel_param(int(seed), (ct.c_double * 3)(*x_range), channel.encode())
        sp = simulation_param(1, int(max_frames), int(fec), result_file.encode())
        if comm is not None:
            nx = self.lib.ldpc_hip_simulate_sharded(self.ctx, comm.handle, _dec(early_term, iterations, decoding), ch, sp,
                                                    ct.byref(res), totals.ctypes.data, ct.byref(stop), int(cli_output))
        else:
            nx = self.lib.ldpc_hip_simulate(self.ctx, _dec(early_term, iterations, decoding), ch, sp, ct.byref(res),
                                            totals.ctypes.data, ct.byref(stop), int(cli_output))
        if nx < 0:
            raise RuntimeError("ldpc_hip_simulate: " + self.lib.ldpc_hip_last_error().decode())
        out = {k: v[:nx] for k, v in arrs.items()}
        out["totals"] = totals[:4 * nx].reshape(nx, 4)
        return out
