"""Host-side mirror of the reference's Python wrapper (pyLDPC/ldpc.py:32-218) over our libldpc.so.

Same class name, constructor, methods and return conventions, so code written against
``pyLDPC.ldpc.LDPC`` runs against the MI355X library by changing the import (or by passing
``lib=libldpc_amd.LIB_PATH`` to the reference's own wrapper, which works unchanged).
"""
import ctypes as ct
import threading

import numpy as np

from .binding import LIB_PATH, channel_param, decoder_param, sim_results_t, simulation_param

MAX_POINTS = 50  # the reference wrapper allocates 50 result slots (ldpc.py:56-66)


class LDPC:
    def __init__(self, pc_file: str, gen_file="", lib=LIB_PATH):
        self.pc_file, self.gen_file = pc_file, gen_file
        self.lib = ct.cdll.LoadLibrary(lib)
        n, m, nct, mct = ct.c_int(0), ct.c_int(0), ct.c_int(0), ct.c_int(0)
        self.lib.ldpc_setup(pc_file.encode("utf-8"), gen_file.encode("utf-8"), ct.byref(n), ct.byref(m),
                            ct.byref(nct), ct.byref(mct))
        self.n, self.m, self.nct, self.mct = n.value, m.value, nct.value, mct.value
        self.kct, self.k = self.nct - self.mct, self.n - self.m
        self.sim_stop_flag = ct.c_bool(False)
        self._res = {k: (ct.c_double * MAX_POINTS)() for k in ("fer", "ber", "avg_iter", "time")}
        self._res.update({k: (ct.c_uint64 * MAX_POINTS)() for k in ("fec", "frames")})
        self.sim_results_struct = sim_results_t(*[ct.cast(self._res[k], ct.POINTER(t)) for k, t in (
            ("fer", ct.c_double), ("ber", ct.c_double), ("avg_iter", ct.c_double), ("time", ct.c_double),
            ("fec", ct.c_uint64), ("frames", ct.c_uint64))])
        self.results = {}
        self.sim_params = {"earlyTerm": True, "iterations": 50, "decoding": "BP", "seed": 0, "snr": [],
                           "channel": "AWGN", "threads": 1, "maxFrames": int(10e9), "fec": 50}
        self._thread = None

    def encode(self, info_word):
        if not self.gen_file:
            raise RuntimeError("No generator matrix provided for encoding")
        u = (ct.c_uint8 * self.kct)(*[int(b) for b in info_word])
        cw = (ct.c_uint8 * self.nct)()
        self.lib.encode(ct.byref(u), ct.byref(cw))
        return np.array(cw[0:self.nct])

    def decode(self, llr_in, early_term=True, iters=50, dec_type="BP"):
        p = decoder_param(early_term, iters, dec_type.encode("utf-8"))
        vin = (ct.c_double * self.nct)(*[float(v) for v in llr_in])
        vout = (ct.c_double * self.nct)()
        self.lib.decode.restype = ct.c_int
        it = self.lib.decode(p, ct.byref(vin), ct.byref(vout))
        return np.array(vout[0:self.nct]), it

    def simulate(self, **args):
        self.sim_params = {**self.sim_params, **args}
        sp = self.sim_params
        dec = decoder_param(sp["earlyTerm"], sp["iterations"], sp["decoding"].encode("utf-8"))
        ch = channel_param(sp["seed"], (ct.c_double * 3)(*sp["snr"]), sp["channel"].encode("utf-8"))
        sim = simulation_param(sp["threads"], sp["maxFrames"], sp["fec"], b"")

        def run():
            self.sim_stop_flag.value = False
            self.lib.simulate(dec, ch, sim, ct.byref(self.sim_results_struct), ct.byref(self.sim_stop_flag))

        self._thread = threading.Thread(target=run)
        self._thread.start()

    def wait(self):
        """Block until the running simulation has finished (not in the reference wrapper)."""
        if self._thread is not None:
            self._thread.join()

    def stop_simulation(self):
        if not self.sim_stop_flag.value:
            self.results = self.get_results()
            self.sim_stop_flag.value = True

    def get_results(self):
        if not self.sim_stop_flag.value:
            n = int(np.sum(np.array(self._res["frames"][0:MAX_POINTS]) > 0))
            return {k: list(self._res[k][0:n]) for k in ("fer", "ber", "avg_iter", "time", "fec", "frames")}
        return self.results

    def rank(self):
        return self.lib.calculate_rank()

    def syndrome(self, v):
        word = (ct.c_uint8 * self.n)(*[int(b) for b in v])
        synd = (ct.c_uint8 * self.n)()
        self.lib.syndrome(ct.byref(word), ct.byref(synd))
        return np.array(synd[0:self.m])
