"""PCIe-inclusive rate of the given-LLR entry (ldpc_hip_decode_batch) when the caller hands over HOST buffers:
65 536 frames x 1152 LLRs (604 MB) in, iteration counts + hard decisions (75 MB) out, versus the same call with
device-resident buffers.  (The stream entry generates its inputs on the device: nothing but 8 bytes per frame of
outputs ever crosses PCIe there.)"""
import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import numpy as np, torch, libldpc_amd
B = 65536
dec = libldpc_amd.HipDecoder("tests/golden/h.txt")
dec.stream_begin("AWGN", 0, -4.0)
llr_host = dec.stream_decode(B, want=("llr_in",))["llr_in"]          # [B][nc] float64 on the host (pageable)
for label, llr, outs in (("host pageable in / host out", llr_host, ("iters", "hard")),):
    for rep in range(3):
        t0 = time.perf_counter()
        r = dec.decode_batch(llr, want=outs)
        dt = time.perf_counter() - t0
        print(f"{label}: {B / dt / 1e6:.2f} M frames/s ({dt * 1e3:.1f} ms, {llr.nbytes / dt / 1e9:.1f} GB/s in)")
pinned = torch.from_numpy(llr_host).pin_memory()
it_p = torch.zeros(B, dtype=torch.int32).pin_memory(); hd_p = torch.zeros(B, dec.nc, dtype=torch.uint8).pin_memory()
for rep in range(3):
    t0 = time.perf_counter()
    dec.decode_batch(pinned.numpy(), want=(), out={"iters": it_p.numpy(), "hard": hd_p.numpy()})
    dt = time.perf_counter() - t0
    print(f"host pinned in / pinned out: {B / dt / 1e6:.2f} M frames/s ({dt * 1e3:.1f} ms, {pinned.numel() * 8 / dt / 1e9:.1f} GB/s in)")
dev = torch.device("cuda", 0)
llr_d = pinned.to(dev); it_d = torch.zeros(B, dtype=torch.int32, device=dev); hd_d = torch.zeros(B, dec.nc, dtype=torch.uint8, device=dev)
class P:  # minimal pointer carrier for binding._ptr
    pass
import ctypes as ct
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    dec._check(dec.lib.ldpc_hip_decode_batch(dec.ctx, libldpc_amd.binding._dec(True, 50, "BP"), B, ct.c_void_p(llr_d.data_ptr()),
               ct.byref(libldpc_amd.binding.ldpc_hip_out(it_d.data_ptr(), None, hd_d.data_ptr(), None, None, None)), None), "decode")
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"device in / device out: {B / dt / 1e6:.2f} M frames/s ({dt * 1e3:.1f} ms)")
assert np.array_equal(it_d.cpu().numpy(), r["iters"].astype(np.int32)) and np.array_equal(it_p.numpy(), r["iters"].astype(np.int32))
