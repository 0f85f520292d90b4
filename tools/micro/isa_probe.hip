// isa_probe.hip — the per-node routines of the headline kernel's loop, each wrapped in a probe kernel of its own so that
// its instruction count can be read off the disassembly (tools/isa_budget.py compiles this file to assembly with the
// product's flags and counts instructions per class between the probe's markers).  Nothing here is linked into the
// library.  The routines are the ones kernels.hip instantiates for h.txt: check-node block pairs of degree (3,3), (4,4),
// (4,3) in the shared-reciprocal form, leaf / degree-2 variable-node block pairs, the degree-15 block with its slot indices
// in registers.
#include "../../libldpc_amd/csrc/kernels.hip"

namespace ldpc_amd
{
namespace
{
#define PROBE_BEGIN asm volatile("; PROBE_BEGIN" ::: "memory")
#define PROBE_END asm volatile("; PROBE_END" ::: "memory")

__global__ void probe_cn33(double *msg, uint32_t *out)
{
    extern __shared__ double lds[];
    uint32_t esc = out[0];
    PROBE_BEGIN;
    const uint32_t p = cn_update_ratio2<3, 3, true>(lds + threadIdx.x, lds + 192 + threadIdx.x, esc);
    PROBE_END;
    out[threadIdx.x] = p ^ esc;
}
__global__ void probe_cn44(double *msg, uint32_t *out)
{
    extern __shared__ double lds[];
    uint32_t esc = out[0];
    PROBE_BEGIN;
    const uint32_t p = cn_update_ratio2<4, 4, true>(lds + threadIdx.x, lds + 256 + threadIdx.x, esc);
    PROBE_END;
    out[threadIdx.x] = p ^ esc;
}
__global__ void probe_cn43(double *msg, uint32_t *out)
{
    extern __shared__ double lds[];
    uint32_t esc = out[0];
    PROBE_BEGIN;
    const uint32_t p = cn_update_ratio2<4, 3, true>(lds + threadIdx.x, lds + 256 + threadIdx.x, esc);
    PROBE_END;
    out[threadIdx.x] = p ^ esc;
}
__global__ void probe_cn33_separate(double *msg, uint32_t *out) // round 2's form: one quotient per output
{
    extern __shared__ double lds[];
    uint32_t esc = out[0];
    PROBE_BEGIN;
    const uint32_t p = cn_update_ratio2<3, 3, false>(lds + threadIdx.x, lds + 192 + threadIdx.x, esc);
    PROBE_END;
    out[threadIdx.x] = p ^ esc;
}
__global__ void probe_cn44_separate(double *msg, uint32_t *out)
{
    extern __shared__ double lds[];
    uint32_t esc = out[0];
    PROBE_BEGIN;
    const uint32_t p = cn_update_ratio2<4, 4, false>(lds + threadIdx.x, lds + 256 + threadIdx.x, esc);
    PROBE_END;
    out[threadIdx.x] = p ^ esc;
}
__global__ void probe_vn1x2(double *msg, uint32_t *out, double la, double lb)
{
    extern __shared__ double lds[];
    uint32_t esc = out[0], ia = out[1 + threadIdx.x], ib = out[65 + threadIdx.x];
    double pa, pb;
    PROBE_BEGIN;
    vn_small_ratio2<1, true>(lds, ia, ib, la, lb, esc, pa, pb);
    PROBE_END;
    out[threadIdx.x] = esc;
}
__global__ void probe_vn2x2(double *msg, uint32_t *out, double la, double lb)
{
    extern __shared__ double lds[];
    uint32_t esc = out[0], ia = out[1 + threadIdx.x], ib = out[65 + threadIdx.x];
    double pa, pb;
    PROBE_BEGIN;
    vn_small_ratio2<2, true>(lds, ia, ib, la, lb, esc, pa, pb);
    PROBE_END;
    out[threadIdx.x] = esc;
}
__global__ void probe_vn15(double *msg, uint32_t *out, double lam)
{
    extern __shared__ double lds[];
    uint32_t esc = out[0];
    uint32_t pk[8];
    for (int i = 0; i < 8; ++i)
        pk[i] = out[1 + 64 * i + threadIdx.x];
    PROBE_BEGIN;
    const double prod = vn_update_ratio_regs<15, false, true>(lds, pk, lam, esc);
    PROBE_END;
    out[threadIdx.x] = esc ^ hi_word(prod);
}
} // namespace
} // namespace ldpc_amd
