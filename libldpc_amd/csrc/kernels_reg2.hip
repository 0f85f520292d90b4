// kernels_reg2.hip — register-resident decoder, totals form (kernels_reg2_impl.hpp): the generic instantiations and the
// dispatcher.  The regular-code instantiations live in kernels_reg2u.hip.
#include "kernels_reg2_impl.hpp"

namespace ldpc_amd
{

int launch_decode_reg2_regular(const DecodeArgs &a, const DevReg2Plan &r, bool min_sum, void *stream); // kernels_reg2u.hip

int launch_decode_reg2(const DecodeArgs &a, const DevReg2Plan &r, bool min_sum, void *stream)
{
    if (a.n_frames == 0)
        return hipSuccess;
    if (!a.ws_llr || !a.ws_hb || !a.ws_scr) // ws_scr: [n_frames][(nv0 + nv1) * nt] channel terms
        return hipErrorInvalidValue;
    if (r.nt == 1024 && r.kc == 4 && r.maxd == 6 && r.nv0 == 4 && r.nv1 == 4)
        return r.uniform_cn && r.uniform_vn ? launch_decode_reg2_regular(a, r, min_sum, stream)
                                            : launch_reg2<1024, 4, 6, 4, 4, false>(a, r, min_sum, stream);
    // (512 threads x 256 registers, launch_reg2<512, 8, 6, 8, 8>, was measured 1.25x slower on the n=8192 code)
    return hipErrorInvalidValue;
}

} // namespace ldpc_amd
