// kernels_reg2u.hip — register-resident decoder, totals form (kernels_reg2_impl.hpp): the instantiations for regular
// codes (every check-node block of full degree, every variable-node block full and of degree 3).
#include "kernels_reg2_impl.hpp"

namespace ldpc_amd
{

int launch_decode_reg2_regular(const DecodeArgs &a, const DevReg2Plan &r, bool min_sum, void *stream)
{
    return launch_reg2<1024, 4, 6, 4, 4, true>(a, r, min_sum, stream);
}

} // namespace ldpc_amd
