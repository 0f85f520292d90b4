// selftest.hip — the device arithmetic of detmath.h / device_cn.hpp evaluated element by element, so that tests can
// hold it against values computed by glibc's libm and by extended-precision host arithmetic (tests/test_gpu_math.py)
// instead of against the same header compiled for the host.
#include <hip/hip_runtime.h>

#include "device_cn.hpp"
#include "device_math.hpp"
#include "kernels.hpp"

namespace ldpc_amd
{

namespace
{

template <int D>
__device__ void cn_ratio_rows(uint64_t i, const double *a, double *out)
{
    double v[D];
#pragma unroll
    for (int j = 0; j < D; ++j)
        v[j] = a[i * D + j];
    cn_ratio<D>(v);
#pragma unroll
    for (int j = 0; j < D; ++j)
        out[i * D + j] = v[j];
}

template <int D>
__device__ void cn_shared_rows(uint64_t i, const double *a, double *out)
{
    double v[D];
#pragma unroll
    for (int j = 0; j < D; ++j)
        v[j] = a[i * D + j];
    uint32_t escaped = 0;
    cn_ratio<D, D != 6, D == 6>(v, &escaped, &escaped);
#pragma unroll
    for (int j = 0; j < D; ++j)
        out[i * D + j] = DM_RATIO_ESCAPED(escaped) ? __builtin_nan("") : v[j];
}

template <int D>
__device__ void cn_llr_rows(uint64_t i, const double *a, double *out)
{
    double v[D];
#pragma unroll
    for (int j = 0; j < D; ++j)
        v[j] = a[i * D + j];
    cn_core<D, false>(v);
#pragma unroll
    for (int j = 0; j < D; ++j)
        out[i * D + j] = v[j];
}

__global__ __launch_bounds__(256) void math_selftest_kernel(int fn, uint64_t n, const double *a, const double *b, double *out)
{
    const uint64_t i = blockIdx.x * 256ull + threadIdx.x;
    if (i >= n)
        return;
    switch (fn)
    {
    case kMathExp: out[i] = dm_exp(a[i]); break;
    case kMathLog: out[i] = dm_log(a[i]); break;
    case kMathBoxplus: out[i] = box_jacobian(a[i], b[i]); break; // what the kernels call: dm_boxplus + its short cut
    case kMathRatioDiv: out[i] = dm_ratio_div(a[i], b[i]); break;
    case kMathRatioRho: out[i] = dm_ratio_rho(a[i], b[i]); break;
    case kMathRatioLambda: out[i] = dm_ratio_lambda(a[i], b[i]); break;
    case kMathECombine: out[i] = dm_e_combine(a[i], b[i]); break;
    case kMathExpClamped: out[i] = dm_exp_clamped(a[i]); break;
    case kMathBoxplusExp: out[i] = dm_boxplus_exp(a[i]); break;
    case kMathBoxplusLog: out[i] = dm_boxplus_log(a[i]); break;
    case kMathCnRatio3: cn_ratio_rows<3>(i, a, out); break;
    case kMathCnRatio4: cn_ratio_rows<4>(i, a, out); break;
    case kMathCnRatio5: cn_ratio_rows<5>(i, a, out); break;
    case kMathCnRatio6: cn_ratio_rows<6>(i, a, out); break;
    case kMathCnRatio8: cn_ratio_rows<8>(i, a, out); break;
    case kMathCnLlr4: cn_llr_rows<4>(i, a, out); break;
    case kMathCnLlr6: cn_llr_rows<6>(i, a, out); break;
    case kMathCnRatio3s: cn_shared_rows<3>(i, a, out); break;
    case kMathCnRatio4s: cn_shared_rows<4>(i, a, out); break;
    case kMathCnRatio6s: cn_shared_rows<6>(i, a, out); break;
    default: break;
    }
}

} // namespace

int math_selftest_width(int fn)
{
    switch (fn)
    {
    case kMathCnRatio3: case kMathCnRatio3s: return 3;
    case kMathCnRatio4: case kMathCnLlr4: case kMathCnRatio4s: return 4;
    case kMathCnRatio5: return 5;
    case kMathCnRatio6: case kMathCnLlr6: case kMathCnRatio6s: return 6;
    case kMathCnRatio8: return 8;
    default: return fn >= 0 && fn < kMathCount ? 1 : 0;
    }
}

int launch_math_selftest(int fn, uint64_t n, const double *a, const double *b, double *out, void *stream)
{
    if (n == 0)
        return hipSuccess;
    const unsigned blocks = static_cast<unsigned>((n + 255) / 256);
    hipLaunchKernelGGL(math_selftest_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), fn, n, a, b, out);
    return hipGetLastError();
}

} // namespace ldpc_amd
