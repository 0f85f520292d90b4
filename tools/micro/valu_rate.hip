// Issue cost of the VALU instructions the decoders lean on, in shader cycles per wave64 instruction per SIMD, measured
// with 4 resident waves per SIMD and 8 independent chains per wave (so neither latency nor occupancy hides the rate).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

template <int OP>
__global__ __launch_bounds__(256) void rate_kernel(double *out, uint64_t *cycles, int iters, double seed)
{
    double x[8];
    for (int k = 0; k < 8; ++k)
        x[k] = seed + 0.001 * k + 1e-6 * threadIdx.x;
    float f[8];
    for (int k = 0; k < 8; ++k)
        f[k] = (float)x[k];
    uint32_t u[8];
    for (int k = 0; k < 8; ++k)
        u[k] = threadIdx.x * 2654435761u + k;
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i)
    {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep)
#pragma unroll
            for (int k = 0; k < 8; ++k)
            {
                if constexpr (OP == 0) asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(x[k]));
                if constexpr (OP == 1) asm volatile("v_mul_f64 %0, %0, %0" : "+v"(x[k]));
                if constexpr (OP == 2) asm volatile("v_rcp_f64 %0, %0" : "+v"(x[k]));
                if constexpr (OP == 3) asm volatile("v_rcp_f32 %0, %0" : "+v"(f[k]));
                if constexpr (OP == 4) asm volatile("v_add_f64 %0, %0, %0" : "+v"(x[k]));
                if constexpr (OP == 5) asm volatile("v_ldexp_f64 %0, %0, 1" : "+v"(x[k]));
                if constexpr (OP == 6) asm volatile("v_min_u32 %0, %0, %0" : "+v"(u[k]));
                if constexpr (OP == 7) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(f[k]));
                if constexpr (OP == 8) asm volatile("v_min_f64 %0, %0, %0" : "+v"(x[k]));
                if constexpr (OP == 9) asm volatile("v_rndne_f64 %0, %0" : "+v"(x[k]));
                if constexpr (OP == 10) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(u[k]) : "v"(x[k]));
                if constexpr (OP == 11) asm volatile("v_exp_f32 %0, %0" : "+v"(f[k]));
                if constexpr (OP == 12) asm volatile("v_cmp_le_f64 vcc, %0, %0" : : "v"(x[k]) : "vcc");
                if constexpr (OP == 13) asm volatile("v_alignbit_b32 %0, %0, %0, 15" : "+v"(u[k]));
                if constexpr (OP == 14) asm volatile("v_mov_b64 %0, %0" : "+v"(x[k]));
            }
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    double s = 0;
    for (int k = 0; k < 8; ++k)
        s += x[k] + f[k] + u[k];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0)
        cycles[blockIdx.x] = t1 - t0;
}

int main()
{
    const char *names[] = {"v_fma_f64", "v_mul_f64", "v_rcp_f64", "v_rcp_f32", "v_add_f64", "v_ldexp_f64", "v_min_u32", "v_fma_f32",
                           "v_min_f64", "v_rndne_f64", "v_cvt_i32_f64", "v_exp_f32", "v_cmp_le_f64", "v_alignbit_b32", "v_mov_b64"};
    void (*ks[])(double *, uint64_t *, int, double) = {rate_kernel<0>, rate_kernel<1>, rate_kernel<2>, rate_kernel<3>, rate_kernel<4>,
                                                         rate_kernel<5>, rate_kernel<6>, rate_kernel<7>, rate_kernel<8>, rate_kernel<9>,
                                                         rate_kernel<10>, rate_kernel<11>, rate_kernel<12>, rate_kernel<13>, rate_kernel<14>};
    const int blocks = 256 * 4, iters = 2000; // 4 workgroups of 4 waves per CU: 4 waves per SIMD
    double *out;
    uint64_t *cyc;
    hipMalloc(&out, blocks * 256 * 8);
    hipMalloc(&cyc, blocks * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    for (int op = 0; op < 15; ++op)
    {
        hipLaunchKernelGGL(ks[op], dim3(blocks), dim3(256), 0, 0, out, cyc, 10, 1.5);
        hipEventRecord(e0);
        hipLaunchKernelGGL(ks[op], dim3(blocks), dim3(256), 0, 0, out, cyc, iters, 1.5);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        std::vector<uint64_t> h(blocks);
        hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
        double avg = 0;
        for (auto c : h)
            avg += c;
        avg /= blocks;
        // per SIMD: 4 waves x iters x 32 instructions
        const double n_inst_per_simd = 4.0 * iters * 32;
        std::printf("%-16s kernel %.3f ms   %.2f ns per wave-instruction per SIMD  (= %.1f cycles at 2.4 GHz);  s_memtime-based: %.1f counts per instruction-slot\n",
                    names[op], ms, ms * 1e6 / n_inst_per_simd, ms * 1e6 / n_inst_per_simd * 2.4, avg / (iters * 32.0) / 4.0 * 4.0);
    }
    return 0;
}
