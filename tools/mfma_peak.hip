// mfma_peak.hip — what does v_mfma_f32_32x32x2_f32 sustain on this chip?  Pure register-resident MFMA
// loops, WAVES waves per workgroup, one workgroup per CU slot; variants: dependent chain on one
// accumulator, two and four independent accumulators.  Build: hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b)
{
    f32x16 acc[NACC];
    for (int t = 0; t < NACC; ++t)
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int t = 0; t < NACC; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
    }
    float s = 0.f;
    for (int t = 0; t < NACC; ++t)
        for (int i = 0; i < 16; ++i) s += acc[t][i];
    if (s == 12345.f) out[0] = s;
}

template <int NACC>
void run(int wgs, int threads, const char* name)
{
    float* out;
    hipMalloc(&out, 4);
    const int iters = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NACC>, dim3(wgs), dim3(threads), 0, 0, out, 10, 1.f, 1.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<NACC>, dim3(wgs), dim3(threads), 0, 0, out, iters, 1.f, 1.f);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)wgs * (threads / 64) * iters * 8.0 * NACC * 4096.0;
    printf("%-28s wgs %5d threads %4d: %.3f ms, %.1f TFLOP/s\n", name, wgs, threads, ms, flop / ms / 1e9);
    hipFree(out);
}

int main()
{
    run<1>(256, 256, "1 acc chain, 1 wave/SIMD");
    run<2>(256, 256, "2 acc, 1 wave/SIMD");
    run<4>(256, 256, "4 acc, 1 wave/SIMD");
    run<1>(512, 256, "1 acc chain, 2 waves/SIMD");
    run<1>(768, 256, "1 acc chain, 3 waves/SIMD");
    run<2>(768, 256, "2 acc, 3 waves/SIMD");
    run<1>(1024, 256, "1 acc chain, 4 waves/SIMD");
    return 0;
}
