// Do VALU instructions overlap with v_mfma_f32_32x32x2_f32 on one SIMD?  Each iteration issues 8 MFMAs
// (512 matrix-pipe cycles) plus V independent v_fma_f32 (4 cycles each for a wave64) and,
// optionally, D ds_read_b128.  If VALU/LDS work hides under the MFMA passes the time per iteration
// stays ~512 cycles until V*4 approaches 512.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int V, int D>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b)
{
    __shared__ float4 lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 256) lds[i] = make_float4(a, b, a, b);
    __syncthreads();
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = a + i;
    float4 dsum = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4* lp = lds + (threadIdx.x & 63);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
#pragma unroll
            for (int j = 0; j < V / 8; ++j) v[j & 7] = __builtin_fmaf(v[j & 7], b, a);
            if (u < D) {
                const float4 t = lp[((it + u) & 7) * 64];
                dsum.x += t.x;
            }
        }
    }
    float s = dsum.x;
    for (int i = 0; i < 16; ++i) s += acc[i];
    for (int i = 0; i < 8; ++i) s += v[i];
    if (s == 12345.f) out[0] = s;
}

template <int V, int D>
void run(int wgs, const char* name)
{
    float* out;
    (void)hipMalloc(&out, 4);
    const int iters = 4000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<V, D>), dim3(wgs), dim3(256), 0, 0, out, 10, 1.f, 1.f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<V, D>), dim3(wgs), dim3(256), 0, 0, out, iters, 1.0001f, 0.9999f);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)wgs * 4 * iters * 8.0 * 4096.0;
    printf("%-10s V=%3d valu/iter D=%d ds_read/iter  wgs %4d: %.3f ms  MFMA rate %.1f TFLOP/s\n", name, V, D, wgs, ms, flop / ms / 1e9);
    (void)hipFree(out);
}

int main()
{
    run<0, 0>(256, "1w/SIMD");
    run<32, 0>(256, "1w/SIMD");
    run<64, 0>(256, "1w/SIMD");
    run<128, 0>(256, "1w/SIMD");
    run<0, 0>(768, "3w/SIMD");
    run<32, 0>(768, "3w/SIMD");
    run<64, 0>(768, "3w/SIMD");
    run<128, 0>(768, "3w/SIMD");
    run<0, 2>(768, "3w/SIMD");
    run<0, 8>(768, "3w/SIMD");
    run<32, 2>(768, "3w/SIMD");
    return 0;
}
