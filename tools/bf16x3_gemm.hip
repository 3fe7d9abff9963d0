// bf16x3_gemm.hip — can the bf16 matrix pipe stand in for fp32 MFMA at fp32 accuracy?
// Y[M][N] = X[M][K] . W[N][K]^T computed five ways on the device and compared with an fp64 host
// reference:  (a) v_mfma_f32_32x32x2_f32 (what the product ships),  (b) one bf16 MFMA (plain bf16),
// (c) three-way bf16 split of both operands, 6 product terms,  (d) all 9 terms,  (e) 6 terms with
// the five small terms in their own accumulator.  Errors are reported relative to sum_k |x||w|
// (the scale rounding errors live on) in units of 2^-24.  Also times a register-resident MFMA
// loop of each flavour.  Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split3(float x, __bf16& a, __bf16& b, __bf16& c)
{
    a = (__bf16)x;
    const float r1 = x - (float)a;          // exact
    b = (__bf16)r1;
    const float r2 = r1 - (float)b;         // exact
    c = (__bf16)r2;
}

// one wave per 32x32 output tile; lane (r = lane&31, h = lane>>5)
template <int MODE>
__global__ __launch_bounds__(64) void gemm(const float* __restrict__ X, const float* __restrict__ W, float* __restrict__ Y,
                                           int M, int N, int K)
{
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    const int mt = blockIdx.x, nt = blockIdx.y;
    f32x16 acc, lo;
    for (int i = 0; i < 16; ++i) { acc[i] = 0.f; lo[i] = 0.f; }
    const float* xr = X + (long)(mt * 32 + r) * K;      // B operand: batch row r
    const float* wr = W + (long)(nt * 32 + r) * K;      // A operand: output column r
    if (MODE == 0) {
        for (int k = 0; k < K; k += 2)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wr[k + h], xr[k + h], acc, 0, 0, 0);
    } else {
        for (int kb = 0; kb < K; kb += 16) {
            bf16x8 w0, w1, w2, x0, x1, x2;
            for (int j = 0; j < 8; ++j) {
                __bf16 a, b, c;
                split3(wr[kb + 8 * h + j], a, b, c); w0[j] = a; w1[j] = b; w2[j] = c;
                split3(xr[kb + 8 * h + j], a, b, c); x0[j] = a; x1[j] = b; x2[j] = c;
            }
            if (MODE == 1) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, x0, acc, 0, 0, 0);
            } else if (MODE == 2 || MODE == 3) {
                if (MODE == 3) {
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2, x2, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, x2, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2, x1, acc, 0, 0, 0);
                }
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, x2, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2, x0, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, x1, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, x1, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, x0, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, x0, acc, 0, 0, 0);
            } else {
                lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, x2, lo, 0, 0, 0);
                lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2, x0, lo, 0, 0, 0);
                lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, x1, lo, 0, 0, 0);
                lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, x1, lo, 0, 0, 0);
                lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, x0, lo, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, x0, acc, 0, 0, 0);
            }
        }
        if (MODE == 4) for (int i = 0; i < 16; ++i) acc[i] += lo[i];
    }
    // C layout: A rows = output columns: n = (reg&3) + 8*(reg>>2) + 4*h, batch row = r
    for (int reg = 0; reg < 16; ++reg) {
        const int n = (reg & 3) + 8 * (reg >> 2) + 4 * h;
        Y[(long)(mt * 32 + r) * N + nt * 32 + n] = acc[reg];
    }
}

template <int NMF>
__global__ __launch_bounds__(256) void rate_bf16(float* out, int iters)
{
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)1.0f; b[j] = (__bf16)1.0f; }
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int u = 0; u < NMF; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc[i];
    if (s == 12345.f) out[0] = s;
}

int main()
{
    const int M = 2048, N = 128, K = 256;
    std::vector<float> X((size_t)M * K), W((size_t)N * K), Y((size_t)M * N);
    srand(1);
    auto rnd = []() { return (float)rand() / RAND_MAX; };
    for (auto& v : X) { float g = sqrtf(-2.f * logf(rnd() + 1e-12f)) * cosf(6.2831853f * rnd()); v = g > 0 ? g : expf(g) - 1.f; }   // ELU(randn)
    for (auto& v : W) v = (2.f * rnd() - 1.f) / 16.f;
    std::vector<double> ref((size_t)M * N), scale((size_t)M * N);
    for (int m = 0; m < M; ++m)
        for (int n = 0; n < N; ++n) {
            double s = 0, a = 0;
            for (int k = 0; k < K; ++k) { s += (double)X[(size_t)m * K + k] * W[(size_t)n * K + k]; a += fabs((double)X[(size_t)m * K + k] * W[(size_t)n * K + k]); }
            ref[(size_t)m * N + n] = s; scale[(size_t)m * N + n] = a;
        }
    float *dX, *dW, *dY;
    (void)hipMalloc(&dX, X.size() * 4); (void)hipMalloc(&dW, W.size() * 4); (void)hipMalloc(&dY, Y.size() * 4);
    (void)hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice);
    const char* names[5] = {"fp32 MFMA 32x32x2", "bf16 x1", "bf16 x3, 6 terms", "bf16 x3, 9 terms", "bf16 x3, 6 terms, split accumulators"};
    for (int mode = 0; mode < 5; ++mode) {
        dim3 g(M / 32, N / 32);
        switch (mode) {
        case 0: hipLaunchKernelGGL(gemm<0>, g, dim3(64), 0, 0, dX, dW, dY, M, N, K); break;
        case 1: hipLaunchKernelGGL(gemm<1>, g, dim3(64), 0, 0, dX, dW, dY, M, N, K); break;
        case 2: hipLaunchKernelGGL(gemm<2>, g, dim3(64), 0, 0, dX, dW, dY, M, N, K); break;
        case 3: hipLaunchKernelGGL(gemm<3>, g, dim3(64), 0, 0, dX, dW, dY, M, N, K); break;
        default: hipLaunchKernelGGL(gemm<4>, g, dim3(64), 0, 0, dX, dW, dY, M, N, K); break;
        }
        (void)hipMemcpy(Y.data(), dY, Y.size() * 4, hipMemcpyDeviceToHost);
        double emax = 0, esum = 0;
        for (size_t i = 0; i < Y.size(); ++i) {
            const double e = fabs((double)Y[i] - ref[i]) / scale[i] * 16777216.0;
            emax = e > emax ? e : emax; esum += e;
        }
        printf("%-40s error / sum|x||w| in units of 2^-24: max %.3f mean %.4f\n", names[mode], emax, esum / Y.size());
    }
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float* out; (void)hipMalloc(&out, 4);
    hipLaunchKernelGGL(rate_bf16<8>, dim3(768), dim3(256), 0, 0, out, 10);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(rate_bf16<8>, dim3(768), dim3(256), 0, 0, out, 20000);
    (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double fl = 768.0 * 4 * 20000 * 8 * 32768.0;
    printf("v_mfma_f32_32x32x16_bf16 register loop, 3 waves/SIMD: %.1f TFLOP/s (%.1f cycles per MFMA at 2.4 GHz per SIMD)\n", fl / ms / 1e9,
           ms * 1e-3 * 2.4e9 / (3.0 * 20000 * 8));
    return 0;
}
