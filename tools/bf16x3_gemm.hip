// bf16x3_gemm.hip — can the bf16 matrix pipe stand in for fp32 MFMA at fp32 accuracy?
// Y[M][N] = X[M][K] . W[N][K]^T computed five ways on the device and compared with an fp64 host
// reference:  (a) v_mfma_f32_32x32x2_f32 (the fp32 A/B),  (b) one bf16 MFMA (plain bf16),
// (c) three-way bf16 split of both operands, 6 product terms,  (d) all 9 terms,  (e) 6 terms with
// the five small terms in their own accumulator (what the product ships),  (f) a TWO-term fp16 split with three or four product
// terms, with and without the per-tensor power-of-two scale fp16's range needs (round 4: the numbers behind DESIGN.md section 9.1).
// Errors are reported relative to sum_k |x||w|
// (the scale rounding errors live on) in units of 2^-24.  Also times a register-resident MFMA
// loop of each flavour.  Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// two-term fp16 split: 11 + 11 significant bits; `sc` = the tensor's power-of-two scale (fp16 spans 6e-8 .. 65504 only)
__device__ __forceinline__ void split2h(float x, float sc, _Float16& a, _Float16& b)
{
    const float y = x * sc;                 // exact
    a = (_Float16)y;
    b = (_Float16)(y - (float)a);           // the residual is exact in fp32; its conversion rounds to 11 bits
}

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
                                           int M, int N, int K, float sx = 1.0f, float sw = 1.0f)
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
    } else if (MODE == 5 || MODE == 6) {      // fp16 x2: three (or four) product terms, small terms first, result unscaled at the end
        for (int kb = 0; kb < K; kb += 16) {
            f16x8 w0, w1, x0, x1;
            for (int j = 0; j < 8; ++j) {
                _Float16 a, b;
                split2h(wr[kb + 8 * h + j], sw, a, b); w0[j] = a; w1[j] = b;
                split2h(xr[kb + 8 * h + j], sx, a, b); x0[j] = a; x1[j] = b;
            }
            if (MODE == 6) lo = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1, x1, lo, 0, 0, 0);
            lo = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0, x1, lo, 0, 0, 0);
            lo = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1, x0, lo, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0, x0, acc, 0, 0, 0);
        }
        const float inv = 1.0f / (sx * sw);
        for (int i = 0; i < 16; ++i) acc[i] = (acc[i] + lo[i]) * inv;
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

// the per-tensor power-of-two scale a two-term fp16 split needs: max |v| lands in [2^13, 2^14)
static float pow2_scale(const std::vector<float>& v)
{
    float m = 0.f;
    for (float x : v) m = fmaxf(m, fabsf(x));
    return exp2f(13.f - floorf(log2f(m)));
}

static void study(const char* title, std::vector<float>& X, std::vector<float>& W, int M, int N, int K)
{
    const float sx = pow2_scale(X), sw = pow2_scale(W);
    std::vector<float> Y((size_t)M * N);
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
    printf("%s (x scale 2^%d, w scale 2^%d for the fp16 rows)\n", title, (int)log2f(sx), (int)log2f(sw));
    const char* names[9] = {"fp32 MFMA 32x32x2", "bf16 x1", "bf16 x3, 6 terms", "bf16 x3, 9 terms", "bf16 x3, 6 terms, split accumulators",
                            "fp16 x2, 3 terms, scaled", "fp16 x2, 4 terms, scaled", "fp16 x2, 3 terms, UNSCALED", "fp16 x2, 4 terms, UNSCALED"};
    for (int mode = 0; mode < 9; ++mode) {
        dim3 g(M / 32, N / 32);
        switch (mode) {
        case 0: hipLaunchKernelGGL(gemm<0>, g, dim3(64), 0, 0, dX, dW, dY, M, N, K, 1.f, 1.f); break;
        case 1: hipLaunchKernelGGL(gemm<1>, g, dim3(64), 0, 0, dX, dW, dY, M, N, K, 1.f, 1.f); break;
        case 2: hipLaunchKernelGGL(gemm<2>, g, dim3(64), 0, 0, dX, dW, dY, M, N, K, 1.f, 1.f); break;
        case 3: hipLaunchKernelGGL(gemm<3>, g, dim3(64), 0, 0, dX, dW, dY, M, N, K, 1.f, 1.f); break;
        case 4: hipLaunchKernelGGL(gemm<4>, g, dim3(64), 0, 0, dX, dW, dY, M, N, K, 1.f, 1.f); break;
        case 5: hipLaunchKernelGGL(gemm<5>, g, dim3(64), 0, 0, dX, dW, dY, M, N, K, sx, sw); break;
        case 6: hipLaunchKernelGGL(gemm<6>, g, dim3(64), 0, 0, dX, dW, dY, M, N, K, sx, sw); break;
        case 7: hipLaunchKernelGGL(gemm<5>, g, dim3(64), 0, 0, dX, dW, dY, M, N, K, 1.f, 1.f); break;
        default: hipLaunchKernelGGL(gemm<6>, g, dim3(64), 0, 0, dX, dW, dY, M, N, K, 1.f, 1.f); break;
        }
        (void)hipMemcpy(Y.data(), dY, Y.size() * 4, hipMemcpyDeviceToHost);
        double emax = 0, esum = 0;
        for (size_t i = 0; i < Y.size(); ++i) {
            const double e = fabs((double)Y[i] - ref[i]) / scale[i] * 16777216.0;
            emax = e > emax ? e : emax; esum += e;
        }
        printf("  %-40s error / sum|x||w| in units of 2^-24: max %.3f mean %.4f\n", names[mode], emax, esum / Y.size());
    }
    (void)hipFree(dX); (void)hipFree(dW); (void)hipFree(dY);
}

int main()
{
    const int M = 2048, N = 128, K = 256;
    std::vector<float> X((size_t)M * K), W((size_t)N * K);
    srand(1);
    auto rnd = []() { return (float)rand() / RAND_MAX; };
    auto gauss = [&]() { return sqrtf(-2.f * logf(rnd() + 1e-12f)) * cosf(6.2831853f * rnd()); };
    for (auto& v : X) { float g = gauss(); v = g > 0 ? g : expf(g) - 1.f; }   // ELU(randn): a forward layer's input
    for (auto& v : W) v = (2.f * rnd() - 1.f) / 16.f;
    study("forward GEMM: activations ELU(randn) x weights U(-1/16, 1/16)", X, W, M, N, K);
    // a weight-gradient product: "x" = dZ rows as the update sees them (1 / 40960 of a unit gradient, a few orders of spread),
    // "w" = activations; fp16 needs the per-tensor scale here: unscaled, dZ sits in fp16's subnormals
    for (auto& v : X) v = gauss() * expf(2.f * gauss()) * (1.f / 40960.f) * 0.05f;
    for (auto& v : W) { float g = gauss(); v = g > 0 ? g : expf(g) - 1.f; }
    study("weight-gradient-like GEMM: dZ ~ 1e-6 with two decades of spread x activations", X, W, M, N, K);
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
