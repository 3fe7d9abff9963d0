// f16_probe.hip -- three facts the fp16x2 arithmetic rests on, checked on the device:
//  (1) v_mfma_f32_32x32x16_f16 multiplies fp16 SUBNORMAL inputs exactly (no flush): the low term of a two-term split is often one;
//  (2) v_cvt_pk_f16_f32 rounds to nearest even and produces subnormals; v_fma_mix_f32 forms y - (float)h exactly;
//  (3) the two-term split x s = h0 + h1 + e with |e| <= max(2^-25, 2^-24 |x s|) (scaled units).
//   hipcc --offload-arch=gfx950 -O3 -o f16_probe tools/f16_probe.hip && ./f16_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float resid_lo(unsigned a, float y) { float r; asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(a), "v"(y)); return r; }
__device__ __forceinline__ float resid_hi(unsigned a, float y) { float r; asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(a), "v"(y)); return r; }
__device__ __forceinline__ float sum_lo(unsigned a, unsigned b) { float r; asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel_hi:[1,0,1]" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float sum_hi(unsigned a, unsigned b) { float r; asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(a), "v"(b)); return r; }

__global__ void mfma_denorm(float* out, float av, float bv)
{
    f16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)0.0f; b[j] = (_Float16)0.0f; }
    a[0] = (_Float16)av; b[0] = (_Float16)bv;        // only k = 0 (lanes 0..31) and k = 8 (lanes 32..63) contribute
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    if (threadIdx.x == 0) { out[0] = acc[0]; out[1] = (float)a[0]; out[2] = (float)b[0]; }
}

__global__ void split_check(const float* x, int n, float s, float* h0o, float* h1o, float* recon)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 >= n) return;
    const float y0 = x[2 * i] * s, y1 = x[2 * i + 1] * s;
    const f32x2 v = {y0, y1};
    const unsigned a = __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2));
    const f32x2 r = {resid_lo(a, y0), resid_hi(a, y1)};
    const unsigned b = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2));
    const f16x2 ah = __builtin_bit_cast(f16x2, a), bh = __builtin_bit_cast(f16x2, b);
    h0o[2 * i] = (float)ah[0]; h0o[2 * i + 1] = (float)ah[1];
    h1o[2 * i] = (float)bh[0]; h1o[2 * i + 1] = (float)bh[1];
    recon[2 * i] = sum_lo(a, b); recon[2 * i + 1] = sum_hi(a, b);
}

int main()
{
    float* d; (void)hipMalloc(&d, 64);
    float h[3];
    const float cases[4][2] = {{3.0e-6f, 1024.0f}, {5.96e-8f, 32768.0f}, {1.0f, 2.0f}, {6.0e-5f, 6.0e-5f}};
    for (auto& c : cases) {
        hipLaunchKernelGGL(mfma_denorm, dim3(1), dim3(64), 0, 0, d, c[0], c[1]);
        (void)hipMemcpy(h, d, 12, hipMemcpyDeviceToHost);
        printf("mfma f16: a = %.9g (as f16 %.9g) x b = %.9g -> acc %.9g, exact product %.9g  %s\n", c[0], h[1], c[1], h[0], (double)h[1] * h[2],
               h[0] == (float)((double)h[1] * h[2]) ? "EXACT" : "DIFFERENT");
    }
    const int n = 1 << 20;
    std::vector<float> x(n), h0(n), h1(n), rc(n);
    srand(3);
    for (int i = 0; i < n; ++i) {
        const float u = (float)rand() / RAND_MAX, g = (float)rand() / RAND_MAX;
        x[i] = (u - 0.5f) * expf(24.f * (g - 0.5f));          // 10 decades of magnitudes
    }
    float *dx, *d0, *d1, *dr;
    (void)hipMalloc(&dx, n * 4); (void)hipMalloc(&d0, n * 4); (void)hipMalloc(&d1, n * 4); (void)hipMalloc(&dr, n * 4);
    (void)hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
    float mx = 0; for (float v : x) mx = fmaxf(mx, fabsf(v));
    const float s = exp2f(9.f - floorf(log2f(mx)));           // max lands in [2^9, 2^10)
    hipLaunchKernelGGL(split_check, dim3(n / 2 / 256), dim3(256), 0, 0, dx, n, s, d0, d1, dr);
    (void)hipMemcpy(h0.data(), d0, n * 4, hipMemcpyDeviceToHost);
    (void)hipMemcpy(h1.data(), d1, n * 4, hipMemcpyDeviceToHost);
    (void)hipMemcpy(rc.data(), dr, n * 4, hipMemcpyDeviceToHost);
    double worst = 0; int bad_rne = 0, bad_rec = 0;
    for (int i = 0; i < n; ++i) {
        const double y = (double)x[i] * s;
        const double e = fabs(y - ((double)h0[i] + (double)h1[i]));
        const double bound = fmax(ldexp(1.0, -25), ldexp(fabs(y), -24));
        worst = fmax(worst, e / bound);
        if ((float)(_Float16)(float)y != h0[i]) ++bad_rne;            // host RNE conversion of the same value
        if ((float)((double)h0[i] + (double)h1[i]) != rc[i]) ++bad_rec;
    }
    printf("split2 on %d values over 10 decades, scale 2^%d: worst |y - h0 - h1| / max(2^-25, 2^-24 |y|) = %.3f; h0 != host RNE: %d; fma_mix sum != h0 + h1: %d\n",
           n, (int)log2f(s), worst, bad_rne, bad_rec);
    return 0;
}
