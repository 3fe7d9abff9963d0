// Does vector work issued between two v_mfma_f32_32x32x16_bf16 of ONE wave hide under the matrix pipe -- and does it depend on
// where the MFMA's accumulator lives (VGPRs vs AGPRs)?  One wave per SIMD (256-thread workgroup, one per CU), a dependent chain
// of MFMAs on one accumulator, F independent v_fma_f32 behind each MFMA.  Everything is inline asm, so the two forms differ in
// the register class of the accumulator only.  Prints cycles per MFMA (s_memtime) for F = 0 .. 8.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_fill_forms tools/mfma_fill_forms.hip && ./mfma_fill_forms
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// filler kinds: 0 v_fma_f32, 1 v_pk_add_f32, 2 v_exp_f32, 3 v_cvt_pk_bf16_f32, 4 v_cmp + v_cndmask (counted as two), 5 ds_write_b64,
// 6 v_lshlrev + v_and (bf16 pair -> two floats), 7 v_fma_f32 as ONE dependent chain (every filler of a gap reads the previous one's
// result: what an epilogue slice looks like), 8 two interleaved dependent chains
template <int F, bool AGPR, int KIND = 0>
__global__ __launch_bounds__(256) void k(unsigned long long* out, int iters, float a)
{
    __shared__ unsigned long long lds[2048];
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    bf16x8 x, y;
    for (int i = 0; i < 8; ++i) { x[i] = (__bf16)(a + i); y[i] = (__bf16)(a - i); }
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = a + i;
    const float b = a * 0.5f;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 pk[4];
    unsigned w[4] = {0, 0, 0, 0};
    for (int i = 0; i < 4; ++i) { pk[i][0] = a + i; pk[i][1] = a - i; }
    if (iters < 0) lds[threadIdx.x] = 0;
    if (AGPR) asm volatile("" : "+a"(acc)); else asm volatile("" : "+v"(acc));
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (AGPR) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(x), "v"(y));
            else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(x), "v"(y));
#pragma unroll
            for (int j = 0; j < F; ++j) {
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j & 7]) : "v"(b), "v"(a));
                if (KIND == 1) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(pk[j & 3]) : "v"(pk[(j + 1) & 3]));
                if (KIND == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(v[j & 7]));
                if (KIND == 3) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w[j & 3]) : "v"(v[j & 7]), "v"(v[(j + 1) & 7]));
                if (KIND == 4 && (j & 1) == 0) asm volatile("v_cmp_lt_f32 vcc, 0, %1\n\ts_nop 1\n\tv_cndmask_b32 %0, %1, %2, vcc" : "+v"(v[j & 7]) : "v"(b), "v"(a) : "vcc");
                if (KIND == 5) asm volatile("ds_write_b64 %0, %1" : : "v"((threadIdx.x * 8 + 2048 * (j & 3)) & 16383), "v"(pk[j & 3]) : "memory");
                if (KIND == 6) asm volatile("v_lshlrev_b32 %0, 16, %2\n\tv_and_b32 %1, 0xffff0000, %2" : "=v"(w[j & 3]), "=v"(w[(j + 1) & 3]) : "v"(v[j & 7]));
                if (KIND == 7) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[0]) : "v"(b), "v"(a));
                if (KIND == 8) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j & 1]) : "v"(b), "v"(a));
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (AGPR) asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" : "+a"(acc)); else asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" : "+v"(acc));
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc[i];
    for (int i = 0; i < 8; ++i) s += v[i];
    for (int i = 0; i < 4; ++i) s += pk[i][0] + pk[i][1] + (float)w[i];
    if (KIND == 5) { asm volatile("s_waitcnt lgkmcnt(0)" : : : "memory"); s += (float)lds[threadIdx.x]; }
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (s == 12345.f) out[0] = 0;
}

template <int F, bool AGPR, int KIND = 0>
void run()
{
    const int wgs = 256, iters = 2000;
    unsigned long long* out;
    (void)hipMalloc(&out, wgs * 8);
    hipLaunchKernelGGL((k<F, AGPR, KIND>), dim3(wgs), dim3(256), 0, 0, out, 10, 1.0f);
    hipLaunchKernelGGL((k<F, AGPR, KIND>), dim3(wgs), dim3(256), 0, 0, out, iters, 1.0f);
    (void)hipDeviceSynchronize();
    unsigned long long h[256];
    (void)hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost);
    double mean = 0;
    for (int i = 0; i < wgs; ++i) mean += (double)h[i];
    mean /= wgs;
    static const char* kinds[] = {"v_fma_f32", "v_pk_add_f32", "v_exp_f32", "v_cvt_pk_bf16_f32", "(v_cmp + s_nop 1 + v_cndmask)/2", "ds_write_b64", "v_lshlrev + v_and",
                                  "v_fma_f32 (ONE dependent chain)", "v_fma_f32 (two dependent chains)"};
    printf("accumulator in %s, %d x %s behind each MFMA: %.1f clock ticks per MFMA\n", AGPR ? "AGPRs" : "VGPRs", F, kinds[KIND], mean / (iters * 8.0));
    (void)hipFree(out);
}

int main()
{
    run<0, false>(); run<2, false>(); run<4, false>(); run<5, false>(); run<6, false>(); run<8, false>();
    run<0, true>(); run<2, true>(); run<4, true>(); run<5, true>(); run<6, true>(); run<8, true>();
    run<4, false, 1>(); run<8, false, 1>(); run<2, false, 2>(); run<4, false, 2>(); run<4, false, 3>(); run<8, false, 3>();
    run<4, false, 4>(); run<8, false, 4>(); run<2, false, 5>(); run<4, false, 5>(); run<2, false, 6>(); run<4, false, 6>();
    run<4, true, 1>(); run<8, true, 1>(); run<4, true, 2>(); run<8, true, 3>(); run<8, true, 4>(); run<4, true, 5>(); run<4, true, 6>();
    run<2, true, 7>(); run<3, true, 7>(); run<4, true, 7>(); run<5, true, 7>(); run<6, true, 7>(); run<8, true, 7>();
    run<4, true, 8>(); run<6, true, 8>(); run<8, true, 8>();
    return 0;
}
