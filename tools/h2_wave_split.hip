// h2_wave_split.hip -- would TWO waves per SIMD (8 waves on one 32-row tile, the K range of a chain GEMM split over a wave pair, the
// partial sums exchanged through LDS) shorten a chain phase of mlp_fused_step_h2_kernel?  One wave per SIMD has nobody to run an
// epilogue under; the question is what the exchange and the second barrier cost against what the halved epilogue buys.
// A chain phase as the kernel runs it (layer 2: K = 256 from the H1 planes, 128 output columns, weights streamed from L2 by every CU,
// ELU epilogue -> scaled two-term split -> the next planes, barrier), built from the kernel's own pieces (mlp_fused_h2.inc), looped:
//   A  4 waves: wave w = column tile w, full K (two K = 128 operands), epilogue on its 16 values per lane
//   B  8 waves: wave (p = w & 3, hh = w >> 2) = column tile p, K half hh; each wave sends the 8 sums it does not finish to its partner
//      through LDS (2 x 16 bytes per lane), barrier, adds the partner's 8 to its own, epilogue on 8 values per lane, barrier
// Prints shader cycles per phase (s_memtime, median over workgroups) on 256 workgroups.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -I include -I fly_bproject_amd/csrc -o h2_wave_split tools/h2_wave_split.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include "flyhip.h"
#include "mlp_layout.h"
namespace {
#include "mlp_gemm.inc"
#include "fs_stamp.inc"
#include "mlp_fused_step.inc"
#include "mlp_fused_h2.inc"

constexpr int WS_LDS_HALVES = 2 * (2 * BM * FS_P1) + 2 * BM * FS_P2;        // two H1-sized plane sets (ping-pong) + one H2-sized
constexpr int WS_XCH_FLOATS = 8 * 2 * 64 * 4;                                 // exchange: [wave][2][lane] float4

// PARTS (4-wave form only): 0 = the whole phase, 1 = GEMM only, 2 = GEMM + epilogue (no barrier), 3 = epilogue + barrier only (no GEMM)
template <int WAVES, int PARTS = 0>
__global__ __launch_bounds__(64 * WAVES, WAVES / 4) void phase_kernel(const u16* __restrict__ PH, const float* __restrict__ bias_g,
                                                                      unsigned long long* out, int phases, float k, float s)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char raw[];
    u16* lds = reinterpret_cast<u16*>(raw);
    u16* A = lds;                                   // [32][256] planes: the phase's input
    u16* B = lds + 2 * BM * FS_P1;                  // its output goes to columns 0 .. 127 of this set (the other 128 keep their values)
    float* xch = reinterpret_cast<float*>(lds + WS_LDS_HALVES);
    float* bias = xch + WS_XCH_FLOATS;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 2 * (2 * BM * FS_P1); i += 64 * WAVES) lds[i] = (u16)(0x2c00 + ((i * 2654435761u) >> 22));   // fp16 0.06 .. 0.12, varied bits
    if (tid < 128) bias[tid] = bias_g[tid];
    __syncthreads();
    float am = 0.0f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int ph = 0; ph < phases; ++ph) {
        u16* in = (ph & 1) ? B : A;
        u16* outp = (ph & 1) ? A : B;
        f32x16 hi, lo;
        if (WAVES == 4) {
            WeightHead2 w2a, w2b;
            h2_gemm_prefetch<MLP_H2>(w2a, PH + H2_OFF_PH2, wave, lane);
            h2_gemm_prefetch<MLP_H2>(w2b, PH + H2_OFF_PH2 + 2 * MLP_H2 * (MLP_H1 / 2), wave, lane);
            if (PARTS != 3) {
                h2_gemm<MLP_H2, FS_P1, false>(w2a, PH + H2_OFF_PH2, wave, in, 0, hi, lo, lane);
                h2_gemm<MLP_H2, FS_P1, false, false>(w2b, PH + H2_OFF_PH2 + 2 * MLP_H2 * (MLP_H1 / 2), wave, in, MLP_H1 / 2, hi, lo, lane);
            } else {
                for (int i = 0; i < 16; ++i) { hi[i] = am + i; lo[i] = 0.001f * i; }
            }
            if (PARTS != 1) h2_epilogue_elu<FS_P1, false>(hi, lo, bias, 32 * wave, outp, lane, nullptr, 32, k, s, am);
            else am += hi[3] + lo[5];
            if (PARTS == 0 || PARTS == 3) __syncthreads();
        } else {
            const int p = wave & 3, hh = wave >> 2;
            WeightHead2 w2;
            const u16* Wh = PH + H2_OFF_PH2 + hh * (2 * MLP_H2 * (MLP_H1 / 2));
            h2_gemm_prefetch<MLP_H2>(w2, Wh, p, lane);
            h2_gemm<MLP_H2, FS_P1, false>(w2, Wh, p, in, hh * (MLP_H1 / 2), hi, lo, lane);
            // the 8 sums (register groups 2 (1 - hh), 2 (1 - hh) + 1) the PARTNER finishes go to it through LDS
            float4* mine = reinterpret_cast<float4*>(xch) + (wave * 2) * 64 + lane;
            const int gs = 2 * (1 - hh), gk = 2 * hh;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int g = gs + j;
                mine[j * 64] = make_float4(hi[4 * g] + lo[4 * g], hi[4 * g + 1] + lo[4 * g + 1], hi[4 * g + 2] + lo[4 * g + 2], hi[4 * g + 3] + lo[4 * g + 3]);
            }
            __syncthreads();
            const float4* theirs = reinterpret_cast<const float4*>(xch) + ((wave ^ 4) * 2) * 64 + lane;
            const int r = lane & 31;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int g = gk + j;
                const float4 o = theirs[j * 64];
                const int nb = 32 * p + acc_n(g, lane);
                const float4 bv = *reinterpret_cast<const float4*>(bias + nb);
                float4 y;
                y.x = elu(fmaf((hi[4 * g + 0] + lo[4 * g + 0]) + o.x, k, bv.x));
                y.y = elu(fmaf((hi[4 * g + 1] + lo[4 * g + 1]) + o.y, k, bv.y));
                y.z = elu(fmaf((hi[4 * g + 2] + lo[4 * g + 2]) + o.z, k, bv.z));
                y.w = elu(fmaf((hi[4 * g + 3] + lo[4 * g + 3]) + o.w, k, bv.w));
                h2_store4<FS_P1>(outp, r, nb, make_float4(y.x * s, y.y * s, y.z * s, y.w * s), am);
            }
            __syncthreads();
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (tid == 0) out[blockIdx.x] = t1 - t0;
    if (am == 12345.678f) out[0] = 0;
}
}  // namespace

template <int WAVES, int PARTS = 0>
static double run(const u16* PH, const float* bias, int phases)
{
    const int wgs = 256;
    unsigned long long* out;
    (void)hipMalloc(&out, wgs * 8);
    const size_t ldsb = WS_LDS_HALVES * 2 + WS_XCH_FLOATS * 4 + 128 * 4;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(phase_kernel<WAVES, PARTS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    for (int rep = 0; rep < 3; ++rep)
        hipLaunchKernelGGL((phase_kernel<WAVES, PARTS>), dim3(wgs), dim3(64 * WAVES), ldsb, 0, PH, bias, out, phases, 1.0f / 64.0f, 4.0f);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(wgs);
    (void)hipMemcpy(h.data(), out, wgs * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    (void)hipFree(out);
    return (double)h[wgs / 2] / phases;
}

int main()
{
    std::vector<uint16_t> hp(H2_PH_HALVES);
    srand(7);
    for (auto& v : hp) v = (uint16_t)(((rand() & 1) << 15) | ((8 + rand() % 6) << 10) | (rand() & 1023));     // fp16 of 2^-7 .. 2^-2, random mantissas
    std::vector<float> hb(128, 0.01f);
    u16* PH; float* bias;
    (void)hipMalloc(&PH, hp.size() * 2); (void)hipMalloc(&bias, 512);
    (void)hipMemcpy(PH, hp.data(), hp.size() * 2, hipMemcpyHostToDevice);
    (void)hipMemcpy(bias, hb.data(), 512, hipMemcpyHostToDevice);
    const int phases = 2000;
    for (int pass = 0; pass < 2; ++pass) {
        const double a = run<4>(PH, bias, phases), b = run<8>(PH, bias, phases);
        printf("layer-2-like phase (K = 256 -> 128 columns, ELU, split, barrier): 4 waves %.0f cycles | 8 waves (K halves + exchange) %.0f cycles | ratio %.3f\n",
               a, b, b / a);
    }
    printf("4 waves, parts: GEMM only %.0f | GEMM + epilogue %.0f | epilogue + barrier %.0f cycles\n", run<4, 1>(PH, bias, phases), run<4, 2>(PH, bias, phases),
           run<4, 3>(PH, bias, phases));
    return 0;
}
