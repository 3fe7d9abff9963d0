// mlp_mfma.hip — the actor-critic MLP (reference ppo.py:10-102) on the gfx950 matrix cores.
//
// fp32 in / fp32 accumulate MFMA (v_mfma_f32_32x32x2_f32): bit-for-bit an fp32 fma chain, so the
// network keeps the reference's fp32 numerics while running on the matrix pipe (157 TFLOP/s
// dense peak) instead of the Tensile fp32 GEMMs that managed 8-14 TFLOP/s on these skinny shapes
// (profiles/r1a_*).
//
// Forward: ONE launch runs all four layers for a tile of 32 rows per 256-thread workgroup.
//   * activations never leave the CU between layers: each layer's output tile is written to LDS
//     in [row][k] order with a pitch of K+4 floats, which makes the next layer's A-fragment read a
//     conflict-free ds_read_b128 (4 consecutive k per lane);
//   * weights are NOT staged through LDS: every wave owns a slice of the output columns and
//     streams its B-fragments straight from L2 with 16-byte loads (row-major [N][K] = torch's
//     Linear layout, packed by mlp_layout.h), each load feeding four MFMAs;
//   * k is split in two halves across the two 32-lane halves of the wave (lane>>5), the k-order
//     inside a dot product is a fixed bijection, so results are run-to-run deterministic;
//   * bias + ELU fused in the epilogue; layer 4 (32 outputs) is split-K over the four waves and
//     reduced through LDS so that no wave idles.
// Backward (dX chain) reuses the same tile routine on the transposed weights; dW is a separate
// split-over-rows kernel (see below).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstdio>
#include <cstdlib>
#include "flyhip.h"
#include "mlp_layout.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
// 16-byte store of a register group to a saved tensor (nontemporal stores measured 4-6 % slower)
#define GSTORE4(ptr, v) (*reinterpret_cast<float4*>(ptr) = (v))

constexpr int BM = 32;          // rows per workgroup
constexpr int NWAVE = 4;
constexpr int THREADS = 64 * NWAVE;

// ELU as torch evaluates it: x > 0 ? x : exp(x) - 1 (ELU.cpp), with the hardware exp2 path
// (v_exp_f32, ~1 ulp on exp): absolute error vs expm1 <= ~1.2e-7, inside the stated 2e-5 tolerance.
__device__ __forceinline__ float elu(float x) { return x > 0.0f ? x : __expf(x) - 1.0f; }
// derivative of ELU expressed through its OUTPUT y: 1 for y > 0, y + 1 otherwise
__device__ __forceinline__ float elu_grad_from_out(float y) { return y > 0.0f ? 1.0f : y + 1.0f; }
// dA * ELU'(H) with H the saved OUTPUT: ELU' = min(H, 0) + 1, so the product is one min and one fma
// (dA * min(H, 0) + dA, rounded once) instead of compare, select, add, multiply
__device__ __forceinline__ float dact(float da, float h) { return fmaf(da, fminf(h, 0.0f), da); }

// acc[t] (+)= W[col tile t][K] * A[32 rows x K]^T for this wave's NT column tiles.  The WEIGHTS are
// the MFMA "A" operand and the activations the "B" operand, so the 32x32 result tile is
// C[n][row]: a lane owns ONE row (lane&31) and, per group of four accumulator registers, FOUR
// CONSECUTIVE output columns n = 8*(reg>>2) + 4*(lane>>5) + (reg&3) -- which makes the epilogue's
// LDS traffic 16-byte vectors instead of scalars.
//   lds_in : [32][K+4] floats (row-major, k contiguous)
//   Wf     : this layer's weights in fragment order (mlp_layout.h): [col tile][K/8][64 lanes][4];
//            `tile0` = first column tile of this wave.  One wave-instruction = one contiguous KiB.
// k mapping: MFMA step s = 4*kq+q multiplies k = 4*kq+q (lanes 0..31) and k = K/2+4*kq+q (32..63).
// Measured on MI355X (tools/mfma_valu_overlap.hip): VALU, LDS and VMEM instructions do NOT execute under
// a running MFMA of the same SIMD -- every v_mov / address add / s_waitcnt stall in a GEMM loop is
// matrix-pipe time lost.  So the k loop is straight-line code: fully unrolled over a ring of four
// NAMED operand sets (compile-time indices -> registers, no copies), each set reloaded right
// after the four MFMAs that consumed it, i.e. three sets (768 matrix cycles) ahead of its next
// use; sched_barriers pin that order so the scheduler can neither hoist the loads (register
// blow-up) nor sink them (exposed L2 latency).  The first MFMA of an accumulator takes the
// constant 0 as its C operand, so no accumulator clearing is issued either.
//
// The first two weight k-quads of a GEMM are requested AHEAD of it: a caller issues `gemm_prefetch`
// before the previous layer's epilogue / barrier so the L2 round trip (1-2 k cycles, once per layer
// per tile) hides under that work instead of opening every MFMA run.
#ifndef TUNE_RING
#define TUNE_RING 4
#endif
#ifndef TUNE_ARING
#define TUNE_ARING 2
#endif
constexpr int RING = TUNE_RING, HEAD = 2, ARING = TUNE_ARING;      // weight sets (L2 latency), head sets, activation sets (LDS latency)
template <int NT>
struct WeightHead { float4 b[HEAD][NT]; };

template <int K, int NT>
__device__ __forceinline__ void gemm_prefetch(WeightHead<NT>& w, const float* __restrict__ Wf, int tile0, int lane)
{
    static_assert(K / 8 >= RING, "at least RING k-quads");
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const float* bp = Wf + (long)(tile0 + t) * (K / 8) * 256 + lane * 4;
#pragma unroll
        for (int s = 0; s < HEAD; ++s) w.b[s][t] = *reinterpret_cast<const float4*>(bp + 256 * s);
    }
}

// ZERO: the accumulators start from 0 (their incoming value is ignored); otherwise they are added to.
template <int K, int NT, bool ZERO = true>
__device__ __forceinline__ void tile_gemm(const WeightHead<NT>& head, const float* __restrict__ Wf, int tile0,
                                          const float* lds_in, f32x16 (&acc)[NT], int lane)
{
    constexpr int K8 = K / 8;
    const int r = lane & 31, h = lane >> 5;
    const float* ap = lds_in + r * (K + 4) + h * (K / 2);
    const float* bp[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) bp[t] = Wf + (long)(tile0 + t) * K8 * 256 + lane * 4;
    float4 b[RING][NT], a[ARING];
#pragma unroll
    for (int s = 0; s < RING; ++s) {
        if (s < ARING) a[s] = *reinterpret_cast<const float4*>(ap + 4 * s);
#pragma unroll
        for (int t = 0; t < NT; ++t)
            b[s][t] = s < HEAD ? head.b[s][t] : *reinterpret_cast<const float4*>(bp[t] + 256 * s);
    }
#pragma unroll
    for (int kq = 0; kq < K8; ++kq) {
        const int s = kq % RING, sa = kq % ARING;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (ZERO && kq == 0) {
                const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[s][t].x, a[sa].x, z, 0, 0, 0);
            } else {
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[s][t].x, a[sa].x, acc[t], 0, 0, 0);
            }
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[s][t].y, a[sa].y, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[s][t].z, a[sa].z, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[s][t].w, a[sa].w, acc[t], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (kq + RING < K8) {
#pragma unroll
            for (int t = 0; t < NT; ++t) b[s][t] = *reinterpret_cast<const float4*>(bp[t] + 256 * (kq + RING));
        }
        if (kq + ARING < K8) a[sa] = *reinterpret_cast<const float4*>(ap + 4 * (kq + ARING));
    }
    __builtin_amdgcn_sched_barrier(0);
}

template <int K, int NT, bool ZERO = true>
__device__ __forceinline__ void tile_gemm(const float* __restrict__ Wf, int tile0, const float* lds_in,
                                          f32x16 (&acc)[NT], int lane)
{
    WeightHead<NT> head;
    gemm_prefetch<K, NT>(head, Wf, tile0, lane);
    tile_gemm<K, NT, ZERO>(head, Wf, tile0, lds_in, acc, lane);
}

// C/D layout of the 32x32 tile with the operand roles above: lane holds row (lane&31) and output
// columns n = 8*g + 4*(lane>>5) + j for register 4*g + j.
__device__ __forceinline__ int acc_n(int g, int lane) { return 8 * g + 4 * (lane >> 5); }
// Saved activations and dZ tensors ([rows][N], written by these epilogues, read back by the
// backward epilogues and the dW kernel) live in HBM in TILE-FRAGMENT order: per 32-row tile and
// 32-column tile one block of 1024 floats laid out [g][lane][4] exactly like the accumulator
// fragments, so a wave's store or load of one register group is one contiguous KiB (row-major rows
// would be 32-byte pieces of 32 different lines: measured ~30 us of the fused launch).  Offset of
// (row, col) inside its 32 x 32 block; blocks follow each other [row tile][column tile].
__device__ __forceinline__ int frag_off(int row, int col) { return ((col >> 3) * 64 + ((col >> 2) & 1) * 32 + row) * 4 + (col & 3); }
// generic C/D map (A operand indexes rows): row (reg&3) + 8*(reg>>2) + 4*(lane>>5), column lane&31
__device__ __forceinline__ int acc_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

// bias + ELU epilogue: writes the activation tile to LDS (the next layer's operand), 16 bytes per
// store, and -- when `gdst` is given -- the same 16 bytes to the saved-activation rows in HBM.  A
// wave's four stores of one column tile touch the same 32 lines (one 128-byte line per row) and
// together fill them, so L2 merges them into full-line writes; no LDS read-back pass is needed.
// ---------------------------------------------------------------------------------------------
// bf16x3 GEMM path: the same tile GEMMs on v_mfma_f32_32x32x16_bf16 with both operands split into
// three bf16 terms (mlp_layout.h).  Six MFMAs of 8 passes per 16-wide k block replace eight fp32
// MFMAs of 16 passes: 192 instead of 512 matrix cycles, and (tools/bf16x3_gemm.hip) a smaller
// error than the fp32 MFMA chain because each instruction sums its 16 products before rounding.
// The large term w0*x0 accumulates in `hi`, the five small ones in `lo`; the epilogue adds them.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;
struct Frag3 { float4 p[3]; };                                   // one operand fragment: 3 terms x 8 bf16
__device__ __forceinline__ bf16x8 as_bf16x8(const float4& v) { return __builtin_bit_cast(bf16x8, v); }

constexpr int B3_PAD = 8;                                        // LDS row pitch (K + 8) bf16 = 4 banks past a multiple of 64
constexpr int B3_RING = 4, B3_HEAD = 2, B3_ARING = 2;
template <int K> constexpr int b3_plane() { return BM * (K + B3_PAD); }   // 16-bit words per term plane of a [32][K] tile

struct WeightHead3 { Frag3 b[B3_HEAD]; };

template <int K>
__device__ __forceinline__ void gemm_prefetch_b3(WeightHead3& w, const u16* __restrict__ Wb, int tile0, int lane)
{
    static_assert(K / 16 >= B3_HEAD, "at least B3_HEAD k blocks");
    const u16* bp = Wb + (long)tile0 * (K / 16) * 1536 + lane * 8;
#pragma unroll
    for (int s = 0; s < B3_HEAD; ++s)
#pragma unroll
        for (int p = 0; p < 3; ++p) w.b[s].p[p] = *reinterpret_cast<const float4*>(bp + s * 1536 + p * 512);
}

// (hi, lo) (+)= W[col tile][K] * A[32 rows x K]^T; lds_in = term 0 plane of the activation tile
// ([32][K + 8] bf16 per plane, planes b3_plane<K>() apart); Wb = this operand's planes.
template <int K, bool ZERO = true>
__device__ __forceinline__ void tile_gemm_b3(const WeightHead3& head, const u16* __restrict__ Wb, int tile0,
                                             const u16* lds_in, f32x16& hi, f32x16& lo, int lane)
{
    constexpr int K16 = K / 16;
    constexpr int RING = K16 < B3_RING ? K16 : B3_RING;
    constexpr int ARING = K16 < B3_ARING ? K16 : B3_ARING;
    const int r = lane & 31, h = lane >> 5;
    const u16* ap = lds_in + r * (K + B3_PAD) + 8 * h;
    const u16* bp = Wb + (long)tile0 * K16 * 1536 + lane * 8;
    Frag3 b[RING], a[ARING];
#pragma unroll
    for (int s = 0; s < RING; ++s) {
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            if (s < ARING) a[s].p[p] = *reinterpret_cast<const float4*>(ap + p * b3_plane<K>() + 16 * s);
            b[s].p[p] = s < B3_HEAD ? head.b[s].p[p] : *reinterpret_cast<const float4*>(bp + s * 1536 + p * 512);
        }
    }
    const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < K16; ++kb) {
        const int s = kb % RING, sa = kb % ARING;
        __builtin_amdgcn_sched_barrier(0);
        const bf16x8 w0 = as_bf16x8(b[s].p[0]), w1 = as_bf16x8(b[s].p[1]), w2 = as_bf16x8(b[s].p[2]);
        const bf16x8 x0 = as_bf16x8(a[sa].p[0]), x1 = as_bf16x8(a[sa].p[1]), x2 = as_bf16x8(a[sa].p[2]);
        lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, x2, (ZERO && kb == 0) ? z : lo, 0, 0, 0);
        hi = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, x0, (ZERO && kb == 0) ? z : hi, 0, 0, 0);
        lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2, x0, lo, 0, 0, 0);
        lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, x1, lo, 0, 0, 0);
        lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, x1, lo, 0, 0, 0);
        lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, x0, lo, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (kb + RING < K16) {
#pragma unroll
            for (int p = 0; p < 3; ++p) b[s].p[p] = *reinterpret_cast<const float4*>(bp + (kb + RING) * 1536 + p * 512);
        }
        if (kb + ARING < K16) {
#pragma unroll
            for (int p = 0; p < 3; ++p) a[sa].p[p] = *reinterpret_cast<const float4*>(ap + p * b3_plane<K>() + 16 * (kb + ARING));
        }
    }
    __builtin_amdgcn_sched_barrier(0);
}

// x = a + b + c exactly, each a bf16 (round to nearest even): the terms of the split
__device__ __forceinline__ void split3(float x, u16& a, u16& b, u16& c)
{
    const __bf16 t0 = (__bf16)x;
    const float r1 = x - (float)t0;
    const __bf16 t1 = (__bf16)r1;
    const float r2 = r1 - (float)t1;
    const __bf16 t2 = (__bf16)r2;
    a = __builtin_bit_cast(u16, t0); b = __builtin_bit_cast(u16, t1); c = __builtin_bit_cast(u16, t2);
}

// the same split for two values at once, terms returned as packed bf16 pairs (x in the low half):
// one v_cvt_pk_bf16_f32 per term, a shift and a mask to widen a pair back to fp32
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split3x2(float x, float y, unsigned& a, unsigned& b, unsigned& c)
{
    f32x2 v = {x, y};
    a = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
    v[0] -= __builtin_bit_cast(float, a << 16);
    v[1] -= __builtin_bit_cast(float, a & 0xffff0000u);
    b = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
    v[0] -= __builtin_bit_cast(float, b << 16);
    v[1] -= __builtin_bit_cast(float, b & 0xffff0000u);
    c = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}

// four consecutive columns of one row -> the three term planes of an LDS tile (8-byte stores)
template <int K>
__device__ __forceinline__ void store_split4(u16* lds_plane0, int row, int col, const float4& y)
{
    unsigned a0, b0, c0, a1, b1, c1;
    split3x2(y.x, y.y, a0, b0, c0);
    split3x2(y.z, y.w, a1, b1, c1);
    u16* q = lds_plane0 + row * (K + B3_PAD) + col;
    *reinterpret_cast<uint2*>(q) = make_uint2(a0, a1);
    *reinterpret_cast<uint2*>(q + b3_plane<K>()) = make_uint2(b0, b1);
    *reinterpret_cast<uint2*>(q + 2 * b3_plane<K>()) = make_uint2(c0, c1);
}

// bias + ELU epilogue of the bf16x3 path: hi + lo, activation to HBM as fp32 (the backward and dW
// read it) and to the LDS tile of the next GEMM as three bf16 terms.  N = width of the LDS tile.
template <int N, int NG = N>
__device__ __forceinline__ void epilogue_elu_b3(const f32x16& hi, const f32x16& lo, const float* __restrict__ bias, int col0,
                                                u16* lds_plane0, int lane, float* __restrict__ gtile, int nvalid)
{
    const int r = lane & 31;
    float4 y[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int nb = col0 + acc_n(g, lane);
        const float4 bv = *reinterpret_cast<const float4*>(bias + nb);
        y[g].x = elu((hi[4 * g + 0] + lo[4 * g + 0]) + bv.x);
        y[g].y = elu((hi[4 * g + 1] + lo[4 * g + 1]) + bv.y);
        y[g].z = elu((hi[4 * g + 2] + lo[4 * g + 2]) + bv.z);
        y[g].w = elu((hi[4 * g + 3] + lo[4 * g + 3]) + bv.w);
        store_split4<N>(lds_plane0, r, nb, y[g]);
    }
    if (gtile != nullptr && r < nvalid) {
        float* gl = gtile + (col0 / 32) * 1024 + lane * 4;
#pragma unroll
        for (int g = 0; g < 4; ++g) GSTORE4(gl + g * 256, y[g]);
    }
}

// `gtile` points at this tile's first row of the destination (a wave-uniform pointer); lanes
// address it with small 32-bit offsets, and all stores sit under ONE predicate (`nvalid` rows of
// the tile exist): 64-bit per-lane address arithmetic and per-store exec masking are issue slots
// the matrix pipe does not get back.
template <int N, int NT, int NG = N>
__device__ __forceinline__ void epilogue_elu(const f32x16 (&acc)[NT], const float* __restrict__ bias, int col0,
                                             float* lds_out, int lane, float* __restrict__ gtile, int nvalid)
{
    const int r = lane & 31;
    float4 y[NT][4];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int nb = col0 + 32 * t + acc_n(g, lane);
            const float4 bv = *reinterpret_cast<const float4*>(bias + nb);
            y[t][g].x = elu(acc[t][4 * g + 0] + bv.x);
            y[t][g].y = elu(acc[t][4 * g + 1] + bv.y);
            y[t][g].z = elu(acc[t][4 * g + 2] + bv.z);
            y[t][g].w = elu(acc[t][4 * g + 3] + bv.w);
            *reinterpret_cast<float4*>(lds_out + r * (N + 4) + nb) = y[t][g];
        }
    }
    if (gtile != nullptr && r < nvalid) {
        float* gl = gtile + (col0 / 32) * 1024 + lane * 4;        // column tile col0/32 of this row tile, fragment order
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) GSTORE4(gl + t * 1024 + g * 256, y[t][g]);
    }
}

// Both tile kernels fit four workgroups per CU: <= 40 KB of LDS and <= 128 VGPRs each.
constexpr int WGS_PER_CU = 4;
constexpr int PERSIST_GRID = 256 * WGS_PER_CU;       // persistent launches: every slot of the chip
constexpr int LDS_A_FLOATS = BM * (MLP_H2 + 4);      // one 128-column half of H1 at a time, later H3
constexpr int LDS_B_FLOATS = BM * (MLP_H2 + 4);      // X0, later H2, later the split-K partials
constexpr int LB1 = 0, LB2 = MLP_H1, LB3 = LB2 + MLP_H2, LB4 = LB3 + MLP_H3, LSD = LB4 + MLP_OUT, LLG = LSD + 32;
constexpr int LDS_C_FLOATS = LLG + 32;               // biases + sampling constants (2.4 KB)

// In-kernel phase stamps (diagnostic instantiation only, tools/stamp_forward.py; the shipped
// instantiation compiles them out): wave 0 / lane 0 of each workgroup stores s_memtime at the phase
// boundaries into a buffer nothing else reads.
template <bool STAMP>
__device__ __forceinline__ void stamp(unsigned long long* buf, int slot)
{
    if (STAMP) {
        if (threadIdx.x == 0) {
            unsigned long long t;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
            buf[slot] = t;
        }
    }
}

// x [n][73] -> out [n][32] (cols 0..17 = mean after ELU, col 18 = value, rest 0).
// mu_out [n][18] / v_out [n] / h*_save are optional.
// Persistent workgroups: the grid is at most 3 workgroups per CU and each walks tiles
// blockIdx.x, blockIdx.x + gridDim.x, ...; the NEXT tile's input rows are fetched into registers
// (16-byte loads: a 32x73 tile is one contiguous, 16-byte aligned block) while the current tile
// computes, so the ~5 us HBM round trip that used to open every workgroup is hidden.
constexpr int XV = (BM * MLP_IN / 4 + THREADS - 1) / THREADS;       // float4 per thread for one x tile (584 / 256 -> 3)

constexpr int FWD_LDS_FLOATS = LDS_A_FLOATS + LDS_B_FLOATS + LDS_C_FLOATS;

// The body walks tiles first_tile, first_tile + tile_stride, ... (< ntiles) on the caller's LDS
// arena.
template <bool STAMP>
__device__ __forceinline__ void forward_body(
    float* lds, const long first_tile, const long tile_stride,
    const float* __restrict__ P, const float* __restrict__ PF, const float* __restrict__ x, long n,
    float* __restrict__ mu_out, float* __restrict__ v_out, float* __restrict__ out_save,
    float* __restrict__ h1_save, float* __restrict__ h2_save, float* __restrict__ h3_save,
    const float* __restrict__ smp_eps, const float* __restrict__ smp_var, float* __restrict__ smp_act,
    float* __restrict__ smp_logp, unsigned long long* __restrict__ stamps_base,
    const int smp_var_steps = 0, const float smp_var_decay = 0.0f, const float smp_var_min = 0.0f)
{
    unsigned long long* stamps = stamps_base;
    float* ldsA = lds;
    float* ldsB = lds + LDS_A_FLOATS;
    float* ldsBias = ldsB + LDS_B_FLOATS;   // b1 | b2 | b3 | b4 | sqrt(var) | log sqrt(var): read by every epilogue
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave index in an SGPR: weight bases become scalar
    const long ntiles = (n + BM - 1) / BM;
    const long total = n * MLP_IN;

    float4 xr[XV];
    auto x_load = [&](long tile) {
        const long base = tile * (BM * MLP_IN);
#pragma unroll
        for (int u = 0; u < XV; ++u) {
            const long f = base + 4L * (tid + u * THREADS);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (tid + u * THREADS < BM * MLP_IN / 4) {
                if (f + 3 < total) v = *reinterpret_cast<const float4*>(x + f);
                else {
                    if (f < total) v.x = x[f];
                    if (f + 1 < total) v.y = x[f + 1];
                    if (f + 2 < total) v.z = x[f + 2];
                }
            }
            xr[u] = v;
        }
    };
    auto x_store = [&](int tid) {   // registers -> ldsB as [32][80+4]; pad columns 73..79 zeroed
#pragma unroll
        for (int u = 0; u < XV; ++u) {
            const int i4 = tid + u * THREADS;
            if (i4 < BM * MLP_IN / 4) {
                const float e[4] = {xr[u].x, xr[u].y, xr[u].z, xr[u].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int f = 4 * i4 + j;
                    const int rr = f / MLP_IN, cc = f - rr * MLP_IN;
                    ldsB[rr * (MLP_IN_PAD + 4) + cc] = e[j];
                }
            }
        }
        if (tid < BM * (MLP_IN_PAD - MLP_IN)) {
            const int rr = tid / (MLP_IN_PAD - MLP_IN), cc = MLP_IN + tid - rr * (MLP_IN_PAD - MLP_IN);
            ldsB[rr * (MLP_IN_PAD + 4) + cc] = 0.0f;
        }
    };

    long tile = first_tile;
    if (tile < ntiles) x_load(tile);
    {   // biases (and the sampling constants) -> LDS once per workgroup; the first tile's barrier publishes them
        ldsBias[LB1 + tid] = P[MLP_OFF_B1 + tid];
        if (tid < MLP_H2) ldsBias[LB2 + tid] = P[MLP_OFF_B2 + tid];
        else ldsBias[LB3 + tid - MLP_H2] = P[MLP_OFF_B3 + tid - MLP_H2];
        if (tid < MLP_OUT) ldsBias[LB4 + tid] = P[MLP_OFF_B4 + tid];
        if (smp_var && tid >= 64 && tid < 64 + MLP_NACT) {
            float v = smp_var[tid - 64];
            for (int i = 0; i < smp_var_steps; ++i) v = fmaxf(smp_var_min, v - smp_var_decay);   // ppo.py:236-237, not yet applied to the tensor
            const float L = sqrtf(v);
            ldsBias[LSD + tid - 64] = L;
            ldsBias[LLG + tid - 64] = logf(L);
        }
    }
    for (; tile < ntiles; tile += tile_stride) {
        const long row0 = tile * BM;
        const int nvalid = (int)(n - row0 < BM ? n - row0 : BM);            // rows of this tile that exist
        float* h1_tile = h1_save ? h1_save + row0 * MLP_H1 : nullptr;       // wave-uniform tile bases
        float* h2_tile = h2_save ? h2_save + row0 * MLP_H2 : nullptr;
        float* h3_tile = h3_save ? h3_save + row0 * MLP_H3 : nullptr;
        // opaque per-iteration copy of the thread index: keeps the dozens of tile-invariant LDS/global
        // offsets from being hoisted out of the tile loop (they would all be live across it and spill)
        int tl = tid;
        asm volatile("" : "+v"(tl));
        // the sampling noise of this tile's (row, column) slots of the last phase, requested now: one
        // wave per SIMD (8192 rollout rows = one tile per CU) has nobody to hide a late HBM load behind
        float eps_pre[BM * MLP_OUT / THREADS];
        if (smp_eps) {
#pragma unroll
            for (int k = 0; k < BM * MLP_OUT / THREADS; ++k) {
                const int i = tl + k * THREADS, row = i >> 5, col = i & 31;
                eps_pre[k] = (col < MLP_NACT && row < nvalid) ? (smp_eps + row0 * MLP_NACT)[row * MLP_NACT + col] : 0.0f;
            }
        }
        if (STAMP) stamps = stamps_base + tile * 16;         // one 16-slot record per tile
        stamp<STAMP>(stamps, 0);
        if (STAMP && threadIdx.x == 0) {
            unsigned long long t;
            asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
            // top 16 bits: which CU this workgroup landed on (HW_ID cu/sh/se bits 8..15, XCC_ID)
            const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20);
            t = (t & 0xffffffffffffull) | ((unsigned long long)(((hw >> 8) & 0xff) | ((xcc & 0xf) << 8)) << 48);
            stamps[14] = t;
        }
        // Layers 1 and 2 run in two halves of 128 hidden-1 columns so that only a [32][132] slice of
        // H1 is ever in LDS (36 KB per workgroup -> four workgroups per CU): L1 produces columns
        // [0,128), L2 accumulates their k range, L1 produces [128,256), L2 accumulates the rest.
        WeightHead<1> w1a, w1b, w2a, w2b, w3;
        gemm_prefetch<MLP_IN_PAD, 1>(w1a, PF + MLP_OFF_F1, wave, lane);       // lands during the x staging
        x_store(tl);
        __syncthreads();
        if (tile + tile_stride < ntiles) x_load(tile + tile_stride);     // lands during this tile's MFMAs
        stamp<STAMP>(stamps, 1);
        f32x16 acc2[1];
        {   // L1, columns [0,128): wave owns 32 of them
            f32x16 acc[1];
            tile_gemm<MLP_IN_PAD, 1>(w1a, PF + MLP_OFF_F1, wave, ldsB, acc, lane);
            stamp<STAMP>(stamps, 2);
            gemm_prefetch<MLP_H2, 1>(w2a, PF + MLP_OFF_F2, wave, lane);           // both land during the epilogue
            gemm_prefetch<MLP_IN_PAD, 1>(w1b, PF + MLP_OFF_F1, 4 + wave, lane);
            epilogue_elu<MLP_H2, 1, MLP_H1>(acc, ldsBias + LB1, wave * 32, ldsA, lane, h1_tile, nvalid);
        }
        stamp<STAMP>(stamps, 3);
        __syncthreads();
        stamp<STAMP>(stamps, 4);
        {   // L2 over k in [0,128), then L1 columns [128,256) -- one uninterrupted run of MFMAs
            f32x16 acc[1];
            tile_gemm<MLP_H2, 1>(w2a, PF + MLP_OFF_F2, wave, ldsA, acc2, lane);
            tile_gemm<MLP_IN_PAD, 1>(w1b, PF + MLP_OFF_F1, 4 + wave, ldsB, acc, lane);
            stamp<STAMP>(stamps, 5);
            gemm_prefetch<MLP_H2, 1>(w2b, PF + MLP_OFF_F2 + MLP_H2 * (MLP_H1 / 2), wave, lane);
            __syncthreads();                                   // every wave has finished reading the first half of H1
            epilogue_elu<MLP_H2, 1, MLP_H1>(acc, ldsBias + LB1 + MLP_H1 / 2, wave * 32, ldsA, lane, h1_tile ? h1_tile + 4 * 1024 : nullptr, nvalid);
        }
        stamp<STAMP>(stamps, 6);
        __syncthreads();
        stamp<STAMP>(stamps, 7);
        {   // L2 over k in [128,256): 256 -> 128 complete, wave owns 32 columns
            tile_gemm<MLP_H2, 1, false>(w2b, PF + MLP_OFF_F2 + MLP_H2 * (MLP_H1 / 2), wave, ldsA, acc2, lane);
            stamp<STAMP>(stamps, 8);
            gemm_prefetch<MLP_H2, 1>(w3, PF + MLP_OFF_F3, wave, lane);
            epilogue_elu<MLP_H2, 1>(acc2, ldsBias + LB2, wave * 32, ldsB, lane, h2_tile, nvalid);   // x is dead: every wave passed the barrier above
        }
        __syncthreads();
        float4 w4[4];                                                            // layer-4 weights of this wave's k range
        {   // L3: 128 -> 128 (actor | critic heads stacked)
            f32x16 acc[1];
            stamp<STAMP>(stamps, 9);
            tile_gemm<MLP_H2, 1>(w3, PF + MLP_OFF_F3, wave, ldsB, acc, lane);
            stamp<STAMP>(stamps, 10);
            const float* bp = PF + MLP_OFF_F4 + (wave * 4) * 256 + lane * 4;    // [wave][kq][lane][4]
#pragma unroll
            for (int kq = 0; kq < 4; ++kq) w4[kq] = *reinterpret_cast<const float4*>(bp + 256 * kq);
            epilogue_elu<MLP_H3, 1>(acc, ldsBias + LB3, wave * 32, ldsA, lane, h3_tile, nvalid);
        }
        stamp<STAMP>(stamps, 11);
        __syncthreads();
        stamp<STAMP>(stamps, 12);
        {   // L4: 128 -> 32, split-K over the four waves (32 k each), partials reduced through LDS
            f32x16 acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
            const int r = lane & 31, h = lane >> 5;
            const float* ap = ldsA + r * (MLP_H3 + 4) + wave * 32 + h * 16;
#pragma unroll
            for (int kq = 0; kq < 4; ++kq) {
                const float4 a = *reinterpret_cast<const float4*>(ap + 4 * kq);
                const float4 b = w4[kq];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b.x, a.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b.y, a.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b.z, a.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b.w, a.w, acc, 0, 0, 0);
            }
            float* part = ldsB + wave * (BM * MLP_OUT);                          // [row][32]
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4*>(part + r * MLP_OUT + acc_n(g, lane)) =
                    make_float4(acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]);
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < BM * MLP_OUT / THREADS; ++k) {
            const int i = tl + k * THREADS;
            const int row = i >> 5, col = i & 31;
            float z = ((ldsB[i] + ldsB[BM * MLP_OUT + i]) + ldsB[2 * BM * MLP_OUT + i]) + ldsB[3 * BM * MLP_OUT + i];
            z += ldsBias[LB4 + col];
            float y = (col < MLP_NACT) ? elu(z) : ((col == MLP_NACT) ? z : 0.0f);   // ELU on the mean (ppo.py:30), none on v
            const bool in = row < nvalid;
            if (in) {
                if (out_save) (out_save + row0 * MLP_OUT)[frag_off(row, col)] = y;
                if (mu_out && col < MLP_NACT) (mu_out + row0 * MLP_NACT)[row * MLP_NACT + col] = y;
                if (v_out && col == MLP_NACT) (v_out + row0)[row] = y;
            }
            if (smp_eps) {
                // ppo.py:215-220 fused: the 32 lanes that hold one output row sample its action
                // (a = mu + sqrt(var) eps), reduce the Mahalanobis term and sum log L with a fixed
                // xor-butterfly over the half-wave, and write the clipped action and the log-prob.
                float x2 = 0.0f, lg = 0.0f, a = 0.0f;
                const bool on = (col < MLP_NACT) && in;
                if (on) {
                    const float L = ldsBias[LSD + col];
                    a = y + L * eps_pre[k];
                    const float xj = (a - y) / L;
                    x2 = xj * xj;
                    lg = ldsBias[LLG + col];
                }
#pragma unroll
                for (int o = 1; o < 32; o <<= 1) { x2 += __shfl_xor(x2, o, 32); lg += __shfl_xor(lg, o, 32); }
                if (on) (smp_act + row0 * MLP_NACT)[row * MLP_NACT + col] = fminf(fmaxf(a, -1.0f), 1.0f);
                if (col == 0 && in) (smp_logp + row0)[row] = -0.5f * (33.08178959434617f + x2) - lg;
            }
        }
        stamp<STAMP>(stamps, 13);
        if (STAMP && threadIdx.x == 0) {
            unsigned long long t;
            asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
            stamps[15] = t;
        }
        __syncthreads();            // ldsB (partials) is the next tile's input buffer
    }
}

// bf16x3 variant of the forward body: same tile flow, activations live in LDS as three bf16 term planes
constexpr int B3_TILE_FLOATS = 3 * b3_plane<MLP_H2>() / 2;        // a [32][128] tile as three planes, in floats
constexpr int FWD_B3_LDS_FLOATS = 2 * B3_TILE_FLOATS + LB4;      // two tiles + b1 | b2 | b3: 54 272 B, three workgroups per CU

// The body walks tiles first_tile, first_tile + tile_stride, ... (< ntiles) on the caller's LDS
// arena.
template <bool STAMP>
__device__ __forceinline__ void forward_body_b3(
    float* lds, const long first_tile, const long tile_stride,
    const float* __restrict__ P, const u16* __restrict__ PB, const float* __restrict__ x, long n,
    float* __restrict__ mu_out, float* __restrict__ v_out, float* __restrict__ out_save,
    float* __restrict__ h1_save, float* __restrict__ h2_save, float* __restrict__ h3_save,
    const float* __restrict__ smp_eps, const float* __restrict__ smp_var, float* __restrict__ smp_act,
    float* __restrict__ smp_logp, unsigned long long* __restrict__ stamps_base,
    const int smp_var_steps = 0, const float smp_var_decay = 0.0f, const float smp_var_min = 0.0f)
{
    unsigned long long* stamps = stamps_base;
    u16* ldsA = reinterpret_cast<u16*>(lds);                                   // H1 half / H3: three [32][136] planes
    u16* ldsB = reinterpret_cast<u16*>(lds + B3_TILE_FLOATS);                  // X ([32][88] planes) / H2 / fp32 split-K partials
    float* ldsBf = lds + B3_TILE_FLOATS;
    float* ldsBias = lds + 2 * B3_TILE_FLOATS;   // b1 | b2 | b3 | b4 | sqrt(var) | log sqrt(var): read by every epilogue
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave index in an SGPR: weight bases become scalar
    const long ntiles = (n + BM - 1) / BM;
    const long total = n * MLP_IN;

    float4 xr[XV];
    auto x_load = [&](long tile) {
        const long base = tile * (BM * MLP_IN);
#pragma unroll
        for (int u = 0; u < XV; ++u) {
            const long f = base + 4L * (tid + u * THREADS);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (tid + u * THREADS < BM * MLP_IN / 4) {
                if (f + 3 < total) v = *reinterpret_cast<const float4*>(x + f);
                else {
                    if (f < total) v.x = x[f];
                    if (f + 1 < total) v.y = x[f + 1];
                    if (f + 2 < total) v.z = x[f + 2];
                }
            }
            xr[u] = v;
        }
    };
    auto x_store = [&](int tid) {   // registers -> ldsB as three [32][80+8] bf16 term planes; pad columns 73..79 zeroed
#pragma unroll
        for (int u = 0; u < XV; ++u) {
            const int i4 = tid + u * THREADS;
            if (i4 < BM * MLP_IN / 4) {
                const float e[4] = {xr[u].x, xr[u].y, xr[u].z, xr[u].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int f = 4 * i4 + j;
                    const int rr = f / MLP_IN, cc = f - rr * MLP_IN;
                    u16 sa, sb, sc;
                    split3(e[j], sa, sb, sc);
                    u16* q = ldsB + rr * (MLP_IN_PAD + B3_PAD) + cc;
                    q[0] = sa; q[b3_plane<MLP_IN_PAD>()] = sb; q[2 * b3_plane<MLP_IN_PAD>()] = sc;
                }
            }
        }
        if (tid < BM * (MLP_IN_PAD - MLP_IN)) {
            const int rr = tid / (MLP_IN_PAD - MLP_IN), cc = MLP_IN + tid - rr * (MLP_IN_PAD - MLP_IN);
            u16* q = ldsB + rr * (MLP_IN_PAD + B3_PAD) + cc;
            q[0] = 0; q[b3_plane<MLP_IN_PAD>()] = 0; q[2 * b3_plane<MLP_IN_PAD>()] = 0;
        }
    };

    long tile = first_tile;
    if (tile < ntiles) x_load(tile);
    {   // biases (and the sampling constants) -> LDS once per workgroup; the first tile's barrier publishes them
        ldsBias[LB1 + tid] = P[MLP_OFF_B1 + tid];
        if (tid < MLP_H2) ldsBias[LB2 + tid] = P[MLP_OFF_B2 + tid];
        else ldsBias[LB3 + tid - MLP_H2] = P[MLP_OFF_B3 + tid - MLP_H2];
    }
    // this thread's output column is the same in every tile: its layer-4 bias and sampling constants
    // stay in registers (the LDS budget of this path is exactly three workgroups per CU)
    const float b4v = P[MLP_OFF_B4 + (tid & 31)];
    float smpL = 1.0f, smpLog = 0.0f;
    if (smp_var && (tid & 31) < MLP_NACT) {
        float v = smp_var[tid & 31];
        for (int i = 0; i < smp_var_steps; ++i) v = fmaxf(smp_var_min, v - smp_var_decay);   // ppo.py:236-237, not yet applied to the tensor
        smpL = sqrtf(v);
        smpLog = logf(smpL);
    }
    for (; tile < ntiles; tile += tile_stride) {
        const long row0 = tile * BM;
        const int nvalid = (int)(n - row0 < BM ? n - row0 : BM);            // rows of this tile that exist
        float* h1_tile = h1_save ? h1_save + row0 * MLP_H1 : nullptr;       // wave-uniform tile bases
        float* h2_tile = h2_save ? h2_save + row0 * MLP_H2 : nullptr;
        float* h3_tile = h3_save ? h3_save + row0 * MLP_H3 : nullptr;
        // opaque per-iteration copy of the thread index: keeps the dozens of tile-invariant LDS/global
        // offsets from being hoisted out of the tile loop (they would all be live across it and spill)
        int tl = tid;
        asm volatile("" : "+v"(tl));
        // the sampling noise of this tile's (row, column) slots of the last phase, requested now: one
        // wave per SIMD (8192 rollout rows = one tile per CU) has nobody to hide a late HBM load behind
        float eps_pre[BM * MLP_OUT / THREADS];
        if (smp_eps) {
#pragma unroll
            for (int k = 0; k < BM * MLP_OUT / THREADS; ++k) {
                const int i = tl + k * THREADS, row = i >> 5, col = i & 31;
                eps_pre[k] = (col < MLP_NACT && row < nvalid) ? (smp_eps + row0 * MLP_NACT)[row * MLP_NACT + col] : 0.0f;
            }
        }
        if (STAMP) stamps = stamps_base + tile * 16;         // one 16-slot record per tile
        stamp<STAMP>(stamps, 0);
        if (STAMP && threadIdx.x == 0) {
            unsigned long long t;
            asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
            // top 16 bits: which CU this workgroup landed on (HW_ID cu/sh/se bits 8..15, XCC_ID)
            const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20);
            t = (t & 0xffffffffffffull) | ((unsigned long long)(((hw >> 8) & 0xff) | ((xcc & 0xf) << 8)) << 48);
            stamps[14] = t;
        }
        // Layers 1 and 2 run in two halves of 128 hidden-1 columns so that only a [32][128] slice of
        // H1 is ever in LDS: L1 produces columns
        // [0,128), L2 accumulates their k range, L1 produces [128,256), L2 accumulates the rest.
        WeightHead3 w1a, w1b, w2a, w2b, w3;
        gemm_prefetch_b3<MLP_IN_PAD>(w1a, PB + MLP_OFF_PB1, wave, lane);       // lands during the x staging
        x_store(tl);
        __syncthreads();
        if (tile + tile_stride < ntiles) x_load(tile + tile_stride);     // lands during this tile's MFMAs
        stamp<STAMP>(stamps, 1);
        f32x16 hi2, lo2;
        {   // L1, columns [0,128): wave owns 32 of them
            f32x16 hi, lo;
            tile_gemm_b3<MLP_IN_PAD>(w1a, PB + MLP_OFF_PB1, wave, ldsB, hi, lo, lane);
            stamp<STAMP>(stamps, 2);
            gemm_prefetch_b3<MLP_H2>(w2a, PB + MLP_OFF_PB2, wave, lane);              // both land during the epilogue
            gemm_prefetch_b3<MLP_IN_PAD>(w1b, PB + MLP_OFF_PB1, 4 + wave, lane);
            epilogue_elu_b3<MLP_H2, MLP_H1>(hi, lo, ldsBias + LB1, wave * 32, ldsA, lane, h1_tile, nvalid);
        }
        stamp<STAMP>(stamps, 3);
        __syncthreads();
        stamp<STAMP>(stamps, 4);
        {   // L2 over k in [0,128), then L1 columns [128,256) -- one uninterrupted run of MFMAs
            f32x16 hi, lo;
            tile_gemm_b3<MLP_H2>(w2a, PB + MLP_OFF_PB2, wave, ldsA, hi2, lo2, lane);
            tile_gemm_b3<MLP_IN_PAD>(w1b, PB + MLP_OFF_PB1, 4 + wave, ldsB, hi, lo, lane);
            stamp<STAMP>(stamps, 5);
            gemm_prefetch_b3<MLP_H2>(w2b, PB + MLP_OFF_PB2 + 3 * MLP_H2 * (MLP_H1 / 2), wave, lane);
            __syncthreads();                                   // every wave has finished reading the first half of H1
            epilogue_elu_b3<MLP_H2, MLP_H1>(hi, lo, ldsBias + LB1 + MLP_H1 / 2, wave * 32, ldsA, lane, h1_tile ? h1_tile + 4 * 1024 : nullptr, nvalid);
        }
        stamp<STAMP>(stamps, 6);
        __syncthreads();
        stamp<STAMP>(stamps, 7);
        {   // L2 over k in [128,256): 256 -> 128 complete, wave owns 32 columns
            tile_gemm_b3<MLP_H2, false>(w2b, PB + MLP_OFF_PB2 + 3 * MLP_H2 * (MLP_H1 / 2), wave, ldsA, hi2, lo2, lane);
            stamp<STAMP>(stamps, 8);
            gemm_prefetch_b3<MLP_H2>(w3, PB + MLP_OFF_PB3, wave, lane);
            epilogue_elu_b3<MLP_H2>(hi2, lo2, ldsBias + LB2, wave * 32, ldsB, lane, h2_tile, nvalid);   // x is dead: every wave passed the barrier above
        }
        __syncthreads();
        Frag3 w4[2];                                                             // layer-4 weights of this wave's two k blocks
        {   // L3: 128 -> 128 (actor | critic heads stacked)
            f32x16 hi, lo;
            stamp<STAMP>(stamps, 9);
            tile_gemm_b3<MLP_H2>(w3, PB + MLP_OFF_PB3, wave, ldsB, hi, lo, lane);
            stamp<STAMP>(stamps, 10);
            const u16* bp = PB + MLP_OFF_PB4 + (wave * 2) * 1536 + lane * 8;    // [k block][term][lane][8]
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) w4[kk].p[pl] = *reinterpret_cast<const float4*>(bp + kk * 1536 + pl * 512);
            epilogue_elu_b3<MLP_H3>(hi, lo, ldsBias + LB3, wave * 32, ldsA, lane, h3_tile, nvalid);
        }
        stamp<STAMP>(stamps, 11);
        __syncthreads();
        stamp<STAMP>(stamps, 12);
        {   // L4: 128 -> 32, split-K over the four waves (32 k each), partials reduced through LDS
            f32x16 hi, lo;
#pragma unroll
            for (int i = 0; i < 16; ++i) { hi[i] = 0.0f; lo[i] = 0.0f; }
            const int r = lane & 31, h = lane >> 5;
            const u16* ap = ldsA + r * (MLP_H3 + B3_PAD) + wave * 32 + 8 * h;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                bf16x8 xq[3], wq[3];
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    xq[pl] = as_bf16x8(*reinterpret_cast<const float4*>(ap + pl * b3_plane<MLP_H3>() + 16 * kk));
                    wq[pl] = as_bf16x8(w4[kk].p[pl]);
                }
                lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq[0], xq[2], lo, 0, 0, 0);
                hi = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq[0], xq[0], hi, 0, 0, 0);
                lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq[2], xq[0], lo, 0, 0, 0);
                lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq[1], xq[1], lo, 0, 0, 0);
                lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq[0], xq[1], lo, 0, 0, 0);
                lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq[1], xq[0], lo, 0, 0, 0);
            }
            // every wave must be done reading H2 (ldsB, layer 3) before the partials overwrite it:
            // they are -- the barrier after the layer-3 epilogue is behind all of them
            float* part = ldsBf + wave * (BM * MLP_OUT);                         // [row][32]
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4*>(part + r * MLP_OUT + acc_n(g, lane)) =
                    make_float4(hi[4 * g] + lo[4 * g], hi[4 * g + 1] + lo[4 * g + 1], hi[4 * g + 2] + lo[4 * g + 2],
                                hi[4 * g + 3] + lo[4 * g + 3]);
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < BM * MLP_OUT / THREADS; ++k) {
            const int i = tl + k * THREADS;
            const int row = i >> 5, col = i & 31;
            float z = ((ldsBf[i] + ldsBf[BM * MLP_OUT + i]) + ldsBf[2 * BM * MLP_OUT + i]) + ldsBf[3 * BM * MLP_OUT + i];
            z += b4v;
            float y = (col < MLP_NACT) ? elu(z) : ((col == MLP_NACT) ? z : 0.0f);   // ELU on the mean (ppo.py:30), none on v
            const bool in = row < nvalid;
            if (in) {
                if (out_save) (out_save + row0 * MLP_OUT)[frag_off(row, col)] = y;
                if (mu_out && col < MLP_NACT) (mu_out + row0 * MLP_NACT)[row * MLP_NACT + col] = y;
                if (v_out && col == MLP_NACT) (v_out + row0)[row] = y;
            }
            if (smp_eps) {
                // ppo.py:215-220 fused: the 32 lanes that hold one output row sample its action
                // (a = mu + sqrt(var) eps), reduce the Mahalanobis term and sum log L with a fixed
                // xor-butterfly over the half-wave, and write the clipped action and the log-prob.
                float x2 = 0.0f, lg = 0.0f, a = 0.0f;
                const bool on = (col < MLP_NACT) && in;
                if (on) {
                    const float L = smpL;
                    a = y + L * eps_pre[k];
                    const float xj = (a - y) / L;
                    x2 = xj * xj;
                    lg = smpLog;
                }
#pragma unroll
                for (int o = 1; o < 32; o <<= 1) { x2 += __shfl_xor(x2, o, 32); lg += __shfl_xor(lg, o, 32); }
                if (on) (smp_act + row0 * MLP_NACT)[row * MLP_NACT + col] = fminf(fmaxf(a, -1.0f), 1.0f);
                if (col == 0 && in) (smp_logp + row0)[row] = -0.5f * (33.08178959434617f + x2) - lg;
            }
        }
        stamp<STAMP>(stamps, 13);
        if (STAMP && threadIdx.x == 0) {
            unsigned long long t;
            asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
            stamps[15] = t;
        }
        __syncthreads();            // ldsB (partials) is the next tile's input buffer
    }
}

template <bool STAMP>
__global__ __launch_bounds__(THREADS, 2) void mlp_forward_kernel(
    const float* __restrict__ P, const float* __restrict__ PF, const float* __restrict__ x, long n,
    float* __restrict__ mu_out, float* __restrict__ v_out, float* __restrict__ out_save,
    float* __restrict__ h1_save, float* __restrict__ h2_save, float* __restrict__ h3_save,
    const float* __restrict__ smp_eps, const float* __restrict__ smp_var, float* __restrict__ smp_act,
    float* __restrict__ smp_logp, unsigned long long* __restrict__ stamps_base, int smp_var_steps, float smp_var_decay,
    float smp_var_min)
{
    __shared__ __attribute__((aligned(16))) float lds[FWD_LDS_FLOATS];
    forward_body<STAMP>(lds, blockIdx.x, gridDim.x, P, PF, x, n, mu_out, v_out, out_save, h1_save,
                        h2_save, h3_save, smp_eps, smp_var, smp_act, smp_logp, stamps_base, smp_var_steps, smp_var_decay,
                        smp_var_min);
}

__global__ __launch_bounds__(THREADS, 1) void mlp_forward_b3_kernel(
    const float* __restrict__ P, const u16* __restrict__ PB, const float* __restrict__ x, long n,
    float* __restrict__ mu_out, float* __restrict__ v_out, float* __restrict__ out_save,
    float* __restrict__ h1_save, float* __restrict__ h2_save, float* __restrict__ h3_save,
    const float* __restrict__ smp_eps, const float* __restrict__ smp_var, float* __restrict__ smp_act,
    float* __restrict__ smp_logp, int smp_var_steps, float smp_var_decay, float smp_var_min)
{
    __shared__ __attribute__((aligned(16))) float lds[FWD_B3_LDS_FLOATS];
    // one tile per workgroup (a constant stride no tile index reaches): this body's straight-line GEMMs
    // leave no registers for the next-tile prefetch of the persistent fp32 variant
    forward_body_b3<false>(lds, blockIdx.x, 1L << 40, P, PB, x, n, mu_out, v_out, out_save, h1_save, h2_save, h3_save,
                           smp_eps, smp_var, smp_act, smp_logp, nullptr, smp_var_steps, smp_var_decay, smp_var_min);
}

// ---------------------------------------------------------------------------------------------
// Backward, part 1: PPO loss gradient at the network outputs + the dX chain (ppo.py:184-197).
// One 32-row tile per workgroup.  The loss (ppo.py:191-194) is
//     mean_i( -min(ratio_i A_i, clamp(ratio_i, 1-c, 1+c) A_i) ) + mean_i huber(v_i - target_i)
// with ratio_i = exp(logp_i - old_logp_i) and logp the diagonal-Gaussian log-density of the
// stored action under the CURRENT variance.  Gradients follow torch's subgradient choices:
// min() splits a tie evenly, clamp() passes gradient on the closed interval.
//   dz4 [n][32]: cols 0..17 d/d(pre-ELU mean), col 18 d/d(value), rest 0
//   dz3 [n][128], dz2 [n][128], dz1 [n][256]: gradients at the pre-activations of layers 3,2,1
//   loss_part [grid][2]: per-workgroup sums of the policy term and of the Huber term
// dZ = dA * ELU'(H).  H is read from the saved activations in exactly the accumulator layout (a
// lane owns row lane&31 and four consecutive columns per register group): one 16-byte load per
// group, requested before the GEMM whose epilogue consumes it.  The four loads of a column tile
// touch one 128-byte line per row, so L1 serves three of them.
template <int NT>
struct HFrag { float4 v[NT][4]; };

template <int N, int NT>
__device__ __forceinline__ void hfrag_load(HFrag<NT>& hf, const float* __restrict__ htile, int nvalid, int col0, int lane)
{
    // the saved activations are in tile-fragment order (see below): this wave's four loads of a
    // column tile are four contiguous KiB.  Rows past the end of the batch read whatever the
    // (allocated) rest of the last tile holds: their results are never stored and a row of an
    // MFMA only feeds the same row.
    const float* gl = htile + (col0 / 32) * 1024 + lane * 4;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) hf.v[t][g] = *reinterpret_cast<const float4*>(gl + t * 1024 + g * 256);
}

// writes dZ to the gradient rows in HBM and (lds_out != nullptr) to the LDS tile the next GEMM reads
template <int N, int NT>
__device__ __forceinline__ void epilogue_dact(const f32x16 (&acc)[NT], const HFrag<NT>& hf, int col0, float* lds_out,
                                              float* __restrict__ gtile, int nvalid, int lane)
{
    const int r = lane & 31;
    float4 z[NT][4];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int nb = col0 + 32 * t + acc_n(g, lane);
            const float4 hv = hf.v[t][g];
            z[t][g].x = dact(acc[t][4 * g + 0], hv.x);
            z[t][g].y = dact(acc[t][4 * g + 1], hv.y);
            z[t][g].z = dact(acc[t][4 * g + 2], hv.z);
            z[t][g].w = dact(acc[t][4 * g + 3], hv.w);
            if (lds_out) *reinterpret_cast<float4*>(lds_out + r * (N + 4) + nb) = z[t][g];
        }
    }
    if (r < nvalid) {
        float* gl = gtile + (col0 / 32) * 1024 + lane * 4;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) GSTORE4(gl + t * 1024 + g * 256, z[t][g]);
    }
}

constexpr int BW_Z2 = 0;                                   // [32][132]
constexpr int BW_Z3 = BW_Z2 + BM * (MLP_H2 + 4);           // [32][132]
constexpr int BW_Z4 = BW_Z3 + BM * (MLP_H3 + 4);           // [32][36]
constexpr int BW_TAIL = BW_Z4 + BM * (MLP_OUT + 4);        // [32][2] per-row loss terms
constexpr int BW_FLOATS = BW_TAIL + 2 * BM;                // 38.9 KB

__device__ __forceinline__ void backward_body(
    float* lds, const long tile,
    const float* __restrict__ PT, const float* __restrict__ out_saved, const float* __restrict__ h1_saved,
    const float* __restrict__ h2_saved, const float* __restrict__ h3_saved,
    const float* __restrict__ action, const float* __restrict__ old_logp, const float* __restrict__ adv,
    const float* __restrict__ target, const float* __restrict__ var, long n, float inv_batch, float clip,
    float* __restrict__ dz4, float* __restrict__ dz3, float* __restrict__ dz2, float* __restrict__ dz1,
    float* __restrict__ loss_part)
{
    float* ldsZ2 = lds + BW_Z2;
    float* ldsZ3 = lds + BW_Z3;
    float* ldsZ4 = lds + BW_Z4;
    float* rowloss = lds + BW_TAIL;                        // [32][2]: policy term, Huber term of each row
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave index in an SGPR: weight bases become scalar
    const long row0 = tile * BM;
    const int nvalid = (int)(n - row0 < BM ? n - row0 : BM);                // rows of this tile that exist

    WeightHead<1> wt4, wt3;
    gemm_prefetch<MLP_OUT, 1>(wt4, PT + MLP_OFF_TF4, wave, lane);
    HFrag<1> hf3;
    hfrag_load<MLP_H3, 1>(hf3, h3_saved + row0 * MLP_H3, nvalid, wave * 32, lane);     // in flight during the loss phase

    // Loss gradient at the outputs, one thread per (row, output column): the 32 lanes of a row
    // reduce the Mahalanobis term and log-determinant with a fixed xor butterfly, every lane then
    // holds the row's d loss / d logp and writes its own column of dZ4 (HBM + the LDS operand).
    {
        const float* out_t = out_saved + row0 * MLP_OUT;       // wave-uniform tile bases, 32-bit lane offsets
        const float* act_t = action + row0 * MLP_NACT;
        const float* olp_t = old_logp + row0;
        const float* adv_t = adv + row0;
        const float* tgt_t = target + row0;
        float* dz4_t = dz4 + row0 * MLP_OUT;
        const int col = tid & 31;
        const bool act = col < MLP_NACT;
        const float L = act ? sqrtf(var[col]) : 1.0f;
        const float inv_L = 1.0f / L, inv_var = act ? 1.0f / var[col] : 0.0f;       // one division each per thread, not per element
        float half_log_det = act ? logf(L) : 0.0f;
#pragma unroll
        for (int o = 1; o < 32; o <<= 1) half_log_det += __shfl_xor(half_log_det, o, 32);
#pragma unroll
        for (int k = 0; k < BM * MLP_OUT / THREADS; ++k) {
            const int row = (tid >> 5) + k * (THREADS / 32);
            const bool in = row < nvalid;
            const float y = in ? out_t[frag_off(row, col)] : 0.0f;                   // mean (cols 0..17), value (col 18)
            const float a = (in && act) ? act_t[row * MLP_NACT + col] : 0.0f;
            const float xj = act ? (a - y) * inv_L : 0.0f;
            float M = xj * xj;
#pragma unroll
            for (int o = 1; o < 32; o <<= 1) M += __shfl_xor(M, o, 32);
            float d = 0.0f, pol = 0.0f, hub = 0.0f;
            if (in) {
                const float logp = -0.5f * (33.08178959434617f + M) - half_log_det;
                const float ratio = expf(logp - olp_t[row]);
                const float A = adv_t[row];
                const float s1 = ratio * A;
                const float rc = fminf(fmaxf(ratio, 1.0f - clip), 1.0f + clip);
                const float s2 = rc * A;
                const float in_range = (ratio >= 1.0f - clip && ratio <= 1.0f + clip) ? 1.0f : 0.0f;
                float dmin;                                    // d min(s1,s2) / d ratio
                if (s1 < s2) dmin = A;
                else if (s1 > s2) dmin = A * in_range;
                else dmin = 0.5f * (A + A * in_range);
                const float c = -inv_batch * ratio * dmin;     // d loss / d logp
                pol = -fminf(s1, s2);
                const float dv = __shfl(y, MLP_NACT, 32) - tgt_t[row];
                hub = fabsf(dv) < 1.0f ? 0.5f * dv * dv : fabsf(dv) - 0.5f;
                if (act) d = c * (a - y) * inv_var * elu_grad_from_out(y);
                else if (col == MLP_NACT) d = inv_batch * fminf(fmaxf(dv, -1.0f), 1.0f);   // smooth_l1', beta = 1
                dz4_t[frag_off(row, col)] = d;
            }
            ldsZ4[row * (MLP_OUT + 4) + col] = d;
            if (col == 0) { rowloss[2 * row] = pol; rowloss[2 * row + 1] = hub; }
        }
    }
    __syncthreads();
    if (tid < 32 && loss_part) {       // fixed-order sum of the 32 rows' loss terms
        float pol = rowloss[2 * tid], hub = rowloss[2 * tid + 1];
        for (int o = 16; o > 0; o >>= 1) { pol += __shfl_down(pol, o, 32); hub += __shfl_down(hub, o, 32); }
        if (tid == 0) { loss_part[2 * tile] = pol; loss_part[2 * tile + 1] = hub; }
    }
    HFrag<1> hf2;
    {   // dA3 = dZ4 . W4  ->  dZ3
        f32x16 acc[1];
        tile_gemm<MLP_OUT, 1>(wt4, PT + MLP_OFF_TF4, wave, ldsZ4, acc, lane);
        gemm_prefetch<MLP_H3, 1>(wt3, PT + MLP_OFF_TF3, wave, lane);          // both land during the epilogue + barrier
        hfrag_load<MLP_H2, 1>(hf2, h2_saved + row0 * MLP_H2, nvalid, wave * 32, lane);
        epilogue_dact<MLP_H3, 1>(acc, hf3, wave * 32, ldsZ3, dz3 + row0 * MLP_H3, nvalid, lane);
    }
    __syncthreads();
    WeightHead<2> wt2;
    HFrag<2> hf1;
    {   // dA2 = dZ3 . W3  ->  dZ2
        f32x16 acc[1];
        tile_gemm<MLP_H3, 1>(wt3, PT + MLP_OFF_TF3, wave, ldsZ3, acc, lane);
        gemm_prefetch<MLP_H2, 2>(wt2, PT + MLP_OFF_TF2, wave * 2, lane);
        epilogue_dact<MLP_H2, 1>(acc, hf2, wave * 32, ldsZ2, dz2 + row0 * MLP_H2, nvalid, lane);
    }
    __syncthreads();
    {   // dA1 = dZ2 . W2  ->  dZ1 (no later GEMM reads it: HBM only)
        hfrag_load<MLP_H1, 2>(hf1, h1_saved + row0 * MLP_H1, nvalid, wave * 64, lane);       // lands during the MFMAs
        f32x16 acc[2];
        tile_gemm<MLP_H2, 2>(wt2, PT + MLP_OFF_TF2, wave * 2, ldsZ2, acc, lane);
        epilogue_dact<MLP_H1, 2>(acc, hf1, wave * 64, nullptr, dz1 + row0 * MLP_H1, nvalid, lane);
    }
}

// dZ = (hi + lo) * ELU'(H) of the bf16x3 path: fp32 rows to HBM, three bf16 terms to the next GEMM's LDS tile
template <int N>
__device__ __forceinline__ void epilogue_dact_b3(const f32x16& hi, const f32x16& lo, const HFrag<1>& hf, int col0,
                                                 u16* lds_plane0, float* __restrict__ gtile, int nvalid, int lane)
{
    const int r = lane & 31;
    float4 z[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float4 hv = hf.v[0][g];
        z[g].x = dact(hi[4 * g + 0] + lo[4 * g + 0], hv.x);
        z[g].y = dact(hi[4 * g + 1] + lo[4 * g + 1], hv.y);
        z[g].z = dact(hi[4 * g + 2] + lo[4 * g + 2], hv.z);
        z[g].w = dact(hi[4 * g + 3] + lo[4 * g + 3], hv.w);
        if (lds_plane0) store_split4<N>(lds_plane0, r, col0 + acc_n(g, lane), z[g]);
    }
    if (r < nvalid) {
        float* gl = gtile + (col0 / 32) * 1024 + lane * 4;
#pragma unroll
        for (int g = 0; g < 4; ++g) GSTORE4(gl + g * 256, z[g]);
    }
}

constexpr int BW_B3_LDS_FLOATS = 2 * B3_TILE_FLOATS + 2 * BM;

__device__ __forceinline__ void backward_body_b3(
    float* lds, const long tile,
    const u16* __restrict__ PTB, const float* __restrict__ out_saved, const float* __restrict__ h1_saved,
    const float* __restrict__ h2_saved, const float* __restrict__ h3_saved,
    const float* __restrict__ action, const float* __restrict__ old_logp, const float* __restrict__ adv,
    const float* __restrict__ target, const float* __restrict__ var, long n, float inv_batch, float clip,
    float* __restrict__ dz4, float* __restrict__ dz3, float* __restrict__ dz2, float* __restrict__ dz1,
    float* __restrict__ loss_part)
{
    u16* ldsZ2 = reinterpret_cast<u16*>(lds);                     // dZ2: three [32][136] term planes
    u16* ldsZ3 = reinterpret_cast<u16*>(lds + B3_TILE_FLOATS);    // dZ3: three [32][136] term planes
    u16* ldsZ4 = ldsZ2;                                           // dZ4 ([32][40] planes) is dead before dZ2 is written
    float* rowloss = lds + 2 * B3_TILE_FLOATS;                        // [32][2]: policy term, Huber term of each row
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave index in an SGPR: weight bases become scalar
    const long row0 = tile * BM;
    const int nvalid = (int)(n - row0 < BM ? n - row0 : BM);                // rows of this tile that exist

    WeightHead3 wt4, wt3;
    gemm_prefetch_b3<MLP_OUT>(wt4, PTB + MLP_OFF_PTB4, wave, lane);
    HFrag<1> hf3;
    hfrag_load<MLP_H3, 1>(hf3, h3_saved + row0 * MLP_H3, nvalid, wave * 32, lane);     // in flight during the loss phase

    // Loss gradient at the outputs, one thread per (row, output column): the 32 lanes of a row
    // reduce the Mahalanobis term and log-determinant with a fixed xor butterfly, every lane then
    // holds the row's d loss / d logp and writes its own column of dZ4 (HBM + the LDS operand).
    {
        const float* out_t = out_saved + row0 * MLP_OUT;       // wave-uniform tile bases, 32-bit lane offsets
        const float* act_t = action + row0 * MLP_NACT;
        const float* olp_t = old_logp + row0;
        const float* adv_t = adv + row0;
        const float* tgt_t = target + row0;
        float* dz4_t = dz4 + row0 * MLP_OUT;
        const int col = tid & 31;
        const bool act = col < MLP_NACT;
        const float L = act ? sqrtf(var[col]) : 1.0f;
        const float inv_L = 1.0f / L, inv_var = act ? 1.0f / var[col] : 0.0f;       // one division each per thread, not per element
        float half_log_det = act ? logf(L) : 0.0f;
#pragma unroll
        for (int o = 1; o < 32; o <<= 1) half_log_det += __shfl_xor(half_log_det, o, 32);
#pragma unroll
        for (int k = 0; k < BM * MLP_OUT / THREADS; ++k) {
            const int row = (tid >> 5) + k * (THREADS / 32);
            const bool in = row < nvalid;
            const float y = in ? out_t[frag_off(row, col)] : 0.0f;                   // mean (cols 0..17), value (col 18)
            const float a = (in && act) ? act_t[row * MLP_NACT + col] : 0.0f;
            const float xj = act ? (a - y) * inv_L : 0.0f;
            float M = xj * xj;
#pragma unroll
            for (int o = 1; o < 32; o <<= 1) M += __shfl_xor(M, o, 32);
            float d = 0.0f, pol = 0.0f, hub = 0.0f;
            if (in) {
                const float logp = -0.5f * (33.08178959434617f + M) - half_log_det;
                const float ratio = expf(logp - olp_t[row]);
                const float A = adv_t[row];
                const float s1 = ratio * A;
                const float rc = fminf(fmaxf(ratio, 1.0f - clip), 1.0f + clip);
                const float s2 = rc * A;
                const float in_range = (ratio >= 1.0f - clip && ratio <= 1.0f + clip) ? 1.0f : 0.0f;
                float dmin;                                    // d min(s1,s2) / d ratio
                if (s1 < s2) dmin = A;
                else if (s1 > s2) dmin = A * in_range;
                else dmin = 0.5f * (A + A * in_range);
                const float c = -inv_batch * ratio * dmin;     // d loss / d logp
                pol = -fminf(s1, s2);
                const float dv = __shfl(y, MLP_NACT, 32) - tgt_t[row];
                hub = fabsf(dv) < 1.0f ? 0.5f * dv * dv : fabsf(dv) - 0.5f;
                if (act) d = c * (a - y) * inv_var * elu_grad_from_out(y);
                else if (col == MLP_NACT) d = inv_batch * fminf(fmaxf(dv, -1.0f), 1.0f);   // smooth_l1', beta = 1
                dz4_t[frag_off(row, col)] = d;
            }
            {
                u16 sa, sb, sc;
                split3(d, sa, sb, sc);
                u16* q = ldsZ4 + row * (MLP_OUT + B3_PAD) + col;
                q[0] = sa; q[b3_plane<MLP_OUT>()] = sb; q[2 * b3_plane<MLP_OUT>()] = sc;
            }
            if (col == 0) { rowloss[2 * row] = pol; rowloss[2 * row + 1] = hub; }
        }
    }
    __syncthreads();
    if (tid < 32 && loss_part) {       // fixed-order sum of the 32 rows' loss terms
        float pol = rowloss[2 * tid], hub = rowloss[2 * tid + 1];
        for (int o = 16; o > 0; o >>= 1) { pol += __shfl_down(pol, o, 32); hub += __shfl_down(hub, o, 32); }
        if (tid == 0) { loss_part[2 * tile] = pol; loss_part[2 * tile + 1] = hub; }
    }
    HFrag<1> hf2;
    {   // dA3 = dZ4 . W4  ->  dZ3
        f32x16 hi, lo;
        tile_gemm_b3<MLP_OUT>(wt4, PTB + MLP_OFF_PTB4, wave, ldsZ4, hi, lo, lane);
        gemm_prefetch_b3<MLP_H3>(wt3, PTB + MLP_OFF_PTB3, wave, lane);        // both land during the epilogue + barrier
        hfrag_load<MLP_H2, 1>(hf2, h2_saved + row0 * MLP_H2, nvalid, wave * 32, lane);
        epilogue_dact_b3<MLP_H3>(hi, lo, hf3, wave * 32, ldsZ3, dz3 + row0 * MLP_H3, nvalid, lane);
    }
    __syncthreads();
    WeightHead3 wt2;
    HFrag<1> hf1;
    {   // dA2 = dZ3 . W3  ->  dZ2
        f32x16 hi, lo;
        tile_gemm_b3<MLP_H3>(wt3, PTB + MLP_OFF_PTB3, wave, ldsZ3, hi, lo, lane);
        gemm_prefetch_b3<MLP_H2>(wt2, PTB + MLP_OFF_PTB2, wave * 2, lane);
        epilogue_dact_b3<MLP_H2>(hi, lo, hf2, wave * 32, ldsZ2, dz2 + row0 * MLP_H2, nvalid, lane);
    }
    __syncthreads();
    {   // dA1 = dZ2 . W2  ->  dZ1 (no later GEMM reads it: HBM only): this wave's two column tiles in turn
        WeightHead3 wt2b;
        HFrag<1> hf1b;
        hfrag_load<MLP_H1, 1>(hf1, h1_saved + row0 * MLP_H1, nvalid, wave * 64, lane);              // lands during the MFMAs
        f32x16 hi, lo;
        tile_gemm_b3<MLP_H2>(wt2, PTB + MLP_OFF_PTB2, wave * 2, ldsZ2, hi, lo, lane);
        gemm_prefetch_b3<MLP_H2>(wt2b, PTB + MLP_OFF_PTB2, wave * 2 + 1, lane);
        hfrag_load<MLP_H1, 1>(hf1b, h1_saved + row0 * MLP_H1, nvalid, wave * 64 + 32, lane);
        epilogue_dact_b3<MLP_H1>(hi, lo, hf1, wave * 64, nullptr, dz1 + row0 * MLP_H1, nvalid, lane);
        tile_gemm_b3<MLP_H2>(wt2b, PTB + MLP_OFF_PTB2, wave * 2 + 1, ldsZ2, hi, lo, lane);
        epilogue_dact_b3<MLP_H1>(hi, lo, hf1b, wave * 64 + 32, nullptr, dz1 + row0 * MLP_H1, nvalid, lane);
    }
}

__global__ __launch_bounds__(THREADS, WGS_PER_CU) void mlp_backward_dx_kernel(
    const float* __restrict__ PT, const float* __restrict__ out_saved, const float* __restrict__ h1_saved,
    const float* __restrict__ h2_saved, const float* __restrict__ h3_saved,
    const float* __restrict__ action, const float* __restrict__ old_logp, const float* __restrict__ adv,
    const float* __restrict__ target, const float* __restrict__ var, long n, float inv_batch, float clip,
    float* __restrict__ dz4, float* __restrict__ dz3, float* __restrict__ dz2, float* __restrict__ dz1,
    float* __restrict__ loss_part)
{
    __shared__ __attribute__((aligned(16))) float lds[BW_FLOATS];
    backward_body(lds, blockIdx.x, PT, out_saved, h1_saved, h2_saved, h3_saved, action,
                  old_logp, adv, target, var, n, inv_batch, clip, dz4, dz3, dz2, dz1, loss_part);
}

__global__ __launch_bounds__(THREADS, 2) void mlp_backward_dx_b3_kernel(
    const u16* __restrict__ PTB, const float* __restrict__ out_saved, const float* __restrict__ h1_saved,
    const float* __restrict__ h2_saved, const float* __restrict__ h3_saved,
    const float* __restrict__ action, const float* __restrict__ old_logp, const float* __restrict__ adv,
    const float* __restrict__ target, const float* __restrict__ var, long n, float inv_batch, float clip,
    float* __restrict__ dz4, float* __restrict__ dz3, float* __restrict__ dz2, float* __restrict__ dz1,
    float* __restrict__ loss_part)
{
    __shared__ __attribute__((aligned(16))) float lds[BW_B3_LDS_FLOATS];
    backward_body_b3(lds, blockIdx.x, PTB, out_saved, h1_saved, h2_saved, h3_saved, action, old_logp, adv, target, var, n,
                     inv_batch, clip, dz4, dz3, dz2, dz1, loss_part);
}

// Forward and backward of one minibatch in ONE launch of 2 * tiles workgroups.  Both are row-local:
// workgroup b < tiles runs the forward of tile b and publishes flag[b]; workgroup tiles + b waits
// for that flag and runs the backward of tile b.  A launch of `tiles` workgroups at 3 per CU ends
// in a ragged tail (5 tiles per CU at 40 960 rows: the last third of its time runs one or two
// workgroups per CU); here the backward workgroups of the tiles that finished first fill the
// slots the forward frees, so only ONE tail is left per minibatch instead of two.
//  * No deadlock: workgroups are dispatched in blockIdx order, so every forward workgroup is
//    resident or finished before its consumer starts to poll; a poll budget (~1 s) turns a lost
//    flag into *err = 1 instead of a hang.
//  * Visibility without cache maintenance: device-scope release/acquire fences would write back and
//    invalidate the XCD's whole L2 per workgroup (measured: 1.6x slower, the weights keep getting
//    evicted).  Instead producer and consumer are required to sit on the SAME XCD, i.e. behind
//    the same L2: workgroups go round-robin over the 8 XCDs, and `tiles` is padded to a multiple
//    of 8 by the launcher, so b and tiles + b land together.  The producer waits until its stores
//    are acknowledged by L2 (vmcnt(0)) before it publishes the flag together with its XCC id; the
//    consumer compares that id with its own and raises *err = 2 on a mismatch (the host then
//    falls back to two launches).  The consumer's L1 cannot hold these lines: L1 is invalidated
//    at kernel start and nothing on the CU has read this tile's rows since.
constexpr int FB_LDS_FLOATS = FWD_LDS_FLOATS > BW_FLOATS ? FWD_LDS_FLOATS : BW_FLOATS;

// s_memrealtime (100 MHz) with the CU identity in the top 16 bits (diagnostic stamps only)
__device__ __forceinline__ unsigned long long realtime_cu()
{
    unsigned long long t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20);
    return (t & 0xffffffffffffull) | ((unsigned long long)(((hw >> 8) & 0xff) | ((xcc & 0xf) << 8)) << 48);
}

constexpr int FB_B3_LDS_FLOATS = FWD_B3_LDS_FLOATS > BW_B3_LDS_FLOATS ? FWD_B3_LDS_FLOATS : BW_B3_LDS_FLOATS;

// B3: the bf16x3 GEMM bodies (PF/PT then point at the 16-bit term planes PB/PTB)
template <bool STAMP, bool B3>
__global__ __launch_bounds__(THREADS, B3 ? 3 : WGS_PER_CU) void mlp_fwd_bwd_kernel(
    const float* __restrict__ P, const void* __restrict__ PF, const void* __restrict__ PT,
    const float* __restrict__ x, long n, float* __restrict__ out_save, float* __restrict__ h1_save,
    float* __restrict__ h2_save, float* __restrict__ h3_save,
    const float* __restrict__ action, const float* __restrict__ old_logp, const float* __restrict__ adv,
    const float* __restrict__ target, const float* __restrict__ var, float inv_batch, float clip,
    float* __restrict__ dz4, float* __restrict__ dz3, float* __restrict__ dz2, float* __restrict__ dz1,
    float* __restrict__ loss_part, int* __restrict__ flags, int epoch, int* __restrict__ err,
    unsigned long long* __restrict__ stamps)
{
    __shared__ __attribute__((aligned(16))) float lds[B3 ? FB_B3_LDS_FLOATS : FB_LDS_FLOATS];
    // bf16x3: the consumer's wait result lives in the arena's last word, past everything the backward
    // body uses (a separate word would push the allocation over a third of the CU's LDS)
    static_assert(FB_B3_LDS_FLOATS > BW_B3_LDS_FLOATS, "room for the flag word");
    __shared__ int ok_word;
    int& ok = B3 ? *reinterpret_cast<int*>(lds + FB_B3_LDS_FLOATS - 1) : ok_word;
    if (STAMP && threadIdx.x == 0) stamps[4L * blockIdx.x] = realtime_cu();
    const long tiles = (n + BM - 1) / BM;
    const long pad_tiles = (tiles + 7) & ~7L;              // consumers start at a multiple of 8: same XCD as their producer
    if ((long)blockIdx.x < pad_tiles) {
        const long tile = blockIdx.x;
        if (tile >= tiles) return;
        if (B3)
            forward_body_b3<false>(lds, tile, tiles, P, static_cast<const u16*>(PF), x, n, nullptr, nullptr, out_save, h1_save,
                                   h2_save, h3_save, nullptr, nullptr, nullptr, nullptr, nullptr);
        else
            forward_body<false>(lds, tile, tiles, P, static_cast<const float*>(PF), x, n, nullptr, nullptr, out_save, h1_save,
                                h2_save, h3_save, nullptr, nullptr, nullptr, nullptr, nullptr);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_waitcnt(0);                    // this thread's stores have been acknowledged by L2
        __syncthreads();
        if (threadIdx.x == 0) {
            const int xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xf;
            __hip_atomic_store(flags + tile, (epoch << 4) | xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (STAMP) stamps[4L * blockIdx.x + 1] = realtime_cu();
        }
    } else {
        const long tile = (long)blockIdx.x - pad_tiles;
        if (tile >= tiles) return;
        if (threadIdx.x == 0) {
            int state = 1;                                 // 0 ok, 1 flag never came, 2 producer on another XCD
            for (int poll = 0; poll < (1 << 21); ++poll) {
                const int f = __hip_atomic_load(flags + tile, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((f >> 4) == epoch) {
                    state = ((f & 0xf) == (int)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xf)) ? 0 : 2;
                    break;
                }
                __builtin_amdgcn_s_sleep(16);
            }
            ok = state;
            if (state) atomicMax(err, state);
        }
        __syncthreads();
        if (ok) return;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");     // ordering only: no cache maintenance (see above)
        if (STAMP && threadIdx.x == 0) stamps[4L * blockIdx.x + 2] = realtime_cu();
        if (B3)
            backward_body_b3(lds, tile, static_cast<const u16*>(PT), out_save, h1_save, h2_save, h3_save, action, old_logp, adv,
                             target, var, n, inv_batch, clip, dz4, dz3, dz2, dz1, loss_part);
        else
            backward_body(lds, tile, static_cast<const float*>(PT), out_save, h1_save, h2_save, h3_save, action, old_logp, adv,
                          target, var, n, inv_batch, clip, dz4, dz3, dz2, dz1, loss_part);
        if (STAMP && threadIdx.x == 0) {
            __builtin_amdgcn_s_waitcnt(0);
            stamps[4L * blockIdx.x + 1] = realtime_cu();
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Backward, part 2: weight gradients  dW[N][K] = dZ^T[N][rows] . A[rows][K],  db[N] = colsum(dZ).
// The reduction runs over the minibatch rows (40 960), so rows are split over workgroups: each
// workgroup accumulates its slab in MFMA accumulators and writes ONE partial [N][K(+bias)] block;
// a second small kernel sums the partials (fixed order: deterministic).  Both operands are read
// straight from the [rows][*] tiles in LDS with lane = output row/column, so every ds_read_b32
// is conflict-free.  Layers are concatenated along blockIdx.x with a work-proportional number of
// workgroups each (layer table in `GradWTable`).
struct GradWLayer {
    const float* dz;      // [rows][N]
    const float* a;       // [rows][Ka]  (row pitch Ka; K <= KP columns used, padded with zeros to KP)
    float* partial;       // [wgs][N*KP + N]: the layer's packed gradient block (weights, then biases) per workgroup
    int N, Ka, KP, wgs, first_block;
};
struct GradWTable { GradWLayer l[4]; };

#ifndef GW_PAD
#define GW_PAD 4
#endif
constexpr int GW_ROWS = 32;      // rows staged per chunk (16 MFMA k-steps)
constexpr int GW_THREADS = 1024; // 16 waves: 4 per SIMD, so a wave's LDS/barrier stalls hide behind three others

// One layer's slab.  Waves form a WN x WK grid over the (N/32) x (KPAD/32) output tiles; each wave
// owns TNW x TKW tiles so one pair of operand reads feeds TNW*TKW MFMAs.  Chunks of 32 rows are
// double-buffered in LDS: the global loads of chunk c+1 are issued (16-byte, coalesced: a chunk is
// one contiguous block of 32*N floats) before the MFMAs of chunk c and written to the other buffer
// afterwards, one barrier per chunk.
template <int N, int KA, int KPAD, int KOUT, int WN, int WK, int TNW, int TKW>
__device__ void grad_w_layer(const GradWLayer& L, long nrows, int wg, float* lds)
{
    static_assert(WN * TNW * 32 == N && WK * TKW * 32 == KPAD, "tile grid must cover the output");
    constexpr bool A_VEC = (KA % 4 == 0);
    constexpr int ZV = (GW_ROWS * N / 4 + GW_THREADS - 1) / GW_THREADS;          // float4 per thread
    constexpr int AV = A_VEC ? (GW_ROWS * KA / 4 + GW_THREADS - 1) / GW_THREADS
                             : (GW_ROWS * KA + GW_THREADS - 1) / GW_THREADS;     // float4 or float per thread
    // LDS pitches: the tile-fragment source hands a wave 32 ROWS of one 4-column group per load, so
    // the tiles it fills are padded (a row's two 16-byte pieces, lanes r and r + 32, then land 8 banks past the previous row's)
    constexpr int ZP = N + GW_PAD, AP = A_VEC ? KPAD + GW_PAD : KPAD;
    constexpr int BUF = GW_ROWS * (ZP + AP);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave index in an SGPR: weight bases become scalar
    const int r = lane & 31, h = lane >> 5;
    const bool active = wave < WN * WK;
    const int wn = active ? wave / WK : 0, wk = active ? wave % WK : 0;
    // slabs are whole 32-row tiles (the saved tensors are stored tile by tile)
    const long tiles = (nrows + GW_ROWS - 1) / GW_ROWS;
    const long rows_per = ((tiles + L.wgs - 1) / L.wgs) * GW_ROWS;
    const long rbeg = (long)wg * rows_per < nrows ? (long)wg * rows_per : nrows;
    const long rend = (rbeg + rows_per < nrows) ? rbeg + rows_per : nrows;
    const float* __restrict__ gz = L.dz;
    const float* __restrict__ ga = L.a;

    f32x16 acc[TNW][TKW];
#pragma unroll
    for (int a = 0; a < TNW; ++a)
#pragma unroll
        for (int b = 0; b < TKW; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.0f;
    float bsum = 0.0f;

    float4 zreg[ZV];
    float4 areg4[A_VEC ? AV : 1];
    float areg1[A_VEC ? 1 : AV];

    // tile-ordered float4 index i = [column tile i>>8][g (i>>6)&3][lane i&63] -> LDS offset of (row lane&31,
    // columns 32*ct + 8*g + 4*(lane>>5)); fixed per thread, computed once
    int zoff[ZV], aoff[A_VEC ? AV : 1];
#pragma unroll
    for (int v = 0; v < ZV; ++v) {
        const int i = tid + v * GW_THREADS;
        zoff[v] = (i & 31) * ZP + (i >> 8) * 32 + ((i >> 6) & 3) * 8 + ((i >> 5) & 1) * 4;
    }
    if (A_VEC) {
#pragma unroll
        for (int v = 0; v < AV; ++v) {
            const int i = tid + v * GW_THREADS;
            aoff[v] = (i & 31) * AP + (i >> 8) * 32 + ((i >> 6) & 3) * 8 + ((i >> 5) & 1) * 4;
        }
    }
    auto load_chunk = [&](long c0) {
#pragma unroll
        for (int v = 0; v < ZV; ++v) {
            const int i = tid + v * GW_THREADS;                 // float4 index inside the chunk = [column tile][g][lane]
            const long g = c0 + (i & 31);                       // its row
            zreg[v] = (i < GW_ROWS * N / 4 && g < rend) ? *reinterpret_cast<const float4*>(gz + c0 * N + 4L * i)
                                                       : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (A_VEC) {
#pragma unroll
            for (int v = 0; v < AV; ++v) {
                const int i = tid + v * GW_THREADS;
                const long g = c0 + (i & 31);
                areg4[v] = (i < GW_ROWS * KA / 4 && g < rend) ? *reinterpret_cast<const float4*>(ga + c0 * KA + 4L * i)
                                                            : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        } else {
#pragma unroll
            for (int v = 0; v < AV; ++v) {
                const int i = tid + v * GW_THREADS;
                const long g = c0 + i / KA;
                areg1[v] = (i < GW_ROWS * KA && g < rend) ? ga[c0 * KA + i] : 0.0f;
            }
        }
    };
    auto store_chunk = [&](float* buf) {
        float* bz = buf;
        float* ba = buf + GW_ROWS * ZP;
        // float4 index i of a tile-ordered chunk = [column tile i>>8][g (i>>6)&3][lane i&63]: row lane&31,
        // columns 32*ct + 8*g + 4*(lane>>5) .. +3
#pragma unroll
        for (int v = 0; v < ZV; ++v) {
            const int i = tid + v * GW_THREADS;
            if (i < GW_ROWS * N / 4) *reinterpret_cast<float4*>(bz + zoff[v]) = zreg[v];
        }
        if (A_VEC) {
#pragma unroll
            for (int v = 0; v < AV; ++v) {
                const int i = tid + v * GW_THREADS;
                if (i < GW_ROWS * KA / 4) *reinterpret_cast<float4*>(ba + aoff[v]) = areg4[v];
            }
        } else {
#pragma unroll
            for (int v = 0; v < AV; ++v) {
                const int i = tid + v * GW_THREADS;
                if (i < GW_ROWS * KA) {
                    const int rr = i / KA, cc = i - rr * KA;
                    ba[rr * AP + cc] = areg1[v];
                }
            }
        }
    };

    // zero the A padding columns once (KA < KPAD only for the input layer)
    if (KA < KPAD) {
        for (int i = tid; i < 2 * BUF; i += GW_THREADS) lds[i] = 0.0f;
        __syncthreads();
    }
    load_chunk(rbeg);
    store_chunk(lds);
    __syncthreads();
    int cur = 0;
    for (long c0 = rbeg; c0 < rend; c0 += GW_ROWS) {
        const bool more = c0 + GW_ROWS < rend;
        if (more) load_chunk(c0 + GW_ROWS);                     // in flight during the MFMAs below
        const float* bz = lds + cur * BUF;
        const float* ba = bz + GW_ROWS * ZP;
        {   // column sums of dZ (bias gradient): all threads take a slice of the 32 rows of one column
            constexpr int PARTS = (GW_THREADS / N) < 1 ? 1 : ((GW_THREADS / N) > GW_ROWS ? GW_ROWS : (GW_THREADS / N));
            constexpr int RPP = GW_ROWS / PARTS;
            const int colb = tid % N, part = tid / N;
            if (part < PARTS) {
#pragma unroll
                for (int rr = 0; rr < RPP; ++rr) bsum += bz[(part * RPP + rr) * ZP + colb];
            }
        }
        if (active) {
            const float* zp = bz + h * ZP + wn * (TNW * 32) + r;
            const float* ap = ba + h * AP + wk * (TKW * 32) + r;
            // operands of step st+1 are read from LDS before the MFMAs of step st are issued
            float zn[TNW], an[TKW];
#pragma unroll
            for (int a = 0; a < TNW; ++a) zn[a] = zp[32 * a];
#pragma unroll
            for (int b = 0; b < TKW; ++b) an[b] = ap[32 * b];
#pragma unroll
            for (int st = 0; st < GW_ROWS / 2; ++st) {
                float zv[TNW], av[TKW];
#pragma unroll
                for (int a = 0; a < TNW; ++a) zv[a] = zn[a];
#pragma unroll
                for (int b = 0; b < TKW; ++b) av[b] = an[b];
                if (st + 1 < GW_ROWS / 2) {
#pragma unroll
                    for (int a = 0; a < TNW; ++a) zn[a] = zp[2 * (st + 1) * ZP + 32 * a];
#pragma unroll
                    for (int b = 0; b < TKW; ++b) an[b] = ap[2 * (st + 1) * AP + 32 * b];
                }
                __builtin_amdgcn_sched_barrier(0);      // keep the prefetch reads ahead of this step's MFMAs
#pragma unroll
                for (int a = 0; a < TNW; ++a)
#pragma unroll
                    for (int b = 0; b < TKW; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(zv[a], av[b], acc[a][b], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (more) store_chunk(lds + (cur ^ 1) * BUF);
        __syncthreads();
        cur ^= 1;
    }
    float* out = L.partial + (long)wg * (N * KOUT + N);
    if (active) {
#pragma unroll
        for (int a = 0; a < TNW; ++a)
#pragma unroll
            for (int b = 0; b < TKW; ++b) {
                const int col = wk * (TKW * 32) + 32 * b + r;                   // k index
                if (col < KOUT) {
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) {
                        const int row = wn * (TNW * 32) + 32 * a + acc_row(reg, lane);   // n index
                        out[row * KOUT + col] = acc[a][b][reg];
                    }
                }
            }
    }
    {   // combine the per-thread slices of the bias gradient through LDS (fixed order)
        constexpr int PARTS = (GW_THREADS / N) < 1 ? 1 : ((GW_THREADS / N) > GW_ROWS ? GW_ROWS : (GW_THREADS / N));
        __syncthreads();
        if (tid / N < PARTS) lds[tid] = bsum;
        __syncthreads();
        if (tid < N) {
            float s = 0.0f;
#pragma unroll
            for (int p = 0; p < PARTS; ++p) s += lds[p * N + tid];
            out[N * KOUT + tid] = s;
        }
    }
}

__global__ __launch_bounds__(GW_THREADS) void mlp_grad_w_kernel(GradWTable T, long nrows)
{
    extern __shared__ __attribute__((aligned(16))) float lds_dyn[];
    const int b = blockIdx.x;
    if (b >= T.l[3].first_block)
        grad_w_layer<MLP_OUT, MLP_H3, MLP_H3, MLP_H3, 1, 4, 1, 1>(T.l[3], nrows, b - T.l[3].first_block, lds_dyn);
    else if (b >= T.l[2].first_block)
        grad_w_layer<MLP_H3, MLP_H2, MLP_H2, MLP_H2, 4, 4, 1, 1>(T.l[2], nrows, b - T.l[2].first_block, lds_dyn);
    else if (b >= T.l[1].first_block)
        grad_w_layer<MLP_H2, MLP_H1, MLP_H1, MLP_H1, 4, 4, 1, 2>(T.l[1], nrows, b - T.l[1].first_block, lds_dyn);
    else
        grad_w_layer<MLP_H1, MLP_IN, 96, MLP_IN_PAD, 4, 3, 2, 1>(T.l[0], nrows, b, lds_dyn);
}

// sum the per-workgroup partials into the packed gradient buffer.  Every partial slab has the
// layout of its layer's block of P (weights then biases), so this is an elementwise sum over
// slabs: a block owns 64 consecutive float4 (one per lane), its sixteen waves each take every sixteenth
// slab with all their 16-byte loads in flight, and the sixteen sums meet in LDS.  Fixed order.
// When `norm_ws` is given (single-rank runs: no all-reduce between here and the optimizer) the
// block also leaves the sum of squares of its masked gradient elements in norm_ws[1 + block] and
// block 0 advances the step counter, which saves the separate norm launch of the optimizer step.
constexpr int RED_WAVES = 16;
constexpr int RED_BLOCKS = (MLP_PACKED_FLOATS / 4 + 63) / 64;          // 291

__global__ __launch_bounds__(64 * RED_WAVES) void mlp_grad_reduce_kernel(GradWTable T, float* __restrict__ G,
                                                                         const float* __restrict__ mask,
                                                                         float* __restrict__ norm_ws, int* __restrict__ step)
{
    __shared__ float4 red[RED_WAVES][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = blockIdx.x * 64 + lane;                      // float4 index into the packed buffer
    const int o = 4 * q;
    float4 total = make_float4(0.f, 0.f, 0.f, 0.f);
    if (o < MLP_PACKED_FLOATS) {
        int layer, off_w, N, KP;
        if (o < MLP_OFF_W2) { layer = 0; off_w = MLP_OFF_W1; N = MLP_H1; KP = MLP_IN_PAD; }
        else if (o < MLP_OFF_W3) { layer = 1; off_w = MLP_OFF_W2; N = MLP_H2; KP = MLP_H1; }
        else if (o < MLP_OFF_W4) { layer = 2; off_w = MLP_OFF_W3; N = MLP_H3; KP = MLP_H2; }
        else { layer = 3; off_w = MLP_OFF_W4; N = MLP_OUT; KP = MLP_H3; }
        const float* part = layer == 0 ? T.l[0].partial : layer == 1 ? T.l[1].partial : layer == 2 ? T.l[2].partial : T.l[3].partial;
        const int wgs = layer == 0 ? T.l[0].wgs : layer == 1 ? T.l[1].wgs : layer == 2 ? T.l[2].wgs : T.l[3].wgs;
        const long stride = (long)N * KP + N;
        const float4* p4 = reinterpret_cast<const float4*>(part + (o - off_w));
        const long s4 = stride / 4;
        float4 acc4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) acc4[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int w0 = wave; w0 < wgs; w0 += 4 * RED_WAVES) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int w = w0 + u * RED_WAVES;
                v[u] = w < wgs ? p4[(long)w * s4] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) { acc4[u].x += v[u].x; acc4[u].y += v[u].y; acc4[u].z += v[u].z; acc4[u].w += v[u].w; }
        }
        total.x = (acc4[0].x + acc4[1].x) + (acc4[2].x + acc4[3].x);
        total.y = (acc4[0].y + acc4[1].y) + (acc4[2].y + acc4[3].y);
        total.z = (acc4[0].z + acc4[1].z) + (acc4[2].z + acc4[3].z);
        total.w = (acc4[0].w + acc4[1].w) + (acc4[2].w + acc4[3].w);
    }
    red[wave][lane] = total;
    __syncthreads();
    if (wave == 0) {
        float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
        if (o < MLP_PACKED_FLOATS) {
#pragma unroll
            for (int w = 0; w < RED_WAVES; ++w) { g.x += red[w][lane].x; g.y += red[w][lane].y; g.z += red[w][lane].z; g.w += red[w][lane].w; }
            *reinterpret_cast<float4*>(G + o) = g;
        }
        if (norm_ws) {
            float ss = 0.0f;
            if (o < MLP_PACKED_FLOATS) {
                const float4 mk = *reinterpret_cast<const float4*>(mask + o);
                const float a = g.x * mk.x, b = g.y * mk.y, c = g.z * mk.z, d = g.w * mk.w;
                ss = (a * a + b * b) + (c * c + d * d);
            }
            for (int off = 32; off > 0; off >>= 1) ss += __shfl_down(ss, off, 64);
            if (lane == 0) {
                norm_ws[1 + blockIdx.x] = ss;
                if (blockIdx.x == 0) *step += 1;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Optimizer step (ppo.py:196-199): clip_grad_norm_(max_norm) + Adam (torch defaults: no weight
// decay, no amsgrad) over the packed parameter buffer, plus the refresh of the transposed weights
// the next backward pass streams.  Two small launches: per-block sums of squares of the masked,
// scaled gradient (fixed order: deterministic), then the update, where every block re-adds the
// 73 block sums in the same order and so derives the same clip coefficient.  The step counter
// lives in device memory (incremented by the first launch), so both are graph-capturable.
constexpr int ADAM_THREADS = 1024;
constexpr int ADAM_BLOCKS = (MLP_PACKED_FLOATS + ADAM_THREADS - 1) / ADAM_THREADS;   // 73

__global__ __launch_bounds__(ADAM_THREADS) void mlp_adam_norm_kernel(const float* __restrict__ G,
                                                                     const float* __restrict__ mask, float grad_scale,
                                                                     float* __restrict__ norm_ws, int* __restrict__ step)
{
    __shared__ float red[16];
    const int tid = threadIdx.x;
    const int i = blockIdx.x * ADAM_THREADS + tid;
    float ss = 0.0f;
    if (i < MLP_PACKED_FLOATS) { const float g = G[i] * grad_scale * mask[i]; ss = g * g; }
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_down(ss, o, 64);
    if ((tid & 63) == 0) red[tid >> 6] = ss;
    __syncthreads();
    if (tid == 0) {
        float t = 0.0f;
        for (int w = 0; w < 16; ++w) t += red[w];
        norm_ws[1 + blockIdx.x] = t;
        if (blockIdx.x == 0) *step += 1;
    }
}

__global__ __launch_bounds__(ADAM_THREADS) void mlp_adam_apply_kernel(float* __restrict__ P, float* __restrict__ PF,
                                                                      float* __restrict__ PT,
                                                                      const int* __restrict__ idx_f,
                                                                      const int* __restrict__ idx_t,
                                                                      const float* __restrict__ G,
                                                                      const float* __restrict__ mask,
                                                                      float* __restrict__ m, float* __restrict__ v,
                                                                      const int* __restrict__ step, float lr, float beta1,
                                                                      float beta2, float eps, float max_norm,
                                                                      float grad_scale, float* __restrict__ norm_ws,
                                                                      int nparts, float part_scale,
                                                                      u16* __restrict__ PB, u16* __restrict__ PTB,
                                                                      const int* __restrict__ idx_fb,
                                                                      const int* __restrict__ idx_tb)
{
    __shared__ float s_coef, s_step_size, s_bc2_sqrt;
    __shared__ float red[16];
    const int tid = threadIdx.x;
    {   // every block re-adds the same partial sums in the same order: identical clip coefficient
        float t = 0.0f;
        for (int b = tid; b < nparts; b += ADAM_THREADS) t += norm_ws[1 + b];
        for (int o = 32; o > 0; o >>= 1) t += __shfl_down(t, o, 64);
        if ((tid & 63) == 0) red[tid >> 6] = t;
    }
    __syncthreads();
    if (tid == 0) {
        float t = 0.0f;
        for (int w = 0; w < 16; ++w) t += red[w];
        t *= part_scale;                                    // partials of the unscaled gradient: scale^2
        const float norm = sqrtf(t);
        const float coef = max_norm / (norm + 1e-6f);          // torch.nn.utils.clip_grad_norm_
        s_coef = coef < 1.0f ? coef : 1.0f;
        if (blockIdx.x == 0) norm_ws[0] = norm;
        // bias corrections of torch.optim.Adam, once per workgroup (two powf per thread otherwise)
        const float ts = (float)*step;
        const float bc1 = 1.0f - powf(beta1, ts);
        const float bc2 = 1.0f - powf(beta2, ts);
        s_step_size = lr / bc1;
        s_bc2_sqrt = sqrtf(bc2);
    }
    __syncthreads();
    const int i = blockIdx.x * ADAM_THREADS + tid;
    if (i >= MLP_PACKED_FLOATS) return;
    const float coef = s_coef * grad_scale;
    const float step_size = s_step_size;
    const float bc2_sqrt = s_bc2_sqrt;
    const float mk = mask[i];
    const float g = G[i] * coef * mk;
    const float mi = beta1 * m[i] + (1.0f - beta1) * g;
    const float vi = beta2 * v[i] + (1.0f - beta2) * g * g;
    m[i] = mi; v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    const float p = P[i] - mk * (step_size * (mi / denom));
    P[i] = p;
    // keep the fragment-ordered copies the kernels stream in step with the master weights
    const int jf = idx_f[i], jt = idx_t[i];
    if (jf >= 0) PF[jf] = p;
    if (jt >= 0) PT[jt] = p;
    if (PB) {       // and the three-term bf16 planes of the bf16x3 GEMM path
        u16 a, b, c;
        split3(p, a, b, c);
        const int kf = idx_fb[i], kt = idx_tb[i];
        if (kf >= 0) { PB[kf] = a; PB[kf + 512] = b; PB[kf + 1024] = c; }
        if (kt >= 0) { PTB[kt] = a; PTB[kt + 512] = b; PTB[kt + 1024] = c; }
    }
}

}  // namespace

extern "C" hipError_t flyhip_launch_mlp_forward(const float* P, const float* PF, const float* x, int64_t n, float* mu_out,
                                                float* v_out, float* out_save, float* h1_save, float* h2_save,
                                                float* h3_save, const uint16_t* PB, void* stream)
{
    if (PB) {       // bf16x3 GEMMs
        const long tiles = (n + BM - 1) / BM;
        const int grid = (int)tiles;
        hipLaunchKernelGGL(mlp_forward_b3_kernel, dim3(grid), dim3(THREADS), 0, (hipStream_t)stream, P, PB, x, (long)n, mu_out,
                           v_out, out_save, h1_save, h2_save, h3_save, (const float*)nullptr, (const float*)nullptr,
                           (float*)nullptr, (float*)nullptr, 0, 0.0f, 0.0f);
        return hipGetLastError();
    }
    // Large inputs (the critic pass over the whole rollout) run persistent workgroups, 3 per CU, each
    // walking many tiles with the next tile's rows prefetched.  Up to a few tiles per slot the
    // hardware's dynamic workgroup dispatch balances better than a fixed walk (1280 tiles over 768
    // slots would leave part of the chip idle for the second half), so those launch one tile each.
    const long tiles = (n + BM - 1) / BM;
    const int grid = (int)(tiles <= 4 * PERSIST_GRID ? tiles : PERSIST_GRID);
    hipLaunchKernelGGL(mlp_forward_kernel<false>, dim3(grid), dim3(THREADS), 0, (hipStream_t)stream, P, PF, x, (long)n,
                       mu_out, v_out, out_save, h1_save, h2_save, h3_save, (const float*)nullptr, (const float*)nullptr,
                       (float*)nullptr, (float*)nullptr, (unsigned long long*)nullptr, 0, 0.0f, 0.0f);
    return hipGetLastError();
}

extern "C" hipError_t flyhip_launch_mlp_forward_sample(const float* P, const float* PF, const float* x, int64_t n,
                                                       const float* eps, const float* var, int var_steps,
                                                       float var_decay, float var_min, float* act_out,
                                                       float* logp_out, float* mu_out, float* v_out, const uint16_t* PB,
                                                       void* stream)
{
    if (PB) {
        const long tiles = (n + BM - 1) / BM;
        const int grid = (int)tiles;
        hipLaunchKernelGGL(mlp_forward_b3_kernel, dim3(grid), dim3(THREADS), 0, (hipStream_t)stream, P, PB, x, (long)n, mu_out,
                           v_out, (float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr, eps, var, act_out,
                           logp_out, var_steps, var_decay, var_min);
        return hipGetLastError();
    }
    const long tiles = (n + BM - 1) / BM;
    const int grid = (int)(tiles <= 4 * PERSIST_GRID ? tiles : PERSIST_GRID);
    hipLaunchKernelGGL(mlp_forward_kernel<false>, dim3(grid), dim3(THREADS), 0, (hipStream_t)stream, P, PF, x, (long)n,
                       mu_out, v_out, (float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr, eps,
                       var, act_out, logp_out, (unsigned long long*)nullptr, var_steps, var_decay, var_min);
    return hipGetLastError();
}

// diagnostic build of the same kernel with phase stamps (tools/stamp_forward.py); not part of the ABI header
extern "C" int flyhip_debug_mlp_forward_stamped(const float* P, const float* PF, const float* x, int64_t n,
                                                float* out_save, float* h1_save, float* h2_save, float* h3_save,
                                                unsigned long long* stamps, void* stream, int grid_override)
{
    const long tiles = (n + BM - 1) / BM;
    const int grid = grid_override > 0 ? grid_override : (int)(tiles <= 4 * PERSIST_GRID ? tiles : PERSIST_GRID);
    hipLaunchKernelGGL(mlp_forward_kernel<true>, dim3(grid), dim3(THREADS), 0, (hipStream_t)stream, P, PF, x, (long)n,
                       (float*)nullptr, (float*)nullptr, out_save, h1_save, h2_save, h3_save, (const float*)nullptr,
                       (const float*)nullptr, (float*)nullptr, (float*)nullptr, stamps, 0, 0.0f, 0.0f);
    return (int)hipGetLastError();
}

extern "C" hipError_t flyhip_launch_mlp_backward_dx(const float* PT, const float* out_saved, const float* h1,
                                                    const float* h2, const float* h3, const float* action,
                                                    const float* old_logp, const float* adv, const float* target,
                                                    const float* var, int64_t n, float inv_batch, float clip,
                                                    float* dz4, float* dz3, float* dz2, float* dz1, float* loss_part,
                                                    const uint16_t* PTB, void* stream)
{
    if (PTB) {
        hipLaunchKernelGGL(mlp_backward_dx_b3_kernel, dim3((unsigned)((n + BM - 1) / BM)), dim3(THREADS), 0, (hipStream_t)stream,
                           PTB, out_saved, h1, h2, h3, action, old_logp, adv, target, var, (long)n, inv_batch, clip, dz4, dz3,
                           dz2, dz1, loss_part);
        return hipGetLastError();
    }
    const int grid = (int)((n + BM - 1) / BM);
    hipLaunchKernelGGL(mlp_backward_dx_kernel, dim3(grid), dim3(THREADS), 0, (hipStream_t)stream, PT, out_saved,
                       h1, h2, h3, action, old_logp, adv, target, var, (long)n, inv_batch, clip, dz4, dz3, dz2, dz1,
                       loss_part);
    return hipGetLastError();
}

// workgroups per layer, proportional to the layer's share of the dW FLOPs (256 in total)
// slabs are whole 32-row tiles; a chunk costs its MFMAs plus a fixed staging/barrier overhead, so the thin layers
// (3 and 4) get more workgroups than their FLOP share: measured best of a dozen splits at 1280 tiles
static int kGradWgs[4] = {80, 100, 52, 24};
static bool gw_env_read = false;
static void gw_read_env()
{   // tuning aid: FLYHIP_GW_SPLIT="a,b,c,d" overrides the split (sum <= 256)
    if (gw_env_read) return;
    gw_env_read = true;
    const char* e = getenv("FLYHIP_GW_SPLIT");
    int v[4];
    if (e && sscanf(e, "%d,%d,%d,%d", &v[0], &v[1], &v[2], &v[3]) == 4 && v[0] > 0 && v[1] > 0 && v[2] > 0 && v[3] > 0 &&
        v[0] + v[1] + v[2] + v[3] <= 1024)
        for (int i = 0; i < 4; ++i) kGradWgs[i] = v[i];
}

extern "C" hipError_t flyhip_launch_mlp_fwd_bwd(const float* P, const float* PF, const float* PT, const float* x, int64_t n,
                                                float* out_save, float* h1_save, float* h2_save, float* h3_save,
                                                const float* action, const float* old_logp, const float* adv,
                                                const float* target, const float* var, float inv_batch, float clip,
                                                float* dz4, float* dz3, float* dz2, float* dz1, float* loss_part,
                                                int* flags, int epoch, int* err, const uint16_t* PB, const uint16_t* PTB,
                                                void* stream)
{
    const long tiles = (n + BM - 1) / BM;
    const long pad_tiles = (tiles + 7) & ~7L;
    if (PB && PTB)
        hipLaunchKernelGGL((mlp_fwd_bwd_kernel<false, true>), dim3((unsigned)(pad_tiles + tiles)), dim3(THREADS), 0,
                           (hipStream_t)stream, P, (const void*)PB, (const void*)PTB, x, (long)n, out_save, h1_save, h2_save,
                           h3_save, action, old_logp, adv, target, var, inv_batch, clip, dz4, dz3, dz2, dz1, loss_part, flags,
                           epoch, err, (unsigned long long*)nullptr);
    else
        hipLaunchKernelGGL((mlp_fwd_bwd_kernel<false, false>), dim3((unsigned)(pad_tiles + tiles)), dim3(THREADS), 0,
                           (hipStream_t)stream, P, (const void*)PF, (const void*)PT, x, (long)n, out_save, h1_save, h2_save,
                           h3_save, action, old_logp, adv, target, var, inv_batch, clip, dz4, dz3, dz2, dz1, loss_part, flags,
                           epoch, err, (unsigned long long*)nullptr);
    return hipGetLastError();
}

// diagnostic build with per-workgroup start/end stamps (tools/stamp_fwd_bwd.py); not part of the ABI header
extern "C" int flyhip_debug_mlp_fwd_bwd_stamped(const float* P, const float* PF, const float* PT, const float* x, int64_t n,
                                                float* out_save, float* h1_save, float* h2_save, float* h3_save,
                                                const float* action, const float* old_logp, const float* adv,
                                                const float* target, const float* var, float inv_batch, float clip,
                                                float* dz4, float* dz3, float* dz2, float* dz1, float* loss_part,
                                                int* flags, int epoch, int* err, unsigned long long* stamps, void* stream)
{
    const long tiles = (n + BM - 1) / BM;
    const long pad_tiles = (tiles + 7) & ~7L;
    hipLaunchKernelGGL((mlp_fwd_bwd_kernel<true, false>), dim3((unsigned)(pad_tiles + tiles)), dim3(THREADS), 0, (hipStream_t)stream, P,
                       (const void*)PF, (const void*)PT, x, (long)n, out_save, h1_save, h2_save, h3_save, action, old_logp, adv, target, var, inv_batch,
                       clip, dz4, dz3, dz2, dz1, loss_part, flags, epoch, err, stamps);
    return (int)hipGetLastError();
}

extern "C" int64_t flyhip_mlp_grad_workspace_floats(void)
{
    gw_read_env();
    return (int64_t)kGradWgs[0] * MLP_H1 * (MLP_IN_PAD + 1) + (int64_t)kGradWgs[1] * MLP_H2 * (MLP_H1 + 1) +
           (int64_t)kGradWgs[2] * MLP_H3 * (MLP_H2 + 1) + (int64_t)kGradWgs[3] * MLP_OUT * (MLP_H3 + 1);   // N*KP + N per slab
}

extern "C" hipError_t flyhip_launch_mlp_grad_w(const float* x, const float* h1, const float* h2, const float* h3,
                                               const float* dz1, const float* dz2, const float* dz3, const float* dz4,
                                               int64_t n, float* workspace, float* grad_out, const float* norm_mask,
                                               float* norm_ws, int* norm_step, void* stream)
{
    gw_read_env();
    GradWTable T;
    const float* dz[4] = {dz1, dz2, dz3, dz4};
    const float* a[4] = {x, h1, h2, h3};
    const int N[4] = {MLP_H1, MLP_H2, MLP_H3, MLP_OUT};
    const int Ka[4] = {MLP_IN, MLP_H1, MLP_H2, MLP_H3};
    const int KP[4] = {MLP_IN_PAD, MLP_H1, MLP_H2, MLP_H3};
    float* ws = workspace;
    int first = 0;
    for (int l = 0; l < 4; ++l) {
        T.l[l].dz = dz[l]; T.l[l].a = a[l]; T.l[l].partial = ws;
        T.l[l].N = N[l]; T.l[l].Ka = Ka[l]; T.l[l].KP = KP[l]; T.l[l].wgs = kGradWgs[l]; T.l[l].first_block = first;
        ws += (long)kGradWgs[l] * ((long)N[l] * KP[l] + N[l]);
        first += kGradWgs[l];
    }
    // dynamic LDS: two buffers of the largest layer's chunk (padded pitches): 2 x 32 x (136 + 264) floats = 100 KiB
    const size_t lds_bytes = sizeof(float) * 2 * GW_ROWS * (MLP_H2 + GW_PAD + MLP_H1 + GW_PAD);
    static_assert(MLP_H1 + GW_PAD + 96 <= MLP_H2 + GW_PAD + MLP_H1 + GW_PAD, "layer 1 chunk fits");
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_grad_w_kernel),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (ea != hipSuccess) return ea;
        attr_set = true;
    }
    hipLaunchKernelGGL(mlp_grad_w_kernel, dim3(first), dim3(GW_THREADS), lds_bytes, (hipStream_t)stream, T, (long)n);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(mlp_grad_reduce_kernel, dim3(RED_BLOCKS), dim3(64 * RED_WAVES), 0,
                       (hipStream_t)stream, T, grad_out, norm_mask, norm_ws, norm_step);
    return hipGetLastError();
}

extern "C" int flyhip_mlp_reduce_blocks(void) { return RED_BLOCKS; }

extern "C" hipError_t flyhip_launch_mlp_adam(float* P, float* PF, float* PT, const int* idx_f, const int* idx_t,
                                             const float* G, const float* mask, float* m,
                                             float* v, int* step, float lr, float beta1, float beta2, float eps,
                                             float max_norm, float grad_scale, float* norm_ws, int norm_ready,
                                             uint16_t* PB, uint16_t* PTB, const int* idx_fb, const int* idx_tb,
                                             void* stream)
{
    int nparts = ADAM_BLOCKS;
    float part_scale = 1.0f;
    if (norm_ready) {               // mlp_grad_w already left per-block sums of squares (unscaled) and advanced the step
        nparts = RED_BLOCKS;
        part_scale = grad_scale * grad_scale;
    } else {
        hipLaunchKernelGGL(mlp_adam_norm_kernel, dim3(ADAM_BLOCKS), dim3(ADAM_THREADS), 0, (hipStream_t)stream, G, mask,
                           grad_scale, norm_ws, step);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(mlp_adam_apply_kernel, dim3(ADAM_BLOCKS), dim3(ADAM_THREADS), 0, (hipStream_t)stream, P, PF, PT,
                       idx_f, idx_t, G, mask, m, v, step, lr, beta1, beta2, eps, max_norm, grad_scale, norm_ws, nparts,
                       part_scale, PB, PTB, idx_fb, idx_tb);
    return hipGetLastError();
}
