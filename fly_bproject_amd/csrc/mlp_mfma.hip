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
#include "flyhip.h"
#include "mlp_layout.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 32;          // rows per workgroup
constexpr int NWAVE = 4;
constexpr int THREADS = 64 * NWAVE;

__device__ __forceinline__ float elu(float x) { return x > 0.0f ? x : expm1f(x); }
// derivative of ELU expressed through its OUTPUT y: 1 for y > 0, y + 1 otherwise
__device__ __forceinline__ float elu_grad_from_out(float y) { return y > 0.0f ? 1.0f : y + 1.0f; }

// acc[t] += A[32 x K] * W[col tile t][K]^T for this wave's NT column tiles.
//   lds_in : [32][K+4] floats (row-major, k contiguous)
//   W      : global, row-major [N][K]; `col0` = first output column of this wave
// k mapping: MFMA step s = 4*kq+q multiplies k = 4*kq+q (lanes 0..31) and k = K/2+4*kq+q (32..63).
template <int K, int NT>
__device__ __forceinline__ void tile_gemm(const float* __restrict__ W, int col0, const float* lds_in,
                                          f32x16 (&acc)[NT], int lane)
{
    const int r = lane & 31, h = lane >> 5;
    const float* ap = lds_in + r * (K + 4) + h * (K / 2);
    const float* bp[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) bp[t] = W + (long)(col0 + 32 * t + r) * K + h * (K / 2);
    float4 bnext[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) bnext[t] = *reinterpret_cast<const float4*>(bp[t]);
#pragma unroll 4
    for (int kq = 0; kq < K / 8; ++kq) {
        float4 b[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) b[t] = bnext[t];
        if (kq + 1 < K / 8) {
#pragma unroll
            for (int t = 0; t < NT; ++t) bnext[t] = *reinterpret_cast<const float4*>(bp[t] + 4 * (kq + 1));
        }
        const float4 a = *reinterpret_cast<const float4*>(ap + 4 * kq);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b[t].x, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b[t].y, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b[t].z, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b[t].w, acc[t], 0, 0, 0);
        }
    }
}

// C/D layout of the 32x32 tile: lane holds column (lane&31), rows (reg&3) + 8*(reg>>2) + 4*(lane>>5).
__device__ __forceinline__ int acc_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

// bias + ELU epilogue: writes the activation tile to LDS (next layer's A operand) and, when
// `save` is non-null, to global [rows][N] for the backward pass.
template <int N, int NT>
__device__ __forceinline__ void epilogue_elu(const f32x16 (&acc)[NT], const float* __restrict__ bias, int col0,
                                             float* lds_out, float* __restrict__ save, long row0, long nrows, int lane)
{
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int col = col0 + 32 * t + (lane & 31);
        const float bv = bias[col];
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int row = acc_row(reg, lane);
            const float y = elu(acc[t][reg] + bv);
            lds_out[row * (N + 4) + col] = y;
            if (save && row0 + row < nrows) save[(row0 + row) * N + col] = y;
        }
    }
}

template <int NT>
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[NT])
{
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.0f;
}

constexpr int LDS_A_FLOATS = BM * (MLP_H1 + 4);      // H1, later H3
constexpr int LDS_B_FLOATS = BM * (MLP_H2 + 4);      // X0, later H2, later the split-K partials

// x [n][73] -> out [n][32] (cols 0..17 = mean after ELU, col 18 = value, rest 0).
// mu_out [n][18] / v_out [n] / h*_save are optional.
__global__ __launch_bounds__(THREADS) void mlp_forward_kernel(
    const float* __restrict__ P, const float* __restrict__ x, long n,
    float* __restrict__ mu_out, float* __restrict__ v_out, float* __restrict__ out_save,
    float* __restrict__ h1_save, float* __restrict__ h2_save, float* __restrict__ h3_save)
{
    __shared__ __attribute__((aligned(16))) float lds[LDS_A_FLOATS + LDS_B_FLOATS];
    float* ldsA = lds;
    float* ldsB = lds + LDS_A_FLOATS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long row0 = (long)blockIdx.x * BM;

    // stage the 32 x 73 input tile (contiguous in HBM) as [32][80+4], zero padded
    for (int i = tid; i < BM * (MLP_IN_PAD + 4); i += THREADS) ldsB[i] = 0.0f;
    __syncthreads();
    {
        const long base = row0 * MLP_IN;
        const long lim = n * MLP_IN;
        for (int i = tid; i < BM * MLP_IN; i += THREADS) {
            const int rr = i / MLP_IN, cc = i - rr * MLP_IN;
            if (base + i < lim) ldsB[rr * (MLP_IN_PAD + 4) + cc] = x[base + i];
        }
    }
    __syncthreads();

    {   // L1: 80 -> 256, wave owns 64 columns
        f32x16 acc[2];
        zero_acc(acc);
        tile_gemm<MLP_IN_PAD, 2>(P + MLP_OFF_W1, wave * 64, ldsB, acc, lane);
        epilogue_elu<MLP_H1, 2>(acc, P + MLP_OFF_B1, wave * 64, ldsA, h1_save, row0, n, lane);
    }
    __syncthreads();
    {   // L2: 256 -> 128, wave owns 32 columns
        f32x16 acc[1];
        zero_acc(acc);
        tile_gemm<MLP_H1, 1>(P + MLP_OFF_W2, wave * 32, ldsA, acc, lane);
        epilogue_elu<MLP_H2, 1>(acc, P + MLP_OFF_B2, wave * 32, ldsB, h2_save, row0, n, lane);
    }
    __syncthreads();
    {   // L3: 128 -> 128 (actor | critic heads stacked)
        f32x16 acc[1];
        zero_acc(acc);
        tile_gemm<MLP_H2, 1>(P + MLP_OFF_W3, wave * 32, ldsB, acc, lane);
        epilogue_elu<MLP_H3, 1>(acc, P + MLP_OFF_B3, wave * 32, ldsA, h3_save, row0, n, lane);
    }
    __syncthreads();
    {   // L4: 128 -> 32, split-K over the four waves (32 k each), partials reduced through LDS
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
        const int r = lane & 31, h = lane >> 5;
        const float* ap = ldsA + r * (MLP_H3 + 4) + wave * 32 + h * 16;
        const float* bp = P + MLP_OFF_W4 + (long)r * MLP_H3 + wave * 32 + h * 16;
#pragma unroll
        for (int kq = 0; kq < 4; ++kq) {
            const float4 a = *reinterpret_cast<const float4*>(ap + 4 * kq);
            const float4 b = *reinterpret_cast<const float4*>(bp + 4 * kq);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
        }
        float* part = ldsB + wave * (BM * MLP_OUT);
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) part[acc_row(reg, lane) * MLP_OUT + r] = acc[reg];
    }
    __syncthreads();
    for (int i = tid; i < BM * MLP_OUT; i += THREADS) {
        const int row = i / MLP_OUT, col = i - row * MLP_OUT;
        float z = ((ldsB[i] + ldsB[BM * MLP_OUT + i]) + ldsB[2 * BM * MLP_OUT + i]) + ldsB[3 * BM * MLP_OUT + i];
        z += P[MLP_OFF_B4 + col];
        float y = (col < MLP_NACT) ? elu(z) : ((col == MLP_NACT) ? z : 0.0f);   // ELU on the mean (ppo.py:30), none on v
        const long grow = row0 + row;
        if (grow < n) {
            if (out_save) out_save[grow * MLP_OUT + col] = y;
            if (mu_out && col < MLP_NACT) mu_out[grow * MLP_NACT + col] = y;
            if (v_out && col == MLP_NACT) v_out[grow] = y;
        }
    }
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
__device__ __forceinline__ void epilogue_dact(const f32x16& acc, int col, const float* __restrict__ h_saved, int N,
                                              float* lds_out, float* __restrict__ dz_out, long row0, long nrows, int lane)
{
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int row = acc_row(reg, lane);
        const long grow = row0 + row;
        const float y = (grow < nrows) ? h_saved[grow * N + col] : 0.0f;
        const float d = acc[reg] * elu_grad_from_out(y);
        if (lds_out) lds_out[row * (N + 4) + col] = d;
        if (grow < nrows) dz_out[grow * N + col] = d;
    }
}

__global__ __launch_bounds__(THREADS) void mlp_backward_dx_kernel(
    const float* __restrict__ PT, const float* __restrict__ out_saved, const float* __restrict__ h1_saved,
    const float* __restrict__ h2_saved, const float* __restrict__ h3_saved,
    const float* __restrict__ action, const float* __restrict__ old_logp, const float* __restrict__ adv,
    const float* __restrict__ target, const float* __restrict__ var, long n, float inv_batch, float clip,
    float* __restrict__ dz4, float* __restrict__ dz3, float* __restrict__ dz2, float* __restrict__ dz1,
    float* __restrict__ loss_part)
{
    __shared__ __attribute__((aligned(16))) float lds[BM * (MLP_OUT + 4) + 2 * BM * (MLP_H3 + 4) + BM + 8];
    float* ldsZ4 = lds;                                   // [32][36]
    float* ldsZ3 = lds + BM * (MLP_OUT + 4);              // [32][132]
    float* ldsZ2 = ldsZ3 + BM * (MLP_H3 + 4);             // [32][132]
    float* coef = ldsZ2 + BM * (MLP_H2 + 4);              // [32] per-row d loss / d logp
    float* red = coef + BM;                               // [8] loss partials
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long row0 = (long)blockIdx.x * BM;

    // per-row loss terms (one lane per row)
    if (tid < BM) {
        const long g = row0 + tid;
        float c = 0.0f, pol = 0.0f, hub = 0.0f;
        if (g < n) {
            float M = 0.0f, half_log_det = 0.0f;
#pragma unroll
            for (int j = 0; j < MLP_NACT; ++j) {
                const float L = sqrtf(var[j]);
                const float xj = (action[g * MLP_NACT + j] - out_saved[g * MLP_OUT + j]) / L;
                M += xj * xj;
                half_log_det += logf(L);
            }
            const float logp = -0.5f * (33.08178959434617f + M) - half_log_det;
            const float ratio = expf(logp - old_logp[g]);
            const float A = adv[g];
            const float s1 = ratio * A;
            const float rc = fminf(fmaxf(ratio, 1.0f - clip), 1.0f + clip);
            const float s2 = rc * A;
            const float in_range = (ratio >= 1.0f - clip && ratio <= 1.0f + clip) ? 1.0f : 0.0f;
            float dmin;                                    // d min(s1,s2) / d ratio
            if (s1 < s2) dmin = A;
            else if (s1 > s2) dmin = A * in_range;
            else dmin = 0.5f * (A + A * in_range);
            c = -inv_batch * ratio * dmin;                 // d loss / d logp
            pol = -fminf(s1, s2);
            const float d = out_saved[g * MLP_OUT + MLP_NACT] - target[g];
            hub = fabsf(d) < 1.0f ? 0.5f * d * d : fabsf(d) - 0.5f;
        }
        coef[tid] = c;
        // wave 0 holds rows 0..31 in lanes 0..31: reduce the two loss sums in-wave
        for (int o = 16; o > 0; o >>= 1) { pol += __shfl_down(pol, o, 32); hub += __shfl_down(hub, o, 32); }
        if (tid == 0) { red[0] = pol; red[1] = hub; }
    }
    __syncthreads();
    if (tid == 0 && loss_part) { loss_part[2 * blockIdx.x] = red[0]; loss_part[2 * blockIdx.x + 1] = red[1]; }
    // dz4 tile
    for (int i = tid; i < BM * MLP_OUT; i += THREADS) {
        const int row = i / MLP_OUT, col = i - row * MLP_OUT;
        const long g = row0 + row;
        float d = 0.0f;
        if (g < n) {
            if (col < MLP_NACT) {
                const float mu = out_saved[g * MLP_OUT + col];
                d = coef[row] * (action[g * MLP_NACT + col] - mu) / var[col] * elu_grad_from_out(mu);
            } else if (col == MLP_NACT) {
                const float dv = out_saved[g * MLP_OUT + MLP_NACT] - target[g];
                d = inv_batch * fminf(fmaxf(dv, -1.0f), 1.0f);           // smooth_l1', beta = 1
            }
            dz4[g * MLP_OUT + col] = d;
        }
        ldsZ4[row * (MLP_OUT + 4) + col] = d;
    }
    __syncthreads();
    {   // dA3 = dZ4 . W4  ->  dZ3
        f32x16 acc[1];
        zero_acc(acc);
        tile_gemm<MLP_OUT, 1>(PT + MLP_OFF_WT4, wave * 32, ldsZ4, acc, lane);
        epilogue_dact(acc[0], wave * 32 + (lane & 31), h3_saved, MLP_H3, ldsZ3, dz3, row0, n, lane);
    }
    __syncthreads();
    {   // dA2 = dZ3 . W3  ->  dZ2
        f32x16 acc[1];
        zero_acc(acc);
        tile_gemm<MLP_H3, 1>(PT + MLP_OFF_WT3, wave * 32, ldsZ3, acc, lane);
        epilogue_dact(acc[0], wave * 32 + (lane & 31), h2_saved, MLP_H2, ldsZ2, dz2, row0, n, lane);
    }
    __syncthreads();
    {   // dA1 = dZ2 . W2  ->  dZ1
        f32x16 acc[2];
        zero_acc(acc);
        tile_gemm<MLP_H2, 2>(PT + MLP_OFF_WT2, wave * 64, ldsZ2, acc, lane);
        epilogue_dact(acc[0], wave * 64 + (lane & 31), h1_saved, MLP_H1, nullptr, dz1, row0, n, lane);
        epilogue_dact(acc[1], wave * 64 + 32 + (lane & 31), h1_saved, MLP_H1, nullptr, dz1, row0, n, lane);
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
    float* partial;       // [wgs][N][KP + 1]  (last column: bias gradient)
    int N, Ka, KP, wgs, first_block;
};
struct GradWTable { GradWLayer l[4]; };

constexpr int GW_ROWS = 32;      // rows staged per chunk

template <int N, int KP>
__device__ void grad_w_layer(const GradWLayer& L, long nrows, int wg, float* lds)
{
    // 8 waves; output tiles (N/32) x (KP/32... KP may be 80 -> 3 tiles, last partial)
    constexpr int TN = N / 32;
    constexpr int TK = (KP + 31) / 32;
    constexpr int TILES = TN * TK;
    constexpr int WAVES = 8;
    constexpr int PER = (TILES + WAVES - 1) / WAVES;
    float* ldsZ = lds;                       // [32][N]
    float* ldsA = lds + GW_ROWS * N;         // [32][TK*32]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const long rows_per = (nrows + L.wgs - 1) / L.wgs;
    const long rbeg = (long)wg * rows_per;
    const long rend = (rbeg + rows_per < nrows) ? rbeg + rows_per : nrows;
    f32x16 acc[PER];
#pragma unroll
    for (int t = 0; t < PER; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.0f;
    float bsum = 0.0f;                        // thread tid < N: column sum of dZ
    for (long c0 = rbeg; c0 < rend; c0 += GW_ROWS) {
        __syncthreads();
        for (int i = tid; i < GW_ROWS * N; i += 512) {
            const int rr = i / N, cc = i - rr * N;
            const long g = c0 + rr;
            ldsZ[i] = (g < rend) ? L.dz[g * N + cc] : 0.0f;
        }
        for (int i = tid; i < GW_ROWS * TK * 32; i += 512) {
            const int rr = i / (TK * 32), cc = i - rr * (TK * 32);
            const long g = c0 + rr;
            ldsA[i] = (g < rend && cc < L.Ka) ? L.a[g * L.Ka + cc] : 0.0f;
        }
        __syncthreads();
        if (tid < N) {
#pragma unroll 8
            for (int rr = 0; rr < GW_ROWS; ++rr) bsum += ldsZ[rr * N + tid];
        }
#pragma unroll
        for (int t = 0; t < PER; ++t) {
            const int tile = wave + WAVES * t;
            if (tile < TILES) {
                const int tn = tile / TK, tk = tile - tn * TK;
                const float* zp = ldsZ + h * N + tn * 32 + r;
                const float* ap = ldsA + h * (TK * 32) + tk * 32 + r;
#pragma unroll 8
                for (int s = 0; s < GW_ROWS / 2; ++s)
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(zp[2 * s * N], ap[2 * s * (TK * 32)], acc[t], 0, 0, 0);
            }
        }
    }
    float* out = L.partial + (long)wg * N * (KP + 1);
#pragma unroll
    for (int t = 0; t < PER; ++t) {
        const int tile = wave + WAVES * t;
        if (tile < TILES) {
            const int tn = tile / TK, tk = tile - tn * TK;
            const int col = tk * 32 + r;                       // k index
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int row = tn * 32 + acc_row(reg, lane);  // n index
                if (col < KP) out[row * (KP + 1) + col] = acc[t][reg];
            }
        }
    }
    if (tid < N) out[tid * (KP + 1) + KP] = bsum;
}

__global__ __launch_bounds__(512) void mlp_grad_w_kernel(GradWTable T, long nrows)
{
    extern __shared__ __attribute__((aligned(16))) float lds_dyn[];
    const int b = blockIdx.x;
    if (b >= T.l[3].first_block) grad_w_layer<MLP_OUT, MLP_H3>(T.l[3], nrows, b - T.l[3].first_block, lds_dyn);
    else if (b >= T.l[2].first_block) grad_w_layer<MLP_H3, MLP_H2>(T.l[2], nrows, b - T.l[2].first_block, lds_dyn);
    else if (b >= T.l[1].first_block) grad_w_layer<MLP_H2, MLP_H1>(T.l[1], nrows, b - T.l[1].first_block, lds_dyn);
    else grad_w_layer<MLP_H1, MLP_IN_PAD>(T.l[0], nrows, b, lds_dyn);
}

// sum the per-workgroup partials into the packed gradient buffer (layout of P); fixed order.
__global__ __launch_bounds__(256) void mlp_grad_reduce_kernel(GradWTable T, float* __restrict__ G)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= MLP_PACKED_FLOATS) return;
    int layer, off_w, off_b, N, KP;
    if (i < MLP_OFF_W2) { layer = 0; off_w = MLP_OFF_W1; off_b = MLP_OFF_B1; N = MLP_H1; KP = MLP_IN_PAD; }
    else if (i < MLP_OFF_W3) { layer = 1; off_w = MLP_OFF_W2; off_b = MLP_OFF_B2; N = MLP_H2; KP = MLP_H1; }
    else if (i < MLP_OFF_W4) { layer = 2; off_w = MLP_OFF_W3; off_b = MLP_OFF_B3; N = MLP_H3; KP = MLP_H2; }
    else { layer = 3; off_w = MLP_OFF_W4; off_b = MLP_OFF_B4; N = MLP_OUT; KP = MLP_H3; }
    const GradWLayer& L = T.l[layer];
    long idx;
    if (i < off_b) { const int rr = (i - off_w) / KP, cc = (i - off_w) - rr * KP; idx = (long)rr * (KP + 1) + cc; }
    else idx = (long)(i - off_b) * (KP + 1) + KP;
    const long stride = (long)N * (KP + 1);
    float s = 0.0f;
    for (int w = 0; w < L.wgs; ++w) s += L.partial[w * stride + idx];
    G[i] = s;
}


// ---------------------------------------------------------------------------------------------
// Optimizer step (ppo.py:196-199): clip_grad_norm_(max_norm) + Adam (torch defaults: no weight
// decay, no amsgrad) over the packed parameter buffer, plus the refresh of the transposed
// weights the next backward pass streams.  ONE workgroup of 1024 threads: 74 272 elements are
// 73 per thread, the global-norm reduction stays inside the workgroup (fixed order, so the
// step is deterministic), and nothing needs a second launch or a host round trip.  The step
// counter lives in device memory so the launch is graph-capturable.
__global__ __launch_bounds__(1024) void mlp_adam_kernel(float* __restrict__ P, float* __restrict__ PT,
                                                        const float* __restrict__ G, const float* __restrict__ mask,
                                                        float* __restrict__ m, float* __restrict__ v,
                                                        int* __restrict__ step, float lr, float beta1, float beta2,
                                                        float eps, float max_norm, float grad_scale,
                                                        float* __restrict__ norm_out)
{
    __shared__ float red[16];
    __shared__ float s_coef;
    const int tid = threadIdx.x;
    float ss = 0.0f;
    for (int i = tid; i < MLP_PACKED_FLOATS; i += 1024) {
        const float g = G[i] * grad_scale * mask[i];
        ss += g * g;
    }
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_down(ss, o, 64);
    if ((tid & 63) == 0) red[tid >> 6] = ss;
    __syncthreads();
    if (tid == 0) {
        float t = 0.0f;
        for (int w = 0; w < 16; ++w) t += red[w];
        const float norm = sqrtf(t);
        float coef = max_norm / (norm + 1e-6f);            // torch.nn.utils.clip_grad_norm_
        s_coef = coef < 1.0f ? coef : 1.0f;
        if (norm_out) *norm_out = norm;
        *step += 1;
    }
    __syncthreads();
    const float coef = s_coef * grad_scale;
    const int t = *step;
    const float bc1 = 1.0f - powf(beta1, (float)t);
    const float bc2 = 1.0f - powf(beta2, (float)t);
    const float step_size = lr / bc1;
    const float bc2_sqrt = sqrtf(bc2);
    for (int i = tid; i < MLP_PACKED_FLOATS; i += 1024) {
        const float mk = mask[i];
        const float g = G[i] * coef * mk;
        const float mi = beta1 * m[i] + (1.0f - beta1) * g;
        const float vi = beta2 * v[i] + (1.0f - beta2) * g * g;
        m[i] = mi; v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        P[i] = P[i] - mk * (step_size * (mi / denom));
    }
    __syncthreads();
    __threadfence_block();
    // transposes for the dX chain
    for (int i = tid; i < MLP_H2 * MLP_H1; i += 1024) {            // W2 [128][256] -> Wt2 [256][128]
        const int nn = i / MLP_H1, kk = i - nn * MLP_H1;
        PT[MLP_OFF_WT2 + kk * MLP_H2 + nn] = P[MLP_OFF_W2 + i];
    }
    for (int i = tid; i < MLP_H3 * MLP_H2; i += 1024) {            // W3 [128][128] -> Wt3
        const int nn = i / MLP_H2, kk = i - nn * MLP_H2;
        PT[MLP_OFF_WT3 + kk * MLP_H3 + nn] = P[MLP_OFF_W3 + i];
    }
    for (int i = tid; i < MLP_OUT * MLP_H3; i += 1024) {           // W4 [32][128] -> Wt4 [128][32]
        const int nn = i / MLP_H3, kk = i - nn * MLP_H3;
        PT[MLP_OFF_WT4 + kk * MLP_OUT + nn] = P[MLP_OFF_W4 + i];
    }
}

}  // namespace

extern "C" hipError_t flyhip_launch_mlp_forward(const float* P, const float* x, int64_t n, float* mu_out,
                                                float* v_out, float* out_save, float* h1_save, float* h2_save,
                                                float* h3_save, void* stream)
{
    const int grid = (int)((n + BM - 1) / BM);
    hipLaunchKernelGGL(mlp_forward_kernel, dim3(grid), dim3(THREADS), 0, (hipStream_t)stream, P, x, (long)n,
                       mu_out, v_out, out_save, h1_save, h2_save, h3_save);
    return hipGetLastError();
}

extern "C" hipError_t flyhip_launch_mlp_backward_dx(const float* PT, const float* out_saved, const float* h1,
                                                    const float* h2, const float* h3, const float* action,
                                                    const float* old_logp, const float* adv, const float* target,
                                                    const float* var, int64_t n, float inv_batch, float clip,
                                                    float* dz4, float* dz3, float* dz2, float* dz1, float* loss_part,
                                                    void* stream)
{
    const int grid = (int)((n + BM - 1) / BM);
    hipLaunchKernelGGL(mlp_backward_dx_kernel, dim3(grid), dim3(THREADS), 0, (hipStream_t)stream, PT, out_saved,
                       h1, h2, h3, action, old_logp, adv, target, var, (long)n, inv_batch, clip, dz4, dz3, dz2, dz1,
                       loss_part);
    return hipGetLastError();
}

// workgroups per layer, proportional to the layer's share of the dW FLOPs (256 in total)
static const int kGradWgs[4] = {72, 112, 56, 16};

extern "C" int64_t flyhip_mlp_grad_workspace_floats(void)
{
    return (int64_t)kGradWgs[0] * MLP_H1 * (MLP_IN_PAD + 1) + (int64_t)kGradWgs[1] * MLP_H2 * (MLP_H1 + 1) +
           (int64_t)kGradWgs[2] * MLP_H3 * (MLP_H2 + 1) + (int64_t)kGradWgs[3] * MLP_OUT * (MLP_H3 + 1);
}

extern "C" hipError_t flyhip_launch_mlp_grad_w(const float* x, const float* h1, const float* h2, const float* h3,
                                               const float* dz1, const float* dz2, const float* dz3, const float* dz4,
                                               int64_t n, float* workspace, float* grad_out, void* stream)
{
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
        ws += (long)kGradWgs[l] * N[l] * (KP[l] + 1);
        first += kGradWgs[l];
    }
    // dynamic LDS: the largest layer's staging tiles: max over layers of 32*(N + TK*32) floats
    const size_t lds_bytes = sizeof(float) * GW_ROWS * (MLP_H2 + MLP_H1);     // layer 2: 32 x (128 + 256)
    static_assert(GW_ROWS * (MLP_H1 + 96) <= GW_ROWS * (MLP_H2 + MLP_H1), "layer 1 tiles fit");
    hipLaunchKernelGGL(mlp_grad_w_kernel, dim3(first), dim3(512), lds_bytes, (hipStream_t)stream, T, (long)n);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(mlp_grad_reduce_kernel, dim3((MLP_PACKED_FLOATS + 255) / 256), dim3(256), 0,
                       (hipStream_t)stream, T, grad_out);
    return hipGetLastError();
}

extern "C" hipError_t flyhip_launch_mlp_adam(float* P, float* PT, const float* G, const float* mask, float* m,
                                             float* v, int* step, float lr, float beta1, float beta2, float eps,
                                             float max_norm, float grad_scale, float* norm_out, void* stream)
{
    hipLaunchKernelGGL(mlp_adam_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, P, PT, G, mask, m, v, step,
                       lr, beta1, beta2, eps, max_norm, grad_scale, norm_out);
    return hipGetLastError();
}
