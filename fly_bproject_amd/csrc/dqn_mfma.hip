// dqn_mfma.hip — the DQN variant's Q-network (reference UselessFiles/dqn.py:17-29, :64-100; BASELINE
// configs[4]) on the gfx950 matrix cores: 73 -> 256 -> 256 -> 18 with LeakyReLU, fp32 MFMA
// (v_mfma_f32_32x32x2_f32: the reference's fp32 numerics), built from the tile GEMMs of mlp_gemm.inc.
//
//   dqn_act_kernel      dqn.py:89-100   Q-network forward of a 32-env tile + per-env FIRST argmax + eps-greedy mix
//   dqn_forward_kernel  dqn.py:28       Q table of arbitrary rows (tests, diagnostics)
//   dqn_td_kernel       dqn.py:64-79    one sampled replay step (N rows) in ONE launch per tile: target-network
//                                       forward of next_obs -> max_a, online forward of obs (h1/h2 saved),
//                                       TD target, Huber loss, d loss / d Q, and the dX chain down to dZ1
//   dqn_grad_w_kernel   loss.backward() dW / db of the three layers over row slabs (grad_w_layer.inc)
//   dqn_grad_reduce_kernel              fixed-order sum of the partial slabs, ONCE per update (the slabs themselves
//                                       accumulate over the update's sampled steps)
//   dqn_adam_kernel     dqn.py:81-84    Adam (3e-4, torch defaults, NO clipping) on the packed parameters, refresh of
//                                       the fragment-ordered copies, and the soft update of the target network
//                                       (target = target*tau + param*(1-tau)) with its fragment copy
// Everything is row-local except the weight gradient, so no workgroup waits on another.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "flyhip.h"
#include "dqn_layout.h"

namespace {

#include "mlp_gemm.inc"
#include "grad_w_layer.inc"
#include "fs_common.inc"
#include "dqn_fused.inc"
#include "fs_h2.inc"
#include "dqn_fused_h2.inc"

constexpr int DQ_P = DQN_H + 4;                         // LDS pitch of a [32][256] activation tile
constexpr int DQ_TILE = BM * DQ_P;                      // 8320 floats
constexpr int DQ_LB1 = 0, DQ_LB2 = DQN_H, DQ_LB3 = 2 * DQN_H;
constexpr int DQ_BIAS = 2 * DQN_H + DQN_OUT;            // 544
constexpr int DQ_LDS_FLOATS = 2 * DQ_TILE + DQ_BIAS + 2 * BM;   // 68.9 KB: two workgroups per CU
constexpr float LRELU_SLOPE = 0.01f;                    // nn.LeakyReLU() default

__device__ __forceinline__ float lrelu(float x) { return x > 0.0f ? x : x * LRELU_SLOPE; }
// dA * LeakyReLU'(pre-activation), through the saved OUTPUT h (same sign as the pre-activation; 0 -> slope, as torch)
__device__ __forceinline__ float dlrelu(float da, float h) { return h > 0.0f ? da : da * LRELU_SLOPE; }

template <int N, int NT>
__device__ __forceinline__ void epilogue_lrelu(const f32x16 (&acc)[NT], const float* __restrict__ bias, int col0,
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
            y[t][g].x = lrelu(acc[t][4 * g + 0] + bv.x);
            y[t][g].y = lrelu(acc[t][4 * g + 1] + bv.y);
            y[t][g].z = lrelu(acc[t][4 * g + 2] + bv.z);
            y[t][g].w = lrelu(acc[t][4 * g + 3] + bv.w);
            *reinterpret_cast<float4*>(lds_out + r * (N + 4) + nb) = y[t][g];
        }
    }
    if (gtile != nullptr && r < nvalid) {
        const int go = (col0 / 32) * 1024 + lane * 4;             // tile-fragment order (mlp_gemm.inc: frag_off)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) gstore4<false>(gtile, go + t * 1024 + g * 256, y[t][g]);
    }
}

// The Q-network forward of one 32-row tile.  x rows -> arena B, h1 -> arena A, h2 -> arena B, the split-K
// partials of the last layer -> arena A; returns this thread's four Q values: rows (tid>>5) + 8k, column tid&31
// (columns >= 18 are padding and read 0 + 0).
__device__ __forceinline__ void dqn_forward_tile(float* lds, long tile, const float* __restrict__ P, const float* __restrict__ PF,
                                                 const float* __restrict__ x, long n, float* __restrict__ h1_save,
                                                 float* __restrict__ h2_save, float (&qv)[4])
{
    float* ldsA = lds;
    float* ldsB = lds + DQ_TILE;
    float* ldsBias = lds + 2 * DQ_TILE;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long row0 = tile * BM;
    const int nvalid = (int)(n - row0 < BM ? n - row0 : BM);
    const long total = n * DQN_IN, base = row0 * DQN_IN;
    ldsBias[DQ_LB1 + tid] = P[DQN_OFF_B1 + tid];
    ldsBias[DQ_LB2 + tid] = P[DQN_OFF_B2 + tid];
    if (tid < DQN_OUT) ldsBias[DQ_LB3 + tid] = P[DQN_OFF_B3 + tid];
    // the tile's 32 x 73 input block is contiguous and 16-byte aligned: coalesced 16-byte loads, scattered into [32][84]
    constexpr int XV4 = (BM * DQN_IN / 4 + THREADS - 1) / THREADS;
#pragma unroll
    for (int u = 0; u < XV4; ++u) {
        const int i4 = tid + u * THREADS;
        if (i4 < BM * DQN_IN / 4) {
            const long f = base + 4L * i4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (f + 3 < total) v = *reinterpret_cast<const float4*>(x + f);
            else {
                if (f < total) v.x = x[f];
                if (f + 1 < total) v.y = x[f + 1];
                if (f + 2 < total) v.z = x[f + 2];
            }
            const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int ff = 4 * i4 + j, rr = ff / DQN_IN, cc = ff - rr * DQN_IN;
                ldsB[rr * (DQN_IN_PAD + 4) + cc] = e[j];
            }
        }
    }
    if (tid < BM * (DQN_IN_PAD - DQN_IN)) {
        const int rr = tid / (DQN_IN_PAD - DQN_IN), cc = DQN_IN + tid - rr * (DQN_IN_PAD - DQN_IN);
        ldsB[rr * (DQN_IN_PAD + 4) + cc] = 0.0f;
    }
    __syncthreads();
    {   // L1: 80 -> 256, a wave owns 64 columns
        f32x16 acc[2];
        tile_gemm<DQN_IN_PAD, 2>(PF + DQN_OFF_F1, 2 * wave, ldsB, acc, lane);
        epilogue_lrelu<DQN_H, 2>(acc, ldsBias + DQ_LB1, wave * 64, ldsA, lane, h1_save ? h1_save + row0 * DQN_H : nullptr, nvalid);
    }
    __syncthreads();
    {   // L2: 256 -> 256 (x in arena B is dead: every wave passed the barrier)
        f32x16 acc[2];
        tile_gemm<DQN_H, 2>(PF + DQN_OFF_F2, 2 * wave, ldsA, acc, lane);
        epilogue_lrelu<DQN_H, 2>(acc, ldsBias + DQ_LB2, wave * 64, ldsB, lane, h2_save ? h2_save + row0 * DQN_H : nullptr, nvalid);
    }
    __syncthreads();
    {   // L3: 256 -> 32 (18 real), split-K over the four waves (64 k each), partials through arena A
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
        const int r = lane & 31, h = lane >> 5;
        const float* ap = ldsB + r * DQ_P + wave * 64 + h * 32;
        const float* bp = PF + DQN_OFF_F3 + (wave * 8) * 256 + lane * 4;
        float4 w[8];
#pragma unroll
        for (int kq = 0; kq < 8; ++kq) w[kq] = *reinterpret_cast<const float4*>(bp + 256 * kq);
#pragma unroll
        for (int kq = 0; kq < 8; ++kq) {
            const float4 a = *reinterpret_cast<const float4*>(ap + 4 * kq);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[kq].x, a.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[kq].y, a.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[kq].z, a.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[kq].w, a.w, acc, 0, 0, 0);
        }
        float* part = ldsA + wave * (BM * DQN_OUT);                              // [row][32]
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *reinterpret_cast<float4*>(part + r * DQN_OUT + acc_n(g, lane)) =
                make_float4(acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = tid + k * THREADS, col = i & 31;
        const float z = ((ldsA[i] + ldsA[BM * DQN_OUT + i]) + ldsA[2 * BM * DQN_OUT + i]) + ldsA[3 * BM * DQN_OUT + i];
        qv[k] = z + ldsBias[DQ_LB3 + col];
    }
    __syncthreads();                                                             // arenas free for the caller
}

__global__ __launch_bounds__(THREADS, 2) void dqn_forward_kernel(const float* __restrict__ P, const float* __restrict__ PF,
                                                                  const float* __restrict__ x, long n, float* __restrict__ q_out)
{
    __shared__ __attribute__((aligned(16))) float lds[DQ_LDS_FLOATS];
    float qv[4];
    dqn_forward_tile(lds, blockIdx.x, P, PF, x, n, nullptr, nullptr, qv);
    const long row0 = (long)blockIdx.x * BM;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int row = (threadIdx.x >> 5) + 8 * k, col = threadIdx.x & 31;
        if (row0 + row < n && col < DQN_NACT) q_out[(row0 + row) * DQN_NACT + col] = qv[k];
    }
}

// dqn.py:89-100.  true_act = FIRST index of the row maximum / 17; act = coin < eps ? rand : true_act; 2*(act-0.5).
__global__ __launch_bounds__(THREADS, 2) void dqn_act_kernel(const float* __restrict__ P, const float* __restrict__ PF,
                                                              const float* __restrict__ x, long n,
                                                              const float* __restrict__ coin_u, const float* __restrict__ rand_u,
                                                              float epsilon, float* __restrict__ act_out, float* __restrict__ q_out)
{
    __shared__ __attribute__((aligned(16))) float lds[DQ_LDS_FLOATS];
    float qv[4];
    dqn_forward_tile(lds, blockIdx.x, P, PF, x, n, nullptr, nullptr, qv);
    const long row0 = (long)blockIdx.x * BM;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int row = (threadIdx.x >> 5) + 8 * k, col = threadIdx.x & 31;
        const long e = row0 + row;
        if (q_out && e < n && col < DQN_NACT) q_out[e * DQN_NACT + col] = qv[k];
        float best = col < DQN_NACT ? qv[k] : -INFINITY;
        int idx = col;
#pragma unroll
        for (int o = 1; o < 32; o <<= 1) {                     // (max, lowest index) over the 32 lanes of the row
            const float ob = __shfl_xor(best, o, 32);
            const int oi = __shfl_xor(idx, o, 32);
            if (ob > best || (ob == best && oi < idx)) { best = ob; idx = oi; }
        }
        if (col == 0 && e < n) {
            const float true_act = (float)idx / (float)(DQN_NACT - 1);
            const float a = (coin_u[e] < epsilon) ? rand_u[e] : true_act;
            act_out[e] = 2.0f * (a - 0.5f);
        }
    }
}

// dZ = dA * LeakyReLU'(h): h from the saved tile (sc1 loads: the rows were stored by THIS workgroup moments ago;
// bypassing the CU's L1 makes the read independent of what that L1 holds), dZ to HBM and optionally to LDS
template <int NT>
__device__ __forceinline__ void epilogue_dlrelu(const f32x16 (&acc)[NT], const float* __restrict__ htile, int col0, float* lds_out,
                                                float* __restrict__ gtile, int nvalid, int lane)
{
    const int r = lane & 31;
    const int go = (col0 / 32) * 1024 + lane * 4;
    float4 z[NT][4];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 hv = gload4<true>(htile, go + t * 1024 + g * 256);
            z[t][g].x = dlrelu(acc[t][4 * g + 0], hv.x);
            z[t][g].y = dlrelu(acc[t][4 * g + 1], hv.y);
            z[t][g].z = dlrelu(acc[t][4 * g + 2], hv.z);
            z[t][g].w = dlrelu(acc[t][4 * g + 3], hv.w);
            if (lds_out) *reinterpret_cast<float4*>(lds_out + r * DQ_P + col0 + 32 * t + acc_n(g, lane)) = z[t][g];
        }
    if (r < nvalid) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) gstore4<false>(gtile, go + t * 1024 + g * 256, z[t][g]);
    }
}

// One sampled replay step (dqn.py:64-79), 32 rows per workgroup.  obs / next_obs f32 [n][73] (rows of the replay
// ring: no gather), act / reward / done f32 [n].  Leaves h1, h2, dz1, dz2 [n][256] and dz3 [n][32] in tile-fragment
// order for dqn_grad_w, and loss_part[tile] = sum of the tile's Huber terms.  inv_B = 1 / (rows of the WHOLE batch).
__global__ __launch_bounds__(THREADS, 2) void dqn_td_kernel(
    const float* __restrict__ P, const float* __restrict__ PF, const float* __restrict__ PT,
    const float* __restrict__ P_tgt, const float* __restrict__ PF_tgt,
    const float* __restrict__ obs, const float* __restrict__ next_obs, const float* __restrict__ act,
    const float* __restrict__ reward, const float* __restrict__ done, long n, float discount, float inv_B,
    float* __restrict__ h1_save, float* __restrict__ h2_save, float* __restrict__ dz3, float* __restrict__ dz2,
    float* __restrict__ dz1, float* __restrict__ loss_part)
{
    __shared__ __attribute__((aligned(16))) float lds[DQ_LDS_FLOATS];
    float* ldsA = lds;
    float* ldsB = lds + DQ_TILE;
    float* rowloss = lds + 2 * DQ_TILE + DQ_BIAS;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long tile = blockIdx.x, row0 = tile * BM;
    const int nvalid = (int)(n - row0 < BM ? n - row0 : BM);
    const int col = tid & 31;

    float qv[4], qn_max[4];
    dqn_forward_tile(lds, tile, P_tgt, PF_tgt, next_obs, n, nullptr, nullptr, qv);       // q_target(next_obs), dqn.py:73-74
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float m = col < DQN_NACT ? qv[k] : -INFINITY;
#pragma unroll
        for (int o = 1; o < 32; o <<= 1) m = fmaxf(m, __shfl_xor(m, o, 32));
        qn_max[k] = m;
    }
    dqn_forward_tile(lds, tile, P, PF, obs, n, h1_save, h2_save, qv);                    // q(obs), dqn.py:68
    float* ldsZ3 = ldsB;                                                                 // [32][36]
    float* dz3_t = dz3 + row0 * DQN_OUT;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int row = (tid >> 5) + 8 * k;
        const bool in = row < nvalid;
        float d = 0.0f, hub = 0.0f;
        int idx = 0;
        if (in) {
            const float a01 = 0.5f * (act[row0 + row] + 1.0f);
            idx = (int)rintf(a01 * (float)(DQN_NACT - 1));                                // torch.round: half to even (dqn.py:70)
            idx = idx < 0 ? 0 : (idx >= DQN_NACT ? DQN_NACT - 1 : idx);
        }
        const float q_val = __shfl(qv[k], idx, 32);                                       // q_table[b, act] (dqn.py:71)
        if (in) {
            const float target = reward[row0 + row] + discount * qn_max[k] * done[row0 + row];
            const float dv = q_val - target;
            hub = fabsf(dv) < 1.0f ? 0.5f * dv * dv : fabsf(dv) - 0.5f;                    // smooth_l1, beta = 1
            if (col == idx) d = inv_B * fminf(fmaxf(dv, -1.0f), 1.0f);
            dz3_t[frag_off(row, col)] = d;
        }
        ldsZ3[row * (DQN_OUT + 4) + col] = d;
        if (col == 0) rowloss[row] = hub;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's h1 / h2 stores are acknowledged before anyone re-reads them
    __syncthreads();
    if (tid < 32) {
        float hub = rowloss[tid];
        for (int o = 16; o > 0; o >>= 1) hub += __shfl_down(hub, o, 32);
        if (tid == 0) loss_part[tile] = hub;
    }
    {   // dA2 = dZ3 . W3 -> dZ2 (arena A; the split-K partials there are consumed)
        f32x16 acc[2];
        tile_gemm<DQN_OUT, 2>(PT + DQN_OFF_T3, 2 * wave, ldsZ3, acc, lane);
        epilogue_dlrelu<2>(acc, h2_save + row0 * DQN_H, wave * 64, ldsA, dz2 + row0 * DQN_H, nvalid, lane);
    }
    __syncthreads();
    {   // dA1 = dZ2 . W2 -> dZ1 (HBM only)
        f32x16 acc[2];
        tile_gemm<DQN_H, 2>(PT + DQN_OFF_T2, 2 * wave, ldsA, acc, lane);
        epilogue_dlrelu<2>(acc, h1_save + row0 * DQN_H, wave * 64, nullptr, dz1 + row0 * DQN_H, nvalid, lane);
    }
}

// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(GW_THREADS) void dqn_grad_w_kernel(GradWTable T, long nrows)
{
    extern __shared__ __attribute__((aligned(16))) float lds_dyn[];
    const int b = blockIdx.x;
    if (b >= T.l[2].first_block)
        grad_w_layer<DQN_OUT, DQN_H, DQN_H, DQN_H, 1, 8, 1, 1>(T.l[2], nrows, b - T.l[2].first_block, lds_dyn);
    else if (b >= T.l[1].first_block)
        grad_w_layer<DQN_H, DQN_H, DQN_H, DQN_H, 4, 4, 2, 2>(T.l[1], nrows, b - T.l[1].first_block, lds_dyn);
    else
        grad_w_layer<DQN_H, DQN_IN, 96, DQN_IN_PAD, 4, 3, 2, 1>(T.l[0], nrows, b, lds_dyn);
}

constexpr int DQ_RED_WAVES = 16;
constexpr int DQ_RED_BLOCKS = (DQN_PACKED_FLOATS / 4 + 63) / 64;        // 371

// G (+)= sum over the partial slabs (fixed order).  `accumulate` adds to G: the gradient of one DQN update is the sum
// over its sampled replay steps, each reduced by its own launch.
__global__ __launch_bounds__(64 * DQ_RED_WAVES) void dqn_grad_reduce_kernel(GradWTable T, float* __restrict__ G, int accumulate)
{
    __shared__ float4 red[DQ_RED_WAVES][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int o = 4 * (blockIdx.x * 64 + lane);
    float4 total = make_float4(0.f, 0.f, 0.f, 0.f);
    if (o < DQN_PACKED_FLOATS) {
        int layer, off_w, N, KP;
        if (o < DQN_OFF_W2) { layer = 0; off_w = DQN_OFF_W1; N = DQN_H; KP = DQN_IN_PAD; }
        else if (o < DQN_OFF_W3) { layer = 1; off_w = DQN_OFF_W2; N = DQN_H; KP = DQN_H; }
        else { layer = 2; off_w = DQN_OFF_W3; N = DQN_OUT; KP = DQN_H; }
        const float* part = layer == 0 ? T.l[0].partial : layer == 1 ? T.l[1].partial : T.l[2].partial;
        const int wgs = layer == 0 ? T.l[0].wgs : layer == 1 ? T.l[1].wgs : T.l[2].wgs;
        const long s4 = ((long)N * KP + N) / 4;
        const float4* p4 = reinterpret_cast<const float4*>(part + (o - off_w));
        for (int w = wave; w < wgs; w += DQ_RED_WAVES) {
            const float4 v = p4[(long)w * s4];
            total.x += v.x; total.y += v.y; total.z += v.z; total.w += v.w;
        }
    }
    red[wave][lane] = total;
    __syncthreads();
    if (wave == 0 && o < DQN_PACKED_FLOATS) {
        float4 g = accumulate ? *reinterpret_cast<const float4*>(G + o) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int w = 0; w < DQ_RED_WAVES; ++w) { g.x += red[w][lane].x; g.y += red[w][lane].y; g.z += red[w][lane].z; g.w += red[w][lane].w; }
        *reinterpret_cast<float4*>(G + o) = g;
    }
}

constexpr int DQ_ADAM_THREADS = 1024;
constexpr int DQ_ADAM_BLOCKS = (DQN_PACKED_FLOATS + DQ_ADAM_THREADS - 1) / DQ_ADAM_THREADS;   // 93

// dqn.py:81-84: optimizer.step() (Adam, torch defaults) then soft_update(q, q_target, tau) (dqn.py:33-36).
// Block 0 advances the device step counter AFTER every block has read it: each block reads *step + 1.
__global__ __launch_bounds__(DQ_ADAM_THREADS) void dqn_adam_kernel(
    float* __restrict__ P, float* __restrict__ PF, float* __restrict__ PT, float* __restrict__ P_tgt, float* __restrict__ PF_tgt,
    const int* __restrict__ idx_f, const int* __restrict__ idx_t, const float* __restrict__ G, const float* __restrict__ mask,
    float* __restrict__ m, float* __restrict__ v, const int* __restrict__ step, float lr, float beta1, float beta2, float eps,
    float tau, u16* __restrict__ QB, u16* __restrict__ QTB, u16* __restrict__ QB_tgt, const int* __restrict__ idx_fb,
    const int* __restrict__ idx_tb, const int* __restrict__ grad_invalid)
{
    __shared__ float s_step_size, s_bc2_sqrt;
    // fail closed: `grad_invalid` (optional) is the device word dqn_fused_update_h2 sets when a value of the update did not fit fp16 --
    // the gradient is then not a gradient: nothing moves, the step counter stays (every block sees the same word: it is only ever
    // set by the gradient's launches, which are done)
    if (grad_invalid && *grad_invalid != 0) return;
    const int tid = threadIdx.x, i = blockIdx.x * DQ_ADAM_THREADS + tid;
    if (tid == 0) {
        const float ts = (float)(*step + 1);
        s_step_size = lr / (1.0f - powf(beta1, ts));
        s_bc2_sqrt = sqrtf(1.0f - powf(beta2, ts));
    }
    __syncthreads();
    if (i >= DQN_PACKED_FLOATS) return;
    const float mk = mask[i];
    const float g = G[i] * mk;
    const float mi = beta1 * m[i] + (1.0f - beta1) * g;
    const float vi = beta2 * v[i] + (1.0f - beta2) * g * g;
    m[i] = mi; v[i] = vi;
    const float denom = sqrtf(vi) / s_bc2_sqrt + eps;
    const float p = P[i] - mk * (s_step_size * (mi / denom));
    P[i] = p;
    const float pt = P_tgt[i] * tau + p * (1.0f - tau);
    P_tgt[i] = pt;
    const int jf = idx_f[i], jt = idx_t[i];
    if (jf >= 0) { PF[jf] = p; PF_tgt[jf] = pt; }
    if (jt >= 0) PT[jt] = p;
    if (QB) {       // the three-term bf16 planes of the fused update (dqn_fused.inc): online forward / transposed, target forward
        u16 a, b, c;
        const int kf = idx_fb[i], kt = idx_tb[i];
        split3(p, a, b, c);
        if (kf >= 0) { QB[kf] = a; QB[kf + 512] = b; QB[kf + 1024] = c; }
        if (kt >= 0) { QTB[kt] = a; QTB[kt + 512] = b; QTB[kt + 1024] = c; }
        split3(pt, a, b, c);
        if (kf >= 0) { QB_tgt[kf] = a; QB_tgt[kf + 512] = b; QB_tgt[kf + 1024] = c; }
    }
}

__global__ void dqn_step_inc_kernel(int* step, const int* grad_invalid) { if (!(grad_invalid && *grad_invalid != 0)) *step += 1; }

}  // namespace

extern "C" hipError_t flyhip_launch_dqn_forward(const float* P, const float* PF, const float* x, int64_t n, float* q_out,
                                                void* stream)
{
    hipLaunchKernelGGL(dqn_forward_kernel, dim3((unsigned)((n + BM - 1) / BM)), dim3(THREADS), 0, (hipStream_t)stream, P, PF, x,
                       (long)n, q_out);
    return hipGetLastError();
}

extern "C" hipError_t flyhip_launch_dqn_act(const float* P, const float* PF, const float* x, int64_t n, const float* coin_u,
                                            const float* rand_u, float epsilon, float* act_out, float* q_out, void* stream)
{
    hipLaunchKernelGGL(dqn_act_kernel, dim3((unsigned)((n + BM - 1) / BM)), dim3(THREADS), 0, (hipStream_t)stream, P, PF, x,
                       (long)n, coin_u, rand_u, epsilon, act_out, q_out);
    return hipGetLastError();
}

extern "C" hipError_t flyhip_launch_dqn_td(const float* P, const float* PF, const float* PT, const float* P_tgt,
                                           const float* PF_tgt, const float* obs, const float* next_obs, const float* act,
                                           const float* reward, const float* done, int64_t n, float discount, float inv_B,
                                           float* h1, float* h2, float* dz3, float* dz2, float* dz1, float* loss_part,
                                           void* stream)
{
    hipLaunchKernelGGL(dqn_td_kernel, dim3((unsigned)((n + BM - 1) / BM)), dim3(THREADS), 0, (hipStream_t)stream, P, PF, PT,
                       P_tgt, PF_tgt, obs, next_obs, act, reward, done, (long)n, discount, inv_B, h1, h2, dz3, dz2, dz1,
                       loss_part);
    return hipGetLastError();
}

// workgroups per layer ~ the layer's share of the dW FLOPs (256 in total: one per CU)
static const int kDqnGradWgs[3] = {52, 180, 24};

extern "C" int64_t flyhip_dqn_grad_workspace_floats(void)
{
    return (int64_t)kDqnGradWgs[0] * DQN_H * (DQN_IN_PAD + 1) + (int64_t)kDqnGradWgs[1] * DQN_H * (DQN_H + 1) +
           (int64_t)kDqnGradWgs[2] * DQN_OUT * (DQN_H + 1);
}

// accumulate: bit 0 = add this step's dW to the partial slabs (steps 2.. of an update), bit 1 = this is the LAST step:
// reduce the slabs into `grad` now (one reduction per update, not per sampled step)
extern "C" hipError_t flyhip_launch_dqn_grad_w(const float* x, const float* h1, const float* h2, const float* dz1,
                                               const float* dz2, const float* dz3, int64_t n, float* workspace, float* grad,
                                               int accumulate, void* stream)
{
    GradWTable T;
    const float* dz[3] = {dz1, dz2, dz3};
    const float* a[3] = {x, h1, h2};
    const int N[3] = {DQN_H, DQN_H, DQN_OUT};
    const int Ka[3] = {DQN_IN, DQN_H, DQN_H};
    const int KP[3] = {DQN_IN_PAD, DQN_H, DQN_H};
    float* ws = workspace;
    int first = 0;
    for (int l = 0; l < 3; ++l) {
        T.l[l].dz = dz[l]; T.l[l].a = a[l]; T.l[l].partial = ws;
        T.l[l].N = N[l]; T.l[l].Ka = Ka[l]; T.l[l].KP = KP[l]; T.l[l].wgs = kDqnGradWgs[l]; T.l[l].first_block = first;
        T.l[l].accumulate = accumulate & 1; T.l[l].chunked = 0;
        ws += (long)kDqnGradWgs[l] * ((long)N[l] * KP[l] + N[l]);
        first += kDqnGradWgs[l];
    }
    T.l[3] = T.l[2];
    // dynamic LDS: two buffers of the largest layer's chunk: 2 x 32 x (260 + 260) floats = 130 KiB
    const size_t lds_bytes = sizeof(float) * 2 * GW_ROWS * (DQN_H + GW_PAD + DQN_H + GW_PAD);
    {       // (per launch: the attribute belongs to the CURRENT device)
        hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void*>(dqn_grad_w_kernel),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (ea != hipSuccess) return ea;
    }
    hipLaunchKernelGGL(dqn_grad_w_kernel, dim3(first), dim3(GW_THREADS), lds_bytes, (hipStream_t)stream, T, (long)n);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || !(accumulate & 2)) return e;
    hipLaunchKernelGGL(dqn_grad_reduce_kernel, dim3(DQ_RED_BLOCKS), dim3(64 * DQ_RED_WAVES), 0, (hipStream_t)stream, T, grad, 0);
    return hipGetLastError();
}

extern "C" hipError_t flyhip_launch_dqn_adam(float* P, float* PF, float* PT, float* P_tgt, float* PF_tgt, const int* idx_f,
                                             const int* idx_t, const float* G, const float* mask, float* m, float* v,
                                             int* step, float lr, float beta1, float beta2, float eps, float tau,
                                             uint16_t* QB, uint16_t* QTB, uint16_t* QB_tgt, const int* idx_fb, const int* idx_tb,
                                             const int* grad_invalid, void* stream)
{
    hipLaunchKernelGGL(dqn_adam_kernel, dim3(DQ_ADAM_BLOCKS), dim3(DQ_ADAM_THREADS), 0, (hipStream_t)stream, P, PF, PT, P_tgt,
                       PF_tgt, idx_f, idx_t, G, mask, m, v, step, lr, beta1, beta2, eps, tau, QB, QTB, QB_tgt, idx_fb, idx_tb,
                       grad_invalid);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(dqn_step_inc_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, step, grad_invalid);
    return hipGetLastError();
}

// ---- the fused update: dqn_chain_kernel + dqn_dw2_kernel + the fixed-order reduction of their per-workgroup slabs -----------------
static int dqn_cus()
{
    static int cus[16] = {0};        // per DEVICE: a process may drive more than one
    int dev = 0;
    hipDeviceProp_t pr;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 256;
    if (!cus[dev]) cus[dev] = hipGetDeviceProperties(&pr, dev) == hipSuccess ? pr.multiProcessorCount : 256;
    return cus[dev];
}

static unsigned long long* g_dqn_stamps = nullptr;       // diagnostics (tools/stamp_dqn.py): u64 [workgroups][64], normally null
extern "C" void flyhip_debug_set_dqn_stamps(unsigned long long* p) { g_dqn_stamps = p; }
// measurement only (bench.py times the launches one by one): bit 0 chain kernel, bit 1 dW2 kernel, bit 2 slab reduction; 7 = the update
static int g_dqn_phases = 7;
extern "C" void flyhip_debug_set_dqn_fused_phases(int mask) { g_dqn_phases = mask & 7; }
extern "C" int64_t flyhip_dqn_fused_workspace_floats(void) { return (int64_t)dqn_cus() * DQN_PACKED_FLOATS; }
extern "C" int64_t flyhip_dqn_fused_image_halves(int64_t rows) { return (rows / BM) * 2 * (int64_t)DF_IMAGE_HALVES; }

extern "C" hipError_t flyhip_launch_dqn_fused_update(const float* P, const uint16_t* QB, const uint16_t* QTB, const float* P_tgt,
                                                     const uint16_t* QB_tgt, const void* chunks, int S, int64_t n, float discount,
                                                     float inv_B, uint16_t* images, float* workspace, float* grad, float* loss_part,
                                                     int rows_aligned16, void* stream)
{
    const long tiles_per = n / BM, ntiles = (long)S * tiles_per;
    if (ntiles <= 0 || ntiles + 4096 >= (1L << 31)) return hipErrorInvalidValue;          // the kernels count tiles in 32 bits
    const int cus = dqn_cus();
    const int grid = (int)(ntiles < cus ? ntiles : cus);
    {       // (per launch: the attribute belongs to the CURRENT device)
        hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void*>(dqn_chain_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                            DF_LDS_BYTES);
        if (ea != hipSuccess) return ea;
        ea = hipFuncSetAttribute(reinterpret_cast<const void*>(dqn_dw2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, DW2_LDS_BYTES);
        if (ea != hipSuccess) return ea;
    }
    float* ws1 = workspace;
    float* ws2 = ws1 + (long)cus * DF_STRIDE1;
    float* ws3 = ws2 + (long)cus * DF_STRIDE2;
    hipError_t e = hipSuccess;
    if (g_dqn_phases & 1) {
        hipLaunchKernelGGL(dqn_chain_kernel, dim3(grid), dim3(THREADS), DF_LDS_BYTES, (hipStream_t)stream, P, QB, QTB, P_tgt, QB_tgt,
                           static_cast<const DqnChunk*>(chunks), S, tiles_per, discount, inv_B, images, ws1, ws2, ws3, loss_part,
                           g_dqn_stamps, rows_aligned16);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    if (g_dqn_phases & 2) {
        hipLaunchKernelGGL(dqn_dw2_kernel, dim3(grid), dim3(THREADS), DW2_LDS_BYTES, (hipStream_t)stream, images, ntiles, ws2);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    if (!(g_dqn_phases & 4)) return hipSuccess;
    GradWTable T;
    float* part[3] = {ws1, ws2, ws3};
    const int N[3] = {DQN_H, DQN_H, DQN_OUT};
    const int KP[3] = {DQN_IN_PAD, DQN_H, DQN_H};
    for (int l = 0; l < 3; ++l) {
        T.l[l].dz = nullptr; T.l[l].a = nullptr; T.l[l].partial = part[l];
        T.l[l].N = N[l]; T.l[l].Ka = KP[l]; T.l[l].KP = KP[l]; T.l[l].wgs = grid; T.l[l].first_block = 0; T.l[l].accumulate = 0;
        T.l[l].chunked = 0;
    }
    T.l[3] = T.l[2];
    hipLaunchKernelGGL(dqn_grad_reduce_kernel, dim3(DQ_RED_BLOCKS), dim3(64 * DQ_RED_WAVES), 0, (hipStream_t)stream, T, grad, 0);
    return hipGetLastError();
}

// ---- the same update in the fp16x2 arithmetic (dqn_fused_h2.inc): weight planes + scales, chain, dW2, slab reduction, next scales -----
extern "C" int64_t flyhip_dqn_fused_h2_workspace_floats(void) { return (int64_t)dqn_cus() * (DQN_PACKED_FLOATS + H2_NACT_CLASSES); }
extern "C" int64_t flyhip_dqn_fused_h2_image_halves(int64_t rows) { return (rows / BM) * 2 * (int64_t)DH_IMAGE_HALVES; }

// flags: bit 0 = leave the lagged scales as they are (tests: run-to-run comparisons), bit 1 = calibration pass (no dW2, no
// reduction: only the class maxima -> scales; `grad` is not written), bit 2 = dZ2's image stays home: the chain kernel leaves a
// 2304-byte record per tile in its place and dqn_dw2r_h2_kernel rebuilds dZ2 from it (dqn_fused_h2.inc)
extern "C" hipError_t flyhip_launch_dqn_fused_update_h2(const float* P, uint16_t* QH, uint16_t* QTH, const float* P_tgt, uint16_t* QH_tgt,
                                                        const int* idx_fb, const int* idx_tb, float* fsc, int* ovf, const void* chunks,
                                                        int S, int64_t n, float discount, float inv_B, uint16_t* images, float* workspace,
                                                        float* grad, float* loss_part, int rows_aligned16, int flags, void* stream)
{
    const long tiles_per = n / BM, ntiles = (long)S * tiles_per;
    if (ntiles <= 0 || ntiles + 4096 >= (1L << 31)) return hipErrorInvalidValue;
    const int cus = dqn_cus();
    const int grid = (int)(ntiles < cus ? ntiles : cus);
    {
        hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void*>(dqn_chain_h2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                            DH_LDS_BYTES);
        if (ea != hipSuccess) return ea;
        ea = hipFuncSetAttribute(reinterpret_cast<const void*>(dqn_dw2_h2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, DW2H_LDS_BYTES);
        if (ea != hipSuccess) return ea;
        ea = hipFuncSetAttribute(reinterpret_cast<const void*>(dqn_dw2r_h2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, DW2R_LDS_BYTES);
        if (ea != hipSuccess) return ea;
    }
    float* ws1 = workspace;
    float* ws2 = ws1 + (long)cus * DF_STRIDE1;
    float* ws3 = ws2 + (long)cus * DF_STRIDE2;
    float* wsmax = ws3 + (long)cus * DF_STRIDE3;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(dqn_h2_planes_kernel, dim3(DH_PL_BLOCKS), dim3(DH_PL_THREADS), 0, st, P, P_tgt, idx_fb, idx_tb, QH, QTH, QH_tgt, fsc);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (g_dqn_phases & 1) {
        hipLaunchKernelGGL(dqn_chain_h2_kernel, dim3(grid), dim3(THREADS), DH_LDS_BYTES, st, P, QH, QTH, P_tgt, QH_tgt, fsc,
                           static_cast<const DqnChunk*>(chunks), S, tiles_per, discount, inv_B, images, ws1, ws2, ws3, wsmax, loss_part,
                           g_dqn_stamps, rows_aligned16, (flags & 4) ? 1 : 0);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    if (!(flags & 2)) {
        if (g_dqn_phases & 2) {
            if (flags & 4)
                hipLaunchKernelGGL(dqn_dw2r_h2_kernel, dim3(grid), dim3(THREADS), DW2R_LDS_BYTES, st, images, ntiles, P, fsc, ws2);
            else
                hipLaunchKernelGGL(dqn_dw2_h2_kernel, dim3(grid), dim3(THREADS), DW2H_LDS_BYTES, st, images, ntiles, fsc, ws2);
            e = hipGetLastError();
            if (e != hipSuccess) return e;
        }
        if (g_dqn_phases & 4) {
            GradWTable T;
            float* part[3] = {ws1, ws2, ws3};
            const int N[3] = {DQN_H, DQN_H, DQN_OUT};
            const int KP[3] = {DQN_IN_PAD, DQN_H, DQN_H};
            for (int l = 0; l < 3; ++l) {
                T.l[l].dz = nullptr; T.l[l].a = nullptr; T.l[l].partial = part[l];
                T.l[l].N = N[l]; T.l[l].Ka = KP[l]; T.l[l].KP = KP[l]; T.l[l].wgs = grid; T.l[l].first_block = 0; T.l[l].accumulate = 0;
                T.l[l].chunked = 0;
            }
            T.l[3] = T.l[2];
            hipLaunchKernelGGL(dqn_grad_reduce_kernel, dim3(DQ_RED_BLOCKS), dim3(64 * DQ_RED_WAVES), 0, st, T, grad, 0);
            e = hipGetLastError();
            if (e != hipSuccess) return e;
        }
    }
    if (g_dqn_phases & 1) {
        hipLaunchKernelGGL(dqn_h2_scales_kernel, dim3(1), dim3(64 * DH_NMAX), 0, st, wsmax, grid, fsc, ovf, flags & 1);
        e = hipGetLastError();
    }
    return e;
}
