// ppo_kernels.hip — gfx950 kernels for the PPO rollout math around the policy network.
//
//   ppo_sample_logprob_kernel   ppo.py:213-220  a = mu + sqrt(var) eps; log N(a; mu, diag var); clip
//   ppo_td_gae_kernel           ppo.py:157-171  TD target + delta + reverse GAE recurrence
//
// Both are streaming kernels (HBM-bound by construction): rows of the row-major [n][18] operands
// are staged through LDS so that global traffic is 16-byte coalesced while each lane walks its
// own row; the [T][N] rollout tensors are walked with lane = env, so every time step is one
// coalesced 256-byte line per wave and the recurrence lives in a register.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "flyhip.h"

namespace {

constexpr int NA = FLY_NUM_DOF;
constexpr int SL_BLOCK = 64;            // rows per workgroup (one wave)
constexpr int SL_PAD = NA + 1;          // 19-word row pitch: conflict-free column walks

__global__ __launch_bounds__(SL_BLOCK) void ppo_sample_logprob_kernel(
    const float* __restrict__ mu, const float* __restrict__ var, const float* __restrict__ eps,
    float* __restrict__ act_out, float* __restrict__ logp_out, long n)
{
    __shared__ float s_mu[SL_BLOCK * SL_PAD];
    __shared__ float s_eps[SL_BLOCK * SL_PAD];
    const int tid = threadIdx.x;
    const long row0 = (long)blockIdx.x * SL_BLOCK;
    const long rows = (n - row0) < SL_BLOCK ? (n - row0) : SL_BLOCK;
    const long base = row0 * NA;
    const int count = (int)rows * NA;
    for (int i = tid; i < count; i += SL_BLOCK) {
        int rr = i / NA, cc = i - rr * NA;
        s_mu[rr * SL_PAD + cc] = mu[base + i];
        s_eps[rr * SL_PAD + cc] = eps[base + i];
    }
    __syncthreads();
    // scale_tril = cholesky(diag(var)) = diag(sqrt(var)); half_log_det = sum log L_jj (ppo.py:215-217)
    float L[NA];
    float half_log_det = 0.0f;
#pragma unroll
    for (int j = 0; j < NA; ++j) { L[j] = sqrtf(var[j]); half_log_det = __fadd_rn(half_log_det, logf(L[j])); }
    const float klog2pi = 33.08178959434617f;   // 18 * log(2*pi)
    if (tid < rows) {
        float M = 0.0f;
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            float m = s_mu[tid * SL_PAD + j];
            float a = __fadd_rn(m, __fmul_rn(L[j], s_eps[tid * SL_PAD + j]));   // rsample
            float x = __fsub_rn(a, m) / L[j];                                    // mahalanobis
            M = __fadd_rn(M, __fmul_rn(x, x));
            s_mu[tid * SL_PAD + j] = fminf(fmaxf(a, -1.0f), 1.0f);               // ppo.py:220
        }
        logp_out[row0 + tid] = __fsub_rn(__fmul_rn(-0.5f, __fadd_rn(klog2pi, M)), half_log_det);
    }
    __syncthreads();
    for (int i = tid; i < count; i += SL_BLOCK) {
        int rr = i / NA, cc = i - rr * NA;
        act_out[base + i] = s_mu[rr * SL_PAD + cc];
    }
}

// lane = env; t walks backwards; loads of step t-1.. are independent of the carried advantage,
// so the unrolled body keeps several time steps of loads in flight.
template <int MODE>
__global__ __launch_bounds__(256) void ppo_td_gae_kernel(
    const float* __restrict__ reward, const float* __restrict__ v, const float* __restrict__ v_next,
    const float* __restrict__ done, float gamma, float gl, long T, long N,
    float* __restrict__ target_out, float* __restrict__ adv_out)
{
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= N) return;
    float d_row = (MODE & PPO_GAE_DONE_PER_STEP) ? 0.0f : done[e];
    float a = 0.0f;
#pragma unroll 8
    for (long t = T - 1; t >= 0; --t) {
        const long i = t * N + e;
        float d = (MODE & PPO_GAE_DONE_PER_STEP) ? done[i] : d_row;
        float tg = __fadd_rn(reward[i], __fmul_rn(__fmul_rn(gamma, v_next[i]), d));   // ppo.py:160
        float delta = __fsub_rn(tg, v[i]);                                             // ppo.py:161
        float carry = (MODE & PPO_GAE_MASK_RECURRENCE) ? __fmul_rn(a, d) : a;
        a = __fadd_rn(__fmul_rn(gl, carry), delta);                                    // ppo.py:167
        target_out[i] = tg;
        adv_out[i] = a;
    }
}


// GAE as a wavefront scan, for the shapes where "lane = env" has nothing to run on: few envs, very
// long rollouts (the reference's own 16-env configuration has T = 40 960, ppo.py:118-122).  One wave
// per env; the 64 lanes own 64 consecutive chunks of the time axis.
//   pass 1  every lane runs the recurrence over its chunk from a zero carry (its chunk's own
//           contribution S_l) -- 64 chunks in parallel instead of one 40 960-long chain;
//   carry   the carry into chunk l is A_l = S_{l+1} + c^{len(l+1)} A_{l+1}, a linear recurrence over
//           lanes, solved with a 6-step Kogge-Stone scan of (multiplier, value) pairs on shuffles;
//   pass 2  every lane reruns its chunk from the true carry and writes target / advantage.
// Inside a chunk the arithmetic is the reference's (separately rounded mul/add); only the 63 carries
// are re-associated, so the result agrees with the sequential loop to fp32 rounding (tested at
// 1e-5), not bit for bit: the host picks this kernel only when N is small (PPO_GAE_SCAN).
template <int MODE>
__global__ __launch_bounds__(64) void ppo_td_gae_scan_kernel(
    const float* __restrict__ reward, const float* __restrict__ v, const float* __restrict__ v_next,
    const float* __restrict__ done, float gamma, float gl, long T, long N,
    float* __restrict__ target_out, float* __restrict__ adv_out)
{
    const long e = blockIdx.x;
    const int lane = threadIdx.x;
    const long L = (T + 63) / 64;
    // reverse time: lane 0 owns the LAST chunk, so the scan runs towards higher lanes
    const long t_hi = T - (long)lane * L;                 // exclusive
    const long t_lo = (t_hi - L > 0) ? t_hi - L : 0;
    const float d_row = (MODE & PPO_GAE_DONE_PER_STEP) ? 0.0f : done[e];
    auto delta_at = [&](long t, float& tg) {
        const long i = t * N + e;
        const float d = (MODE & PPO_GAE_DONE_PER_STEP) ? done[i] : d_row;
        tg = __fadd_rn(reward[i], __fmul_rn(__fmul_rn(gamma, v_next[i]), d));
        return __fsub_rn(tg, v[i]);
    };
    // pass 1: chunk contribution and multiplier
    float S = 0.0f, m = 1.0f;
    for (long t = t_hi - 1; t >= t_lo && t_hi > 0; --t) {
        float tg;
        const float dl = delta_at(t, tg);
        S = __fadd_rn(__fmul_rn(gl, S), dl);
        m *= gl;
    }
    if (t_hi <= 0) { S = 0.0f; m = 1.0f; }
    // inclusive scan over lanes (lane 0 first in reverse time): value after chunk l given zero
    // carry into chunk 0:  X_l = S_l + m_l * X_{l-1}
    float sm = m, sv = S;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const float pm = __shfl_up(sm, o, 64), pv = __shfl_up(sv, o, 64);
        if (lane >= o) { sv = sv + sm * pv; sm = sm * pm; }
    }
    float carry = __shfl_up(sv, 1, 64);                   // advantage entering this chunk
    if (lane == 0) carry = 0.0f;
    // pass 2: the real recurrence from the true carry
    float a = carry;
    for (long t = t_hi - 1; t >= t_lo && t_hi > 0; --t) {
        float tg;
        const float dl = delta_at(t, tg);
        const float cin = (MODE & PPO_GAE_MASK_RECURRENCE) ? 0.0f : a;   // masked recurrence is not supported here
        a = __fadd_rn(__fmul_rn(gl, cin), dl);
        target_out[t * N + e] = tg;
        adv_out[t * N + e] = a;
    }
}

// Advantage normalisation (named by BASELINE's north_star; the reference does NOT normalise,
// ppo.py:171, so this is opt-in): per-block partial sums of x and x^2, then (x - mean) / (std + eps)
// with the unbiased std torch uses.  Two launches; `stats` [2 + 2*blocks] is scratch, stats[0..1]
// return mean and std.  With data-parallel ranks the caller all-reduces the two partial totals.
__global__ __launch_bounds__(256) void ppo_adv_stats_kernel(const float* __restrict__ adv, long n,
                                                            float* __restrict__ part)
{
    __shared__ float red[2][4];
    float s = 0.0f, ss = 0.0f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float x = adv[i];
        s += x; ss += x * x;
    }
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_down(s, o, 64); ss += __shfl_down(ss, o, 64); }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s; red[1][threadIdx.x >> 6] = ss; }
    __syncthreads();
    if (threadIdx.x == 0) {
        part[2 * blockIdx.x] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        part[2 * blockIdx.x + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    }
}

__global__ __launch_bounds__(256) void ppo_adv_apply_kernel(float* __restrict__ adv, long n, const float* __restrict__ totals,
                                                            float count, float eps)
{
    const float mean = totals[0] / count;
    const float var = fmaxf((totals[1] - count * mean * mean) / (count - 1.0f), 0.0f);
    const float inv = 1.0f / (sqrtf(var) + eps);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
        adv[i] = (adv[i] - mean) * inv;
}

__global__ __launch_bounds__(256) void ppo_adv_totals_kernel(const float* __restrict__ part, int blocks, float* __restrict__ totals)
{
    __shared__ float red[2][4];
    float s = 0.0f, ss = 0.0f;
    for (int b = threadIdx.x; b < blocks; b += 256) { s += part[2 * b]; ss += part[2 * b + 1]; }
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_down(s, o, 64); ss += __shfl_down(ss, o, 64); }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s; red[1][threadIdx.x >> 6] = ss; }
    __syncthreads();
    if (threadIdx.x == 0) {
        totals[0] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        totals[1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    }
}

// ppo.py:233 + :237 in one tiny launch: score += mean(reward)/num_eval_freq (kept on the device:
// the reference's per-step .item() host sync is gone) and action_var = max(var_min, var - decay).
// One workgroup, fixed reduction order: deterministic.
__global__ __launch_bounds__(1024) void ppo_bookkeeping_kernel(const float* __restrict__ reward, long n,
                                                               float* __restrict__ score_acc, float score_scale,
                                                               float* __restrict__ action_var, int nvar,
                                                               float var_decay, float var_min)
{
    __shared__ float red[16];
    const int tid = threadIdx.x;
    // 16-byte loads, all of a thread's loads in flight at once (n = 8192 -> two float4 per thread)
    float s = 0.0f;
    const long n4 = n >> 2;
    const float4* r4 = reinterpret_cast<const float4*>(reward);
    const bool aligned = (reinterpret_cast<uintptr_t>(reward) & 15) == 0;
    if (aligned) {
        for (long i = tid; i < n4; i += 1024) { const float4 v = r4[i]; s += (v.x + v.y) + (v.z + v.w); }
        for (long i = 4 * n4 + tid; i < n; i += 1024) s += reward[i];
    } else {
        for (long i = tid; i < n; i += 1024) s += reward[i];
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) {
        float t = 0.0f;
        for (int w = 0; w < 16; ++w) t += red[w];
        *score_acc += t / (float)n * score_scale;
    }
    if (tid < nvar && var_decay > 0.0f) action_var[tid] = fmaxf(var_min, action_var[tid] - var_decay);
}

// The same bookkeeping for `rows` consecutive env steps at once (reward [rows][n]), bit for bit what
// `rows` calls of the kernel above leave: one workgroup per row computes that row's score term with
// the identical reduction, a second single-wave launch adds the terms in row order and applies the
// variance decay `rows` times.  The rollout calls this when the score is printed and before an
// update instead of paying a latency-bound single-workgroup launch on every env step.
__global__ __launch_bounds__(1024) void ppo_row_terms_kernel(const float* __restrict__ reward, long n, float score_scale,
                                                             float* __restrict__ terms)
{
    __shared__ float red[16];
    const int tid = threadIdx.x;
    const float* row = reward + (long)blockIdx.x * n;
    float s = 0.0f;
    const long n4 = n >> 2;
    const float4* r4 = reinterpret_cast<const float4*>(row);
    const bool aligned = (reinterpret_cast<uintptr_t>(row) & 15) == 0;
    if (aligned) {
        for (long i = tid; i < n4; i += 1024) { const float4 v = r4[i]; s += (v.x + v.y) + (v.z + v.w); }
        for (long i = 4 * n4 + tid; i < n; i += 1024) s += row[i];
    } else {
        for (long i = tid; i < n; i += 1024) s += row[i];
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) {
        float t = 0.0f;
        for (int w = 0; w < 16; ++w) t += red[w];
        terms[blockIdx.x] = t / (float)n * score_scale;
    }
}

__global__ __launch_bounds__(64) void ppo_rows_apply_kernel(const float* __restrict__ terms, long rows,
                                                            float* __restrict__ score_acc, float* __restrict__ action_var,
                                                            int nvar, float var_decay, float var_min, int* __restrict__ rows_applied)
{
    const int tid = threadIdx.x;
    if (tid == 63) {
        float acc = *score_acc;
        for (long r = 0; r < rows; ++r) acc += terms[r];
        *score_acc = acc;
        if (rows_applied) *rows_applied += (int)rows;       // the policy launches subtract this from their row index
    }
    if (tid < nvar && var_decay > 0.0f) {
        float v = action_var[tid];
        for (long r = 0; r < rows; ++r) v = fmaxf(var_min, v - var_decay);
        action_var[tid] = v;
    }
}

}  // namespace

extern "C" hipError_t flyhip_launch_rollout_bookkeeping(const float* reward, int64_t rows, int64_t n, float* terms,
                                                        float* score_acc, float score_scale, float* action_var, int nvar,
                                                        float var_decay, float var_min, int* rows_applied, void* stream)
{
    hipLaunchKernelGGL(ppo_row_terms_kernel, dim3((unsigned)rows), dim3(1024), 0, (hipStream_t)stream, reward, (long)n,
                       score_scale, terms);
    hipLaunchKernelGGL(ppo_rows_apply_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, terms, (long)rows, score_acc,
                       action_var, nvar, var_decay, var_min, rows_applied);
    return hipGetLastError();
}

extern "C" hipError_t flyhip_launch_bookkeeping(const float* reward, int64_t n, float* score_acc, float score_scale,
                                                float* action_var, int nvar, float var_decay, float var_min,
                                                void* stream)
{
    hipLaunchKernelGGL(ppo_bookkeeping_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, reward, (long)n, score_acc,
                       score_scale, action_var, nvar, var_decay, var_min);
    return hipGetLastError();
}

extern "C" hipError_t flyhip_launch_sample_logprob(const float* mu, const float* var, const float* eps,
                                                   float* act_out, float* logp_out, int64_t n, void* stream)
{
    int grid = (int)((n + SL_BLOCK - 1) / SL_BLOCK);
    hipLaunchKernelGGL(ppo_sample_logprob_kernel, dim3(grid), dim3(SL_BLOCK), 0, (hipStream_t)stream,
                       mu, var, eps, act_out, logp_out, (long)n);
    return hipGetLastError();
}

extern "C" hipError_t flyhip_launch_td_gae(const float* reward, const float* v, const float* v_next,
                                           const float* done, float gamma, float lambda, int64_t T, int64_t N,
                                           float* target_out, float* adv_out, int mode, void* stream)
{
    const float gl = (float)((double)gamma * (double)lambda);   // python double product, ppo.py:167
    if (mode & PPO_GAE_SCAN) {
        if (mode & PPO_GAE_DONE_PER_STEP)
            hipLaunchKernelGGL((ppo_td_gae_scan_kernel<1>), dim3((unsigned)N), dim3(64), 0, (hipStream_t)stream, reward, v,
                               v_next, done, gamma, gl, (long)T, (long)N, target_out, adv_out);
        else
            hipLaunchKernelGGL((ppo_td_gae_scan_kernel<0>), dim3((unsigned)N), dim3(64), 0, (hipStream_t)stream, reward, v,
                               v_next, done, gamma, gl, (long)T, (long)N, target_out, adv_out);
        return hipGetLastError();
    }
    int block = 64;                                              // one wave per workgroup: spread envs over CUs
    int grid = (int)((N + block - 1) / block);
#define GAE(M) hipLaunchKernelGGL((ppo_td_gae_kernel<M>), dim3(grid), dim3(block), 0, (hipStream_t)stream, \
                                  reward, v, v_next, done, gamma, gl, (long)T, (long)N, target_out, adv_out)
    switch (mode & 3) {
    case 0: GAE(0); break;
    case 1: GAE(1); break;
    case 2: GAE(2); break;
    default: GAE(3); break;
    }
#undef GAE
    return hipGetLastError();
}

extern "C" hipError_t flyhip_launch_adv_stats(const float* adv, int64_t n, float* stats, void* stream)
{
    const int blocks = 256;
    hipLaunchKernelGGL(ppo_adv_stats_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, adv, (long)n, stats + 2);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(ppo_adv_totals_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, stats + 2, blocks, stats);
    return hipGetLastError();
}

extern "C" hipError_t flyhip_launch_adv_apply(float* adv, int64_t n, const float* totals, float count, float eps, void* stream)
{
    hipLaunchKernelGGL(ppo_adv_apply_kernel, dim3(512), dim3(256), 0, (hipStream_t)stream, adv, (long)n, totals, count, eps);
    return hipGetLastError();
}
