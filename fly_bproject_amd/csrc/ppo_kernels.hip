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

}  // namespace

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
