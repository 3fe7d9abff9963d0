// dqn_kernels.hip — the two per-row pieces of the reference's DQN variant
// (UselessFiles/dqn.py, configs[4]) that it runs as Python loops / op chains:
//   dqn_eps_greedy_kernel  dqn.py:89-100  per-env first-argmax, eps-greedy mix, map to [-1,1]
//   dqn_huber_td_kernel    dqn.py:68-79   TD target + Huber loss AND its gradient w.r.t. the Q table
// Both are streaming kernels over row-major [rows][A] Q tables (A = 18: 72-byte rows, one row per
// lane; the tables come straight out of the Q-network GEMM).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "flyhip.h"

namespace {

__global__ __launch_bounds__(256) void dqn_eps_greedy_kernel(const float* __restrict__ q, const float* __restrict__ coin_u,
                                                             const float* __restrict__ rand_u, float epsilon, int A,
                                                             float* __restrict__ act_out, long n)
{
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const float* row = q + e * A;
    float best = row[0];
    int idx = 0;
    for (int a = 1; a < A; ++a) {            // (q == q.max()).nonzero()[0]: FIRST maximal entry
        const float v = row[a];
        if (v > best) { best = v; idx = a; }
    }
    const float true_act = (float)idx / (float)(A - 1);
    const float act = (coin_u[e] < epsilon) ? rand_u[e] : true_act;
    act_out[e] = 2.0f * (act - 0.5f);        // maps to -1 .. 1
}

__global__ __launch_bounds__(256) void dqn_huber_td_kernel(const float* __restrict__ q_table, const float* __restrict__ act,
                                                           const float* __restrict__ reward, const float* __restrict__ q_next,
                                                           const float* __restrict__ done, float discount, int A, long B,
                                                           float inv_B, float* __restrict__ dq, float* __restrict__ loss_part)
{
    __shared__ float red[4];
    const long b = (long)blockIdx.x * blockDim.x + threadIdx.x;
    float hub = 0.0f;
    if (b < B) {
        const float a01 = 0.5f * (act[b] + 1.0f);
        int idx = (int)rintf(a01 * (float)(A - 1));             // torch.round: half to even
        idx = idx < 0 ? 0 : (idx >= A ? A - 1 : idx);
        const float* qn = q_next + b * A;
        float mx = qn[0];
        for (int a = 1; a < A; ++a) mx = fmaxf(mx, qn[a]);
        const float target = reward[b] + discount * mx * done[b];
        const float d = q_table[b * A + idx] - target;
        hub = fabsf(d) < 1.0f ? 0.5f * d * d : fabsf(d) - 0.5f;  // smooth_l1, beta = 1
        float* g = dq + b * A;
        for (int a = 0; a < A; ++a) g[a] = 0.0f;
        g[idx] = inv_B * fminf(fmaxf(d, -1.0f), 1.0f);
    }
    for (int o = 32; o > 0; o >>= 1) hub += __shfl_down(hub, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = hub;
    __syncthreads();
    if (threadIdx.x == 0) loss_part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

}  // namespace

extern "C" hipError_t flyhip_launch_dqn_eps_greedy(const float* q, const float* coin_u, const float* rand_u, float epsilon,
                                                   int A, float* act_out, int64_t n, void* stream)
{
    hipLaunchKernelGGL(dqn_eps_greedy_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, q,
                       coin_u, rand_u, epsilon, A, act_out, (long)n);
    return hipGetLastError();
}

extern "C" hipError_t flyhip_launch_dqn_huber_td(const float* q_table, const float* act, const float* reward,
                                                 const float* q_next, const float* done, float discount, int A, int64_t B,
                                                 float* dq, float* loss_part, void* stream)
{
    hipLaunchKernelGGL(dqn_huber_td_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, (hipStream_t)stream, q_table,
                       act, reward, q_next, done, discount, A, (long)B, 1.0f / (float)B, dq, loss_part);
    return hipGetLastError();
}
