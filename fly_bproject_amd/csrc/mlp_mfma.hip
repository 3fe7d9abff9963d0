// mlp_mfma.hip — the actor-critic MLP (reference ppo.py:10-102) on the gfx950 matrix cores:
// translation unit of the MLP kernels (the pieces live in the .inc files it includes) and their
// launchers.
//
//   mlp_gemm.inc      tile GEMMs: straight-line runs of v_mfma_f32_32x32x2_f32 (fp32 in, fp32
//                     accumulate: the reference's fp32 numerics on the matrix pipe) over a ring of
//                     named operand fragments, and the bf16x3 variant on v_mfma_f32_32x32x16_bf16;
//                     bias + ELU epilogues working on accumulator fragments
//   mlp_forward.inc   all four layers of a 32-row tile in one workgroup: activations stay in LDS
//                     between layers, weights stream from L2 in fragment order (mlp_layout.h),
//                     layers 1/2 in two halves of 128 hidden columns, layer 4 split-K over the waves,
//                     optional fused action sampling
//   mlp_backward.inc  PPO loss gradient + dX chain of a tile, and the ONE-launch forward+backward
//                     kernel whose backward workgroups wait on per-tile flags of the forward ones
//   mlp_grad_w.inc    dW = dZ^T A over row slabs (one partial per workgroup) + fixed-order reduction
//   mlp_fused_step.inc  ONE persistent launch per minibatch: forward, loss, dX chain and dW of every tile in the workgroup
//                     that owns it (bf16x3; activations never leave the CU, dW accumulated in registers)
//   mlp_adam.inc      clip_grad_norm_ + Adam on the packed parameters
//
// The operand roles are fixed throughout: weights are the MFMA "A" operand, activations the "B"
// operand, so a lane of a 32x32 result tile owns one batch row and groups of four consecutive
// output columns.  DESIGN.md section 3.4 has the measurements this structure follows from.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include <cstdio>
#include <cstdlib>
#include "flyhip.h"
#include "mlp_layout.h"

namespace {

#include "mlp_gemm.inc"
#include "fs_stamp.inc"
#include "mlp_forward.inc"
#include "mlp_backward.inc"
#include "mlp_grad_w.inc"
#include "mlp_fused_step.inc"
#include "mlp_adam.inc"
#include "fly_body.inc"

// One launch per env step of the rollout (ppo.py:213-230): workgroup b runs the policy forward +
// sampling for the 32 envs of tile b (writing their action / log-prob / value rows) and then the
// fused env step of exactly those envs (fly.py:624-681) -- the env body has the same 256-thread,
// 32-env shape.  The actions go through HBM rows the workgroup itself just wrote (it waits for its
// stores and re-reads them: same CU, and nothing has read those lines since the kernel began).
// Bit for bit what mlp_forward_sample followed by fly_step leave.
constexpr int RS_LDS_FLOATS = FWD_LDS_FLOATS > ENVS_PER_BLOCK * FLY_NUM_OBS ? FWD_LDS_FLOATS : ENVS_PER_BLOCK * FLY_NUM_OBS;
constexpr int RS_B3_LDS_FLOATS = FWD_B3_LDS_FLOATS > ENVS_PER_BLOCK * FLY_NUM_OBS ? FWD_B3_LDS_FLOATS : ENVS_PER_BLOCK * FLY_NUM_OBS;
static_assert(ENVS_PER_BLOCK == BM && BLOCK == THREADS, "one forward tile = one env block");

// B3: the policy body on the bf16 matrix pipe (three-term splits; PF then points at the term planes)
template <bool B3>
__global__ __launch_bounds__(THREADS, B3 ? 1 : 2) void rollout_step_kernel(
    const FlyConfig* __restrict__ c, FlyBuffers b, const float* __restrict__ P, const void* __restrict__ PF,
    const float* __restrict__ x, long n, const float* __restrict__ eps, const float* __restrict__ var, int var_steps,
    float var_decay, float var_min, float* __restrict__ act, float* __restrict__ logp, float* __restrict__ v_out,
    const int* __restrict__ var_base)
{
    __shared__ __attribute__((aligned(16))) float lds[B3 ? RS_B3_LDS_FLOATS : RS_LDS_FLOATS];
    var_steps -= var_base ? *var_base : 0;                  // pending decays: frozen row index minus what is already applied
    constexpr int PH_ALL = PH_SCALE | PH_RESET | PH_INTEGRATE | PH_OBS | PH_PROGRESS | PH_REWARD;
    FlyRegs st;
    fly_load<PH_ALL>(st, c, b, blockIdx.x);                 // the env state's HBM round trip hides under the forward
    if (B3)
        forward_body_b3<false>(lds, blockIdx.x, 1L << 40, P, static_cast<const u16*>(PF), x, n, nullptr, v_out, nullptr, nullptr,
                               nullptr, nullptr, eps, var, act, logp, nullptr, var_steps, var_decay, var_min);
    else
        forward_body<false>(lds, blockIdx.x, 1L << 40, P, static_cast<const float*>(PF), x, n, nullptr, v_out, nullptr, nullptr,
                            nullptr, nullptr, eps, var, act, logp, nullptr, var_steps, var_decay, var_min);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);          // this thread's action stores are acknowledged by L2
    __syncthreads();
    fly_body<PH_ALL>(c, act, b, lds, blockIdx.x, st);
}

// One launch per ROLLOUT (ppo.py:204-237 for T consecutive env steps): the envs of a 32-env tile depend on no other
// tile and the policy does not change inside a rollout, so workgroup b simply loops over the T steps of ITS tile --
// policy forward + sampling on observation row t, env step, observation row t + 1 -- with the env state carried in
// registers.  What a launch per step pays T times (the launch boundary, the prologue, the state's round trip) is paid
// once.  Row t's buffers are row t of the rollout tensors (fixed strides); the deferred bookkeeping is exact because
// row t decays the variance t - *rows_applied times and no flush can run while the launch does.  Between steps the
// workgroup waits for its own observation-row stores (vmcnt(0) + barrier): row t + 1 is read only by the workgroup
// that wrote it, at addresses this CU has not read before.  Bit for bit what T launches of rollout_step_kernel leave.
// WPS = waves per SIMD the registers are budgeted for: 1 when the launch has at most one workgroup per CU (<= 8192
// envs: the step body plus the carried env state want ~300 registers and spill at 256), 2 beyond that.
template <bool B3, int WPS>
__global__ __launch_bounds__(THREADS, WPS) void rollout_all_kernel(
    const FlyConfig* __restrict__ c, FlyBuffers b, const float* __restrict__ P, const void* __restrict__ PF,
    float* __restrict__ obs_ring, long n, const float* __restrict__ eps_all, const float* __restrict__ var, float var_decay,
    float var_min, float* __restrict__ act_all, float* __restrict__ logp_all, float* __restrict__ v_ring,
    float* __restrict__ reward_all, int T, const int* __restrict__ rows_applied, int64_t* __restrict__ reset_rows,
    int64_t* __restrict__ progress_rows)
{
    constexpr int ARENA = B3 ? RS_B3_LDS_FLOATS : RS_LDS_FLOATS;
    __shared__ __attribute__((aligned(16))) float lds[ARENA + 32 + BM * MLP_NACT];
    constexpr int PH_ALL = PH_SCALE | PH_RESET | PH_INTEGRATE | PH_OBS | PH_PROGRESS | PH_REWARD;
    // the variance of the CURRENT row lives in LDS and is decayed once per step (ppo.py:236-237): the same sequence of
    // fp32 operations as the per-step form, O(1) per step however long the rollout (T = 40 960 at 16 envs)
    float* varcur = lds + ARENA;
    // bf16x3 body: the two hand-offs between the policy and the env step of a tile stay in LDS -- the sampled actions go to
    // `acts` as well as to their HBM row, and the observation row the env step stages in the arena IS the next step's
    // policy input -- so neither waits for its stores to be acknowledged nor reads its own row back through L2
    // (the HBM rows are still written: the update reads them).  Same values, same bits.
    float* acts = B3 ? lds + ARENA + 32 : nullptr;
    // the launch is the FIRST device work of its rollout: *rows_applied was zeroed just before it and no bookkeeping
    // flush can run until it has finished, so `var` is exactly the variance of row 0
    if (threadIdx.x < MLP_NACT) varcur[threadIdx.x] = var[threadIdx.x];
    FlyRegs st;
    fly_load<PH_ALL>(st, c, b, blockIdx.x);
    __syncthreads();
    const bool whole = (long)(blockIdx.x + 1) * BM <= n;         // a ragged last tile keeps the HBM hand-offs (its LDS rows are partly stale)
    for (int t = 0; t < T; ++t) {
        const float* x = obs_ring + (long)t * n * FLY_NUM_OBS;
        float* act = act_all + (long)t * n * MLP_NACT;
        b.obs = obs_ring + (long)(t + 1) * n * FLY_NUM_OBS;
        b.reward = reward_all + (long)t * n;
        if (reset_rows) { b.reset = reset_rows + (long)t * n; b.progress = progress_rows + (long)t * n; }   // fly.py:175-177, per step
        const bool in_lds = B3 && whole;
        if (B3)
            forward_body_b3<false>(lds, blockIdx.x, 1L << 40, P, static_cast<const u16*>(PF), x, n, nullptr, v_ring + (long)t * n,
                                   nullptr, nullptr, nullptr, nullptr, eps_all + (long)t * n * MLP_NACT, varcur, act,
                                   logp_all + (long)t * n, nullptr, 0, var_decay, var_min,
                                   in_lds && t > 0 ? lds : nullptr, in_lds ? acts : nullptr, t > 0);
        else
            forward_body<false>(lds, blockIdx.x, 1L << 40, P, static_cast<const float*>(PF), x, n, nullptr, v_ring + (long)t * n,
                                nullptr, nullptr, nullptr, nullptr, eps_all + (long)t * n * MLP_NACT, varcur, act,
                                logp_all + (long)t * n, nullptr, 0, var_decay, var_min);
        if (!in_lds) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_s_waitcnt(0);      // this thread's action stores are acknowledged by L2
        }
        __syncthreads();
        FlyRegs nx;
        fly_body<PH_ALL>(c, act, b, lds, blockIdx.x, st, &nx, in_lds ? acts : nullptr);
        st = nx;
        if (threadIdx.x < MLP_NACT && var_decay > 0.0f) varcur[threadIdx.x] = fmaxf(var_min, varcur[threadIdx.x] - var_decay);
        if (!in_lds) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // observation row t + 1 is in L2 before any wave of this workgroup reads it
        __syncthreads();
    }
}

// The same loop with the policy in the fused step's style (policy_tile_fs, mlp_fused_step.inc): swizzled plane images, the chain
// GEMMs' asm MFMA streams, 135 KB of LDS -- one workgroup per CU, whole tiles only (the launcher checks both).  The env step
// stages its observation block in H2's region of that image, which is where the next step's policy converts it from.
// STAMP (diagnostic instantiation, bench.py / tools/ab_rollout.py): thread 0 of every workgroup stores s_memtime into stamps
// u64 [tile][T + 1][8]: slot 0 the top of step t, 1 .. 5 inside the policy (x converted, layer 1, 2, 3, 4 done), 7 between the
// policy and env halves (+ the 100 MHz real-time counter once, at [T][1]).
#ifndef FR_HOIST
#define FR_HOIST 1          // layer 1's first weights of the NEXT step are requested before the env step (A/B: 0)
#endif
// MULTI: more tiles than workgroups (> 8192 envs on 256 CUs): the outer loop really loops; the single-tile instantiation keeps
// the register allocation of a kernel without it.
template <bool STAMP, bool MULTI>
__global__ __launch_bounds__(THREADS, 1) void rollout_all_fs_kernel(
    const FlyConfig* __restrict__ c, FlyBuffers b, const float* __restrict__ P, const u16* __restrict__ PB,
    float* __restrict__ obs_ring, long n, const float* __restrict__ eps_all, const float* __restrict__ var, float var_decay,
    float var_min, float* __restrict__ act_all, float* __restrict__ logp_all, float* __restrict__ v_ring,
    float* __restrict__ reward_all, int T, int64_t* __restrict__ reset_rows, int64_t* __restrict__ progress_rows,
    unsigned long long* __restrict__ stamps)
{
    extern __shared__ __attribute__((aligned(16))) u16 fr_lds[];
    const FrLds L(fr_lds);
    constexpr int PH_ALL = PH_SCALE | PH_RESET | PH_INTEGRATE | PH_OBS | PH_PROGRESS | PH_REWARD;
    policy_tile_fs_setup(L, P);
    int64_t* const reset0 = b.reset;                          // the CURRENT flags: read once per tile, before the tile's step 0
    int64_t* const progress0 = b.progress;
    const long ntiles = n / BM;                               // whole tiles only (the launcher checks)
    // persistent over tiles: workgroup g runs the T steps of tiles g, g + grid, ... one after the other (one tile per workgroup at
    // <= 8192 envs on 256 CUs; two at 16384).  Tiles are independent, so the order is free and every row equals the per-step launches'.
    long tile = blockIdx.x;
    do {
        unsigned long long* st_tile = STAMP ? stamps + tile * (T + 1) * 8 : nullptr;
        if (threadIdx.x < 32) L.varcur[threadIdx.x] = threadIdx.x < MLP_NACT ? var[threadIdx.x] : 1.0f;   // the variance of row 0 (see rollout_all_kernel)
        b.reset = reset0; b.progress = progress0;
        FlyRegs st;
        fly_load<PH_ALL>(st, c, b, (int)tile);
        FrHead w1;
        policy_tile_fs_head(w1, PB);
        __syncthreads();
        if (STAMP && threadIdx.x == 0) st_tile[6] = realtime_cu();      // (with [T][1]: the shader clock the stamps tick at)
        for (int t = 0; t < T; ++t) {
            stamp<STAMP>(st_tile, 8 * t);
            float* act = act_all + (long)t * n * MLP_NACT;
            b.obs = obs_ring + (long)(t + 1) * n * FLY_NUM_OBS;
            b.reward = reward_all + (long)t * n;
            if (reset_rows) { b.reset = reset_rows + (long)t * n; b.progress = progress_rows + (long)t * n; }   // fly.py:175-177, per step
            if (!FR_HOIST && t > 0) policy_tile_fs_head(w1, PB);
            policy_tile_fs<STAMP>(L, tile, PB, t == 0 ? obs_ring : nullptr, n, v_ring + (long)t * n, eps_all + (long)t * n * MLP_NACT, act,
                                  logp_all + (long)t * n, w1, STAMP ? st_tile + 8 * t : nullptr);
            stamp<STAMP>(st_tile, 8 * t + 7);
            if (FR_HOIST) policy_tile_fs_head(w1, PB);     // the NEXT step's first weights: their round trip hides under the physics
            FlyRegs nx;
            fly_body<PH_ALL>(c, act, b, L.obs, (int)tile, st, &nx, L.acts);
            st = nx;
            if (threadIdx.x < MLP_NACT && var_decay > 0.0f) L.varcur[threadIdx.x] = fmaxf(var_min, L.varcur[threadIdx.x] - var_decay);
            __syncthreads();
        }
        stamp<STAMP>(st_tile, 8 * T);
        if (STAMP && threadIdx.x == 0) st_tile[8 * T + 1] = realtime_cu();
        tile += gridDim.x;
    } while (MULTI && tile < ntiles);
}

}  // namespace

extern "C" hipError_t flyhip_launch_rollout_all(const FlyConfig* dcfg, const FlyBuffers* b, const float* P, const float* PF,
                                                float* obs_ring, int64_t n, const float* eps_all, const float* var,
                                                float var_decay, float var_min, float* act_all, float* logp_all, float* v_ring,
                                                float* reward_all, int T, const int* rows_applied, const uint16_t* PB,
                                                int64_t* reset_rows, int64_t* progress_rows, void* stream,
                                                unsigned long long* stamps)
{
    const dim3 grid((unsigned)((n + BM - 1) / BM));
    int cus = 256;
    { int dev = 0; hipDeviceProp_t pr; if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) cus = pr.multiProcessorCount; }
    const bool one_per_cu = (int)grid.x <= cus;
    const dim3 grid_fs((unsigned)((int)grid.x <= cus ? (int)grid.x : cus));   // the fused-style kernel walks its tiles itself
    const char* fs_env = getenv("FLY_ROLLOUT_FS");               // read per launch: the tests flip it inside one process
    const bool fs_off = fs_env != nullptr && fs_env[0] == '0';
    if (PB && n % BM == 0 && !fs_off) {       // the policy body in the fused step's style (A/B: FLY_ROLLOUT_FS=0); persistent over tiles
        const bool multi = (int)grid.x > cus;
        const int si = (stamps ? 1 : 0) + (multi ? 2 : 0);
        const void* fn = si == 0 ? reinterpret_cast<const void*>(rollout_all_fs_kernel<false, false>)
                       : si == 1 ? reinterpret_cast<const void*>(rollout_all_fs_kernel<true, false>)
                       : si == 2 ? reinterpret_cast<const void*>(rollout_all_fs_kernel<false, true>)
                                 : reinterpret_cast<const void*>(rollout_all_fs_kernel<true, true>);
        {       // (per launch: the attribute belongs to the CURRENT device)
            hipError_t ea = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, FR_LDS_BYTES);
            if (ea != hipSuccess) return ea;
        }
#define RAFS_LAUNCH(S_, M_)                                                                                                           \
        hipLaunchKernelGGL((rollout_all_fs_kernel<S_, M_>), grid_fs, dim3(THREADS), FR_LDS_BYTES, (hipStream_t)stream, dcfg, *b, P, PB,   \
                           obs_ring, (long)n, eps_all, var, var_decay, var_min, act_all, logp_all, v_ring, reward_all, T, reset_rows,  \
                           progress_rows, stamps)
        if (si == 0) RAFS_LAUNCH(false, false); else if (si == 1) RAFS_LAUNCH(true, false);
        else if (si == 2) RAFS_LAUNCH(false, true); else RAFS_LAUNCH(true, true);
#undef RAFS_LAUNCH
        return hipGetLastError();
    }
    if (stamps) return hipErrorInvalidValue;        // only the fused-style kernel has a stamped instantiation
#define RA_LAUNCH(B3_, WPS_, PF_)                                                                                                  \
    hipLaunchKernelGGL((rollout_all_kernel<B3_, WPS_>), grid, dim3(THREADS), 0, (hipStream_t)stream, dcfg, *b, P, (const void*)PF_, \
                       obs_ring, (long)n, eps_all, var, var_decay, var_min, act_all, logp_all, v_ring, reward_all, T, rows_applied, reset_rows, \
                       progress_rows)
    if (PB) { if (one_per_cu) RA_LAUNCH(true, 1, PB); else RA_LAUNCH(true, 1, PB); }      // the bf16x3 body needs one wave per SIMD anyway
    else { if (one_per_cu) RA_LAUNCH(false, 1, PF); else RA_LAUNCH(false, 2, PF); }
#undef RA_LAUNCH
    return hipGetLastError();
}

extern "C" hipError_t flyhip_launch_mlp_forward(const float* P, const float* PF, const float* x, int64_t n, float* mu_out,
                                                float* v_out, float* out_save, float* h1_save, float* h2_save,
                                                float* h3_save, const uint16_t* PB, void* stream)
{
    if (PB) {       // bf16x3 GEMMs
        const long tiles = (n + BM - 1) / BM;
        const int grid = (int)tiles;
        hipLaunchKernelGGL(mlp_forward_b3_kernel, dim3(grid), dim3(THREADS), 0, (hipStream_t)stream, P, PB, x, (long)n, mu_out,
                           v_out, out_save, h1_save, h2_save, h3_save, (const float*)nullptr, (const float*)nullptr,
                           (float*)nullptr, (float*)nullptr, 0, 0.0f, 0.0f, (const int*)nullptr);
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
                       (float*)nullptr, (float*)nullptr, (unsigned long long*)nullptr, 0, 0.0f, 0.0f, (const int*)nullptr);
    return hipGetLastError();
}

extern "C" hipError_t flyhip_launch_mlp_forward_sample(const float* P, const float* PF, const float* x, int64_t n,
                                                       const float* eps, const float* var, int var_steps,
                                                       float var_decay, float var_min, float* act_out,
                                                       float* logp_out, float* mu_out, float* v_out, const uint16_t* PB,
                                                       const int* var_base, void* stream)
{
    if (PB) {
        const long tiles = (n + BM - 1) / BM;
        const int grid = (int)tiles;
        hipLaunchKernelGGL(mlp_forward_b3_kernel, dim3(grid), dim3(THREADS), 0, (hipStream_t)stream, P, PB, x, (long)n, mu_out,
                           v_out, (float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr, eps, var, act_out,
                           logp_out, var_steps, var_decay, var_min, var_base);
        return hipGetLastError();
    }
    const long tiles = (n + BM - 1) / BM;
    const int grid = (int)(tiles <= 4 * PERSIST_GRID ? tiles : PERSIST_GRID);
    hipLaunchKernelGGL(mlp_forward_kernel<false>, dim3(grid), dim3(THREADS), 0, (hipStream_t)stream, P, PF, x, (long)n,
                       mu_out, v_out, (float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr, eps,
                       var, act_out, logp_out, (unsigned long long*)nullptr, var_steps, var_decay, var_min, var_base);
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
                       (const float*)nullptr, (float*)nullptr, (float*)nullptr, stamps, 0, 0.0f, 0.0f, (const int*)nullptr);
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
// the bf16x3 kernel (grad_w_b3.inc) pays a fixed staging + barrier cost per 16-row chunk, so its thin layers get
// the kernel streams: a CU pulls ~22 GB/s, so the split follows each layer's BYTES (54 / 63 / 42 / 26 MB), not its FLOPs
static int kGradWgsB3[4] = {72, 88, 64, 32};
static bool gw_env_read = false;
static void gw_read_env()
{   // tuning aid: FLYHIP_GW_SPLIT="a,b,c,d" overrides the split (sum <= 256)
    if (gw_env_read) return;
    gw_env_read = true;
    const char* e = getenv("FLYHIP_GW_SPLIT");
    int v[4];
    if (e && sscanf(e, "%d,%d,%d,%d", &v[0], &v[1], &v[2], &v[3]) == 4 && v[0] > 0 && v[1] > 0 && v[2] > 0 && v[3] > 0 &&
        v[0] + v[1] + v[2] + v[3] <= 1024)
        for (int i = 0; i < 4; ++i) kGradWgs[i] = kGradWgsB3[i] = v[i];
}

static int g_fb_consumer_shift = 0;     // test hook: moves the consumer range off its producers' XCDs (forces err = 2 in xcd mode)
extern "C" void flyhip_debug_set_fwd_bwd_consumer_shift(int shift) { g_fb_consumer_shift = shift; }

extern "C" hipError_t flyhip_launch_mlp_fwd_bwd(const float* P, const float* PF, const float* PT, const float* x, int64_t n,
                                                float* out_save, float* h1_save, float* h2_save, float* h3_save,
                                                const float* action, const float* old_logp, const float* adv,
                                                const float* target, const float* var, float inv_batch, float clip,
                                                float* dz4, float* dz3, float* dz2, float* dz1, float* loss_part,
                                                int* flags, int epoch, int* err, const uint16_t* PB, const uint16_t* PTB,
                                                int coherent, void* stream)
{
    const long tiles = (n + BM - 1) / BM;
    const int shift = g_fb_consumer_shift;
    const long pad_tiles = ((tiles + 7) & ~7L) + shift;
    const dim3 grid((unsigned)(pad_tiles + tiles));
    const bool b3 = PB && PTB;
    const void* pf = b3 ? (const void*)PB : (const void*)PF;
    const void* pt = b3 ? (const void*)PTB : (const void*)PT;
#define FB_LAUNCH(B3_, COH_)                                                                                                   \
    hipLaunchKernelGGL((mlp_fwd_bwd_kernel<false, B3_, COH_>), grid, dim3(THREADS), 0, (hipStream_t)stream, P, pf, pt, x,      \
                       (long)n, out_save, h1_save, h2_save, h3_save, action, old_logp, adv, target, var, inv_batch, clip, dz4, \
                       dz3, dz2, dz1, loss_part, flags, epoch, err, (unsigned long long*)nullptr, shift)
    if (b3) { if (coherent) FB_LAUNCH(true, true); else FB_LAUNCH(true, false); }
    else { if (coherent) FB_LAUNCH(false, true); else FB_LAUNCH(false, false); }
#undef FB_LAUNCH
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
    hipLaunchKernelGGL((mlp_fwd_bwd_kernel<true, false, true>), dim3((unsigned)(pad_tiles + tiles)), dim3(THREADS), 0, (hipStream_t)stream, P,
                       (const void*)PF, (const void*)PT, x, (long)n, out_save, h1_save, h2_save, h3_save, action, old_logp, adv, target, var, inv_batch,
                       clip, dz4, dz3, dz2, dz1, loss_part, flags, epoch, err, stamps, 0);
    return (int)hipGetLastError();
}

extern "C" hipError_t flyhip_launch_rollout_step(const FlyConfig* dcfg, const FlyBuffers* b, const float* P, const float* PF,
                                                 const float* x, int64_t n, const float* eps, const float* var, int var_steps,
                                                 float var_decay, float var_min, float* act, float* logp, float* v_out,
                                                 const uint16_t* PB, const int* var_base, void* stream)
{
    const dim3 grid((unsigned)((n + BM - 1) / BM));
    if (PB)
        hipLaunchKernelGGL(rollout_step_kernel<true>, grid, dim3(THREADS), 0, (hipStream_t)stream, dcfg, *b, P, (const void*)PB,
                           x, (long)n, eps, var, var_steps, var_decay, var_min, act, logp, v_out, var_base);
    else
        hipLaunchKernelGGL(rollout_step_kernel<false>, grid, dim3(THREADS), 0, (hipStream_t)stream, dcfg, *b, P, (const void*)PF,
                           x, (long)n, eps, var, var_steps, var_decay, var_min, act, logp, v_out, var_base);
    return hipGetLastError();
}

extern "C" int64_t flyhip_mlp_grad_workspace_floats(void)
{
    gw_read_env();
    int64_t need = 0;
    for (const int* w : {kGradWgs, kGradWgsB3}) {
        const int64_t f = (int64_t)w[0] * MLP_H1 * (MLP_IN_PAD + 1) + (int64_t)w[1] * MLP_H2 * (MLP_H1 + 1) +
                          (int64_t)w[2] * MLP_H3 * (MLP_H2 + 1) + (int64_t)w[3] * MLP_OUT * (MLP_H3 + 1);   // N*KP + N per slab
        need = f > need ? f : need;
    }
    return need;
}

extern "C" hipError_t flyhip_launch_mlp_grad_w(const float* x, const float* h1, const float* h2, const float* h3,
                                               const float* dz1, const float* dz2, const float* dz3, const float* dz4,
                                               int64_t n, float* workspace, float* grad_out, const float* norm_mask,
                                               float* norm_ws, int* norm_step, const int* err, int gemm_b3, void* stream)
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
    const int* wgs = gemm_b3 ? kGradWgsB3 : kGradWgs;
    for (int l = 0; l < 4; ++l) {
        T.l[l].dz = dz[l]; T.l[l].a = a[l]; T.l[l].partial = ws;
        T.l[l].N = N[l]; T.l[l].Ka = Ka[l]; T.l[l].KP = KP[l]; T.l[l].wgs = wgs[l]; T.l[l].first_block = first;
        T.l[l].accumulate = 0; T.l[l].chunked = 0;
        ws += (long)wgs[l] * ((long)N[l] * KP[l] + N[l]);
        first += wgs[l];
    }
    // dynamic LDS: two buffers of the largest layer's chunk (padded pitches): 2 x 32 x (136 + 264) floats = 100 KiB
    const size_t lds_bytes = sizeof(float) * 2 * GW_ROWS * (MLP_H2 + GW_PAD + MLP_H1 + GW_PAD);
    static_assert(MLP_H1 + GW_PAD + 96 <= MLP_H2 + GW_PAD + MLP_H1 + GW_PAD, "layer 1 chunk fits");
    {       // (per launch: the attribute belongs to the CURRENT device)
        hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_grad_w_kernel),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (ea != hipSuccess) return ea;
    }
    if (gemm_b3) {
        // two buffers of three bf16 term planes of a chunk: layers 1 / 2 stage 16 rows x (288 + 160) columns (84 KiB),
        // layer 3 32 rows x (160 + 160) (120 KiB, the largest), layer 4 32 rows x (32 + 160) (72 KiB)
        const size_t b3_bytes = 2 * 3 * GB_ROWS_L3 * 2 * gb_pitch<MLP_H3>() * sizeof(u16);
        static_assert(16 * (gb_pitch<MLP_H1>() + gb_pitch<MLP_H2>()) <= GB_ROWS_L3 * 2 * gb_pitch<MLP_H3>() &&
                      GB_ROWS_L4 * (gb_pitch<MLP_OUT>() + gb_pitch<MLP_H3>()) <= GB_ROWS_L3 * 2 * gb_pitch<MLP_H3>(), "layer 3's chunk is the largest");
        {       // (per launch: the attribute belongs to the CURRENT device)
            hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_grad_w_b3_kernel),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)b3_bytes);
            if (ea != hipSuccess) return ea;
        }
        hipLaunchKernelGGL(mlp_grad_w_b3_kernel, dim3(first), dim3(GW_THREADS), b3_bytes, (hipStream_t)stream, T, (long)n);
    } else {
        hipLaunchKernelGGL(mlp_grad_w_kernel, dim3(first), dim3(GW_THREADS), lds_bytes, (hipStream_t)stream, T, (long)n);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(mlp_grad_reduce_kernel<4>, dim3(RED_BLOCKS), dim3(64 * RED_WAVES), 0,
                       (hipStream_t)stream, T, grad_out, norm_mask, norm_ws, norm_step, err);
    return hipGetLastError();
}

extern "C" int flyhip_mlp_reduce_blocks(void) { return RED_BLOCKS; }

// ---- the fused optimizer-step gradient: mlp_fused_step_kernel + the fixed-order reduction of its per-workgroup slabs ------
static int g_fused_grid_override = 0;    // test / tuning hook: fewer workgroups than CUs (each then walks more tiles)
extern "C" void flyhip_debug_set_fused_grid(int grid) { g_fused_grid_override = grid; }
extern "C" int flyhip_debug_get_fused_grid(void) { return g_fused_grid_override; }
static int fused_cus()
{
    static int cus[16] = {0};        // per DEVICE: a process may drive more than one
    int dev = 0;
    hipDeviceProp_t pr;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 256;
    if (!cus[dev]) cus[dev] = hipGetDeviceProperties(&pr, dev) == hipSuccess ? pr.multiProcessorCount : 256;
    return cus[dev];
}
static int fused_grid(int64_t n)
{
    int g = fused_cus();
    if (g_fused_grid_override > 0 && g_fused_grid_override < g) g = g_fused_grid_override;
    const long tiles = (n + BM - 1) / BM;
    return (int)(tiles < g ? tiles : g);
}

extern "C" int64_t flyhip_mlp_fused_workspace_floats(void)
{
    // one partial slab per workgroup; chunked layout: every layer's block padded to whole 1 KiB chunks
    return (int64_t)fused_cus() * (fs_pad256(FS_STRIDE1) + fs_pad256(FS_STRIDE2) + fs_pad256(FS_STRIDE3) + fs_pad256(FS_STRIDE4));
}

extern "C" hipError_t flyhip_launch_mlp_fused_grad(const float* P, const uint16_t* PB, const uint16_t* PTB, const float* x,
                                                   int64_t n, const float* action, const float* old_logp, const float* adv,
                                                   const float* target, const float* var, float inv_batch, float clip,
                                                   float* workspace, float* grad_out, const float* norm_mask, float* norm_ws,
                                                   int* norm_step, float* loss_part, float* const* dump, void* stream)
{
    const int grid = fused_grid(n);
    FusedDump d = {};
    // debug_dump: 8 pointers = the chain dump (tests); ONE pointer followed by NULL = a stamp buffer (tools/stamp_fused.py)
    const int mode = dump == nullptr ? 0 : (dump[1] == nullptr ? 2 : 1);
    if (mode == 1) { d.out = dump[0]; d.h1 = dump[1]; d.h2 = dump[2]; d.h3 = dump[3]; d.dz4 = dump[4]; d.dz3 = dump[5]; d.dz2 = dump[6]; d.dz1 = dump[7]; }
    if (mode == 2) d.out = dump[0];
    const void* fn = mode == 0 ? reinterpret_cast<const void*>(mlp_fused_step_kernel<0>)
                   : mode == 1 ? reinterpret_cast<const void*>(mlp_fused_step_kernel<1>)
                               : reinterpret_cast<const void*>(mlp_fused_step_kernel<2>);
    {           // (per launch: the attribute belongs to the CURRENT device)
        hipError_t ea = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, FS_LDS_BYTES);
        if (ea != hipSuccess) return ea;
    }
#define FS_LAUNCH(M_)                                                                                                             \
    hipLaunchKernelGGL(mlp_fused_step_kernel<M_>, dim3(grid), dim3(THREADS), FS_LDS_BYTES, (hipStream_t)stream, P, PB, PTB, x,     \
                       (long)n, action, old_logp, adv, target, var, inv_batch, clip, workspace, loss_part, d)
    if (mode == 0) FS_LAUNCH(0); else if (mode == 1) FS_LAUNCH(1); else FS_LAUNCH(2);
#undef FS_LAUNCH
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    // the slabs have the layout the reduction already sums: per layer `grid` blocks of N*KP + N floats
    GradWTable T;
    const int N[4] = {MLP_H1, MLP_H2, MLP_H3, MLP_OUT};
    const int KP[4] = {MLP_IN_PAD, MLP_H1, MLP_H2, MLP_H3};
    float* w = workspace;
    for (int l = 0; l < 4; ++l) {
        T.l[l].dz = nullptr; T.l[l].a = nullptr; T.l[l].partial = w;
        T.l[l].N = N[l]; T.l[l].Ka = KP[l]; T.l[l].KP = KP[l]; T.l[l].wgs = grid; T.l[l].first_block = 0; T.l[l].accumulate = 0;
        T.l[l].chunked = FS_SLAB_CHUNKED;
        const long stride = (long)N[l] * KP[l] + N[l];
        w += (long)grid * (FS_SLAB_CHUNKED ? fs_pad256(stride) : stride);
    }
    // (loads in flight per wave: 4 -> 15.8 us for the 256 slabs, 8 -> 17.2, 16 -> 58 (the 1024-thread block's register budget))
    hipLaunchKernelGGL(mlp_grad_reduce_kernel<4>, dim3(RED_BLOCKS), dim3(64 * RED_WAVES), 0, (hipStream_t)stream, T, grad_out,
                       norm_mask, norm_ws, norm_step, (const int*)nullptr);
    return hipGetLastError();
}

extern "C" hipError_t flyhip_launch_mlp_adam(float* P, float* PF, float* PT, const int* idx_f, const int* idx_t,
                                             const float* G, const float* mask, float* m,
                                             float* v, int* step, float lr, float beta1, float beta2, float eps,
                                             float max_norm, float grad_scale, float* norm_ws, int norm_ready,
                                             uint16_t* PB, uint16_t* PTB, const int* idx_fb, const int* idx_tb,
                                             int* step_out, const int* grad_invalid, uint16_t* PH, uint16_t* PTH,
                                             float* h2_scales, int h2_rescale, void* stream)
{
    int nparts = ADAM_BLOCKS;
    float part_scale = 1.0f;
    if (step_out) {                 // one launch: the kernel sums the gradient itself and ping-pongs the step counter
        part_scale = 1.0f;          // its sum is of the SCALED gradient
    } else if (norm_ready) {               // mlp_grad_w already left per-block sums of squares (unscaled) and advanced the step
        nparts = RED_BLOCKS;
        part_scale = grad_scale * grad_scale;
    } else {
        hipLaunchKernelGGL(mlp_adam_norm_kernel, dim3(ADAM_BLOCKS), dim3(ADAM_THREADS), 0, (hipStream_t)stream, G, mask,
                           grad_scale, norm_ws, step, grad_invalid);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(mlp_adam_apply_kernel, dim3(ADAM_BLOCKS), dim3(ADAM_THREADS), 0, (hipStream_t)stream, P, PF, PT,
                       idx_f, idx_t, G, mask, m, v, step, lr, beta1, beta2, eps, max_norm, grad_scale, norm_ws, nparts,
                       part_scale, PB, PTB, idx_fb, idx_tb, step_out, grad_invalid, PH, PTH, h2_scales, h2_rescale);
    return hipGetLastError();
}

extern "C" hipError_t flyhip_launch_mlp_h2_rescale(const float* P, const int* idx_fb, const int* idx_tb, uint16_t* PH, uint16_t* PTH,
                                                   float* h2_scales, void* stream)
{
    hipLaunchKernelGGL(mlp_h2_rescale_kernel, dim3(1), dim3(H2_RESCALE_THREADS), 0, (hipStream_t)stream, P, idx_fb, idx_tb, PH, PTH, h2_scales);
    return hipGetLastError();
}
