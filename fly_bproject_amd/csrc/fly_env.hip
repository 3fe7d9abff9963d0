// fly_env.hip — gfx950 kernels for the vectorised Fly environment step.
//
// Replaces, per launch, the torch op chains and Isaac Gym calls of the reference's Fly.step
// (fly.py:624-681): K1 action scale (:626-657), K2 masked reset (:446-480), K3 physics
// (:482-485, build-defined FlyDyn), K4 observation pack (:771-805),
// progress += 1 (:678), K5 reward/done pack (:685-768).
//
// Mapping (CDNA4, 64-wide waves): 8 lanes own one env, so a wave carries 8 envs and a
// 256-thread workgroup 32.  Inside an env's 8-lane half of a DPP row
//     lanes 0..5   one leg each: 3 PD joints, leg kinematics, tip contact        (contact pass A)
//     lanes 0..4   ALSO one abdomen contact point each                          (contact pass B)
//     lanes 6..7   idle in the contact passes (they still hold the replicated root state)
// The kernel is VALU-issue-bound (15 substeps x ~330 instructions per wave), so what counts is
// instructions per env: the contact code runs twice per substep, but a wave now advances 8 envs
// instead of 4 -- 1.5x fewer instructions per env-substep than one point per lane on 16 lanes
// (11 of 16 lanes busy, root integration replicated 16x).
// The body wrench is an all-reduce over the 8 lanes with three DPP adds per component
// (quad_perm xor1, quad_perm xor2, row_half_mirror); the tree is symmetric so all 8 lanes end
// with bit-identical sums and integrate the replicated root state identically.
// All role differences are selects, not branches: a wave never diverges inside the substep loop.
// Substeps run in registers; HBM is touched once on the way in and once on the way out.
// Observation rows (row-major [N][73], a GEMM operand for the policy) are assembled in an LDS
// tile and written as one contiguous 32x73 block per workgroup with 16-byte stores.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "flyhip.h"

namespace {

#include "fly_body.inc"

template <int PH>
__global__ __launch_bounds__(BLOCK) void fly_kernel(const FlyConfig* __restrict__ c,
                                                    const float* __restrict__ actions, FlyBuffers b)
{
    __shared__ __attribute__((aligned(16))) float obs_tile[ENVS_PER_BLOCK * FLY_NUM_OBS];
    FlyRegs st;
    fly_load<PH>(st, c, b, blockIdx.x);
    fly_body<PH>(c, actions, b, obs_tile, blockIdx.x, st);
}

inline int grid_for(int n) { return (n + ENVS_PER_BLOCK - 1) / ENVS_PER_BLOCK; }

}  // namespace

// Launchers used by flyhip_abi.hip
#define FLY_LAUNCH(PH)                                                                            \
    hipLaunchKernelGGL((fly_kernel<PH>), dim3(grid_for(n)), dim3(BLOCK), 0, (hipStream_t)stream, \
                       dcfg, actions, *b)

extern "C" hipError_t flyhip_launch_env(int phases, const FlyConfig* dcfg, int n, const float* actions,
                                        const FlyBuffers* b, void* stream)
{
    switch (phases) {
    case PH_SCALE | PH_RESET | PH_INTEGRATE | PH_OBS | PH_PROGRESS | PH_REWARD:
        FLY_LAUNCH(PH_SCALE | PH_RESET | PH_INTEGRATE | PH_OBS | PH_PROGRESS | PH_REWARD); break;
    case PH_SCALE: FLY_LAUNCH(PH_SCALE); break;
    case PH_RESET: FLY_LAUNCH(PH_RESET); break;
    case PH_INTEGRATE: FLY_LAUNCH(PH_INTEGRATE); break;
    case PH_OBS: FLY_LAUNCH(PH_OBS); break;
    case PH_REWARD: FLY_LAUNCH(PH_REWARD); break;
    case PH_REWARD | PH_PROGRESS: FLY_LAUNCH(PH_REWARD | PH_PROGRESS); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
