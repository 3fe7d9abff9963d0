// mlp_fused_h2.hip — translation unit of the fused optimizer-step gradient in the fp16x2 arithmetic (mlp_fused_h2.inc): the
// persistent kernel, its slab reduction with the scale bookkeeping, and their launcher.  The bf16x3 kernel it stands beside lives
// in mlp_mfma.hip (mlp_fused_step.inc), whose layout constants, slab addressing and LDS swizzle this unit shares.
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
#include "mlp_fused_step.inc"       // constants and helpers only: mlp_fused_step_kernel is instantiated in mlp_mfma.hip
#include "mlp_fused_h2.inc"

}  // namespace

extern "C" int flyhip_debug_get_fused_grid(void);       // mlp_mfma.hip: the test hook that shrinks the grid of both fused kernels

static int h2_cus()
{
    int dev = 0;
    hipDeviceProp_t pr;
    static int cus[16] = {0};
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 256;
    if (!cus[dev]) cus[dev] = hipGetDeviceProperties(&pr, dev) == hipSuccess ? pr.multiProcessorCount : 256;
    return cus[dev];
}

static int64_t h2_slab_floats() { return fs_pad256(FS_STRIDE1) + fs_pad256(FS_STRIDE2) + fs_pad256(FS_STRIDE3) + fs_pad256(FS_STRIDE4); }

// one partial slab per workgroup (chunked layout) + eight class maxima per workgroup behind them
// (+ 8 floats: the launch's unscale factors, left by workgroup 0 for the reduction)
extern "C" int64_t flyhip_mlp_fused_h2_workspace_floats(void) { return (int64_t)h2_cus() * (h2_slab_floats() + H2_NACT_CLASSES) + 8; }

extern "C" hipError_t flyhip_launch_mlp_fused_grad_h2(const float* P, const uint16_t* PH, const uint16_t* PTH, float* fsc, int* ovf,
                                                      int freeze, const float* x, int64_t n, const float* action,
                                                      const float* old_logp, const float* adv, const float* target, const float* var,
                                                      float inv_batch, float clip, float* workspace, float* grad_out,
                                                      const float* norm_mask, float* norm_ws, int* norm_step, float* loss_part,
                                                      float* const* dump, void* stream)
{
    const int cus = h2_cus();
    int grid = cus;
    const int ovr = flyhip_debug_get_fused_grid();
    if (ovr > 0 && ovr < grid) grid = ovr;
    const long tiles = (n + BM - 1) / BM;
    if (tiles + 4096 >= (1L << 31)) return hipErrorInvalidValue;        // the kernel counts tiles in 32 bits
    if (tiles < grid) grid = (int)tiles;
    FusedDump d = {};
    const int mode = dump == nullptr ? 0 : (dump[1] == nullptr ? 2 : 1);
    if (mode == 1) { d.out = dump[0]; d.h1 = dump[1]; d.h2 = dump[2]; d.h3 = dump[3]; d.dz4 = dump[4]; d.dz3 = dump[5]; d.dz2 = dump[6]; d.dz1 = dump[7]; }
    if (mode == 2) d.out = dump[0];
    const void* fn = mode == 0 ? reinterpret_cast<const void*>(mlp_fused_step_h2_kernel<0>)
                   : mode == 1 ? reinterpret_cast<const void*>(mlp_fused_step_h2_kernel<1>)
                               : reinterpret_cast<const void*>(mlp_fused_step_h2_kernel<2>);
    // (set on every launch: cheap, and right on whichever device is current)
    hipError_t ea = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, H2_LDS_BYTES);
    if (ea != hipSuccess) return ea;
    float* wsmax = workspace + (int64_t)cus * h2_slab_floats();
#define H2_LAUNCH(M_)                                                                                                             \
    hipLaunchKernelGGL(mlp_fused_step_h2_kernel<M_>, dim3(grid), dim3(THREADS), H2_LDS_BYTES, (hipStream_t)stream, P, PH, PTH,     \
                       (const float*)fsc, x, (long)n, action, old_logp, adv, target, var, inv_batch, clip, workspace, wsmax,     \
                       loss_part, d)
    if (mode == 0) H2_LAUNCH(0); else if (mode == 1) H2_LAUNCH(1); else H2_LAUNCH(2);
#undef H2_LAUNCH
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(mlp_grad_reduce_h2_kernel, dim3(H2_RED_BLOCKS), dim3(64 * H2_RED_WAVES), 0, (hipStream_t)stream,
                       (const float*)workspace, grid, grad_out, norm_mask, norm_ws, norm_step, (const float*)wsmax, fsc, ovf, freeze);
    return hipGetLastError();
}
