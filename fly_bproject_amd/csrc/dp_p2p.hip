// dp_p2p.hip — one-shot peer-to-peer all-reduce of the packed policy gradient (297 KB) for the data-parallel ranks
// of one node (at this size a collective is latency-bound; every MI355X has a direct xGMI link to
// each of the 7 others, so "everybody reads everybody" in ONE step beats a ring's 2(N-1) hops).  It stands where
// the reference's single-process update has nothing (ppo.py:196-199); torch.distributed.all_reduce (RCCL) remains
// the default and the fallback.
//
// Each rank owns a FINE-GRAINED (hipDeviceMallocFinegrained: coherent across devices, never cached stale) window
//     pub[2][n] floats | flag[2] u32 | done[2] u32 | pad
// exported through hipIpc and opened by every peer.  One launch per optimizer step, epoch e (parity p = e & 1):
//   1. every workgroup copies its slice of the local gradient into pub[p]          (16-byte stores)
//   2. system-scope fence; the LAST workgroup to finish (a device-local ticket) publishes flag[p] = e
//   3. `world` lanes poll the peers' flag[p] (system-scope loads, bounded spin -> *err = 1, never a hang)
//   4. every workgroup sums the peers' pub[p] slices IN RANK ORDER into the local gradient: the same additions in
//      the same order on every rank, so the replicas stay bit-identical.
// Two parities suffice: a rank can only publish epoch e + 2 after it has finished e + 1, which needs every peer's
// flag for e + 1, which a peer sets only after it has finished reading epoch e.
// All workgroups of the launch must be co-resident (they wait for each other through the ticket): the grid is
// 73 x 256 threads, far below one workgroup per CU.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "flyhip.h"

namespace {

constexpr int P2P_THREADS = 256;
constexpr int P2P_MAX_WORLD = 16;

struct P2PWindow {          // layout of one rank's window, offsets in bytes from its base
    static __host__ __device__ size_t pub(int parity, long n) { return (size_t)parity * n * sizeof(float); }
    static __host__ __device__ size_t flag(int parity, long n) { return 2 * (size_t)n * sizeof(float) + 4 * parity; }
    static __host__ __device__ size_t done(int parity, long n) { return 2 * (size_t)n * sizeof(float) + 8 + 4 * parity; }
    static __host__ __device__ size_t bytes(long n) { return 2 * (size_t)n * sizeof(float) + 64; }
};

struct P2PTable { char* base[P2P_MAX_WORLD]; };

__global__ __launch_bounds__(P2P_THREADS) void dp_allreduce_p2p_kernel(float* __restrict__ G, long n, P2PTable T, int rank,
                                                                       int world, unsigned epoch, int* __restrict__ err,
                                                                       long fail_slot, int poll_budget)
{
    __shared__ int ok;
    const int tid = threadIdx.x, parity = (int)(epoch & 1u);
    const long q = (long)blockIdx.x * P2P_THREADS + tid;          // float4 index
    const long n4 = n / 4;
    char* mine = T.base[rank];
    if (q < n4) reinterpret_cast<float4*>(mine + P2PWindow::pub(parity, n))[q] = reinterpret_cast<const float4*>(G)[q];
    __threadfence_system();                                       // this thread's slice is visible to every device ...
    __syncthreads();                                              // ... and so is the whole workgroup's
    if (tid == 0) {
        unsigned* done = reinterpret_cast<unsigned*>(mine + P2PWindow::done(parity, n));
        const unsigned t = __hip_atomic_fetch_add(done, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (t == gridDim.x - 1) {                                 // the last workgroup of this rank publishes the epoch
            __hip_atomic_store(done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __threadfence_system();
            __hip_atomic_store(reinterpret_cast<unsigned*>(mine + P2PWindow::flag(parity, n)), epoch, __ATOMIC_RELEASE,
                               __HIP_MEMORY_SCOPE_SYSTEM);
        }
        ok = 1;
    }
    __syncthreads();
    if (tid < world) {                                            // one lane per peer polls that peer's flag
        const unsigned* f = reinterpret_cast<const unsigned*>(T.base[tid] + P2PWindow::flag(parity, n));
        bool seen = false;
        for (int poll = 0; poll < poll_budget; ++poll) {
            if (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) == epoch) { seen = true; break; }
            __builtin_amdgcn_s_sleep(8);
        }
        if (!seen) { ok = 0; atomicMax(err, 1); }
    }
    __syncthreads();
    if (!ok) {
        // a peer never published: G is NOT the sum.  Mark it invalid so that mlp_adam_* refuse it on this rank
        // (fail closed); the host finds *err set at its next check and stops the run.
        if (fail_slot >= 0 && q == fail_slot / 4) G[fail_slot] = 1.0f;
        return;
    }
    if (q >= n4) return;
    __threadfence_system();
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int r = 0; r < world; ++r) {                             // rank order: identical on every rank
        const float4 v = reinterpret_cast<const float4*>(T.base[r] + P2PWindow::pub(parity, n))[q];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    reinterpret_cast<float4*>(G)[q] = s;
}

}  // namespace

extern "C" hipError_t flyhip_p2p_alloc(int64_t n_floats, void** out)
{
    void* p = nullptr;
    hipError_t e = hipExtMallocWithFlags(&p, P2PWindow::bytes(n_floats), hipDeviceMallocFinegrained);
    if (e != hipSuccess) return e;
    e = hipMemset(p, 0, P2PWindow::bytes(n_floats));
    if (e != hipSuccess) { (void)hipFree(p); return e; }
    // epoch 0 is never used: flags start at 0, the first epoch is 1
    *out = p;
    return hipDeviceSynchronize();
}

extern "C" hipError_t flyhip_launch_p2p_allreduce(float* G, int64_t n, void* const* bases, int rank, int world,
                                                  uint32_t epoch, int* err, int64_t fail_slot, void* stream)
{
    // how long a rank waits for a late peer (checkpoint save, first-use lazy loading, GC): 2^24 polls with s_sleep
    // are several seconds; FLY_P2P_POLL_LOG2 overrides it
    // (read per launch: a getenv costs nothing next to a launch, and the tests lower it after the start-up self-test)
    int budget;
    {
        const char* e = getenv("FLY_P2P_POLL_LOG2");
        int lg = e ? atoi(e) : 24;
        if (lg < 4) lg = 4;
        if (lg > 30) lg = 30;
        budget = 1 << lg;
    }
    P2PTable T;
    for (int r = 0; r < P2P_MAX_WORLD; ++r) T.base[r] = r < world ? static_cast<char*>(bases[r]) : nullptr;
    const long n4 = n / 4;
    const unsigned grid = (unsigned)((n4 + P2P_THREADS - 1) / P2P_THREADS);
    hipLaunchKernelGGL(dp_allreduce_p2p_kernel, dim3(grid), dim3(P2P_THREADS), 0, (hipStream_t)stream, G, (long)n, T, rank, world,
                       epoch, err, (long)fail_slot, budget);
    return hipGetLastError();
}
