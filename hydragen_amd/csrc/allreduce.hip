// All-reduce(sum) of the tensor-parallel attention block output over peer-mapped device memory (xGMI between the
// GPUs of one node), for the two call sites of /root/reference/hydragen/tp.py:83-87 and :108-112, where the reference
// issues a NCCL all-reduce of [B, q_len, hidden] after the row-parallel o_proj / down_proj.
//
// Why not a ring: xGMI is a full mesh of point-to-point links (7 x ~153 GB/s per GPU), and a decode step's tensor is
// small (C5: 32 MiB, C2: 8 MiB).  Two-shot direct algorithm, every link busy in both shots, 2 * (N-1)/N of the
// tensor crossing each rank's links in total, no intermediate hops:
//   shot 1 (reduce-scatter)  rank r reads slice r of every rank's staged input straight from that rank's memory, sums
//                            in fp32, writes the result to its own `out` slice and to its `reduced` staging block;
//   shot 2 (all-gather)      rank r reads slice p of `reduced` from every peer p into its own `out`.
// Synchronisation: per-rank flag blocks in the same peer-mapped allocation, written with system-scope release
// stores by the peers and polled (relaxed, then one system-scope acquire) by the owner; flag values are the call's
// epoch, which lives in DEVICE memory (a captured graph replays with frozen kernel arguments).  Every spin is
// bounded: on timeout the kernel writes an error code into the rank's status word and finishes without hanging.
//
// The library allocates nothing: the caller owns one "shared block" per rank (hyd_allreduce_block_bytes) and hands
// in every rank's block as mapped in this process (hyd_ipc_* below are thin wrappers over hipIpc*MemHandle).
#include <cstring>
#include <type_traits>

#include "hyd_kernels.h"

namespace hyd {

namespace {

constexpr int kArMaxWorld = 8;
constexpr int kArFlagStride = 32;           // uint32 words per flag (128 B: one flag per cache line)
constexpr int kArFlagWords = 2 * kArMaxWorld * kArFlagStride;  // two phases x world flags
constexpr int kArLocalWords = 64;           // epoch, arrival counter, status (never read by peers)
constexpr size_t kArHeaderPad = 4096;       // flags + local words = 2304 B, padded
static_assert((kArFlagWords + kArLocalWords) * sizeof(uint32_t) <= kArHeaderPad, "header");
constexpr int kArSpinLog2Default = 27;      // polls per wait; one poll (s_sleep 8 + a flag load) measured at ~0.3 us: ~40 s

struct ArArgs {
    char* block[kArMaxWorld];  // every rank's shared block as mapped in this process; block[rank] is our own
    const void* in;
    void* out;
    int64_t count;             // elements
    int64_t slice;             // elements per rank slice (multiple of 8, slice * world >= count)
    size_t stage_bytes;        // bytes reserved for `staged input` inside a block
    int32_t rank, world, dtype;
    uint32_t spin_limit;       // polls per wait before a rank gives up (status word)
};

__device__ __forceinline__ uint32_t* flag_ptr(char* block, int phase, int from) {
    return reinterpret_cast<uint32_t*>(block) + (phase * kArMaxWorld + from) * kArFlagStride;
}
__device__ __forceinline__ uint32_t* local_ptr(char* block, int i) {
    return reinterpret_cast<uint32_t*>(block) + kArFlagWords + i;
}

// one lane waits until *flag >= epoch; returns false after spin_limit polls.  The bound counts polls, not clock
// ticks: a wave that the scheduler saves and restores (several processes sharing one device) can resume on another
// XCD, whose s_memtime has a different base -- a clock difference across that switch reads as an instant timeout.
__device__ __forceinline__ bool wait_flag(uint32_t* flag, uint32_t epoch, uint32_t spin_limit) {
    for (uint32_t n = 0; n < spin_limit; ++n) {
        if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) >= epoch) return true;
        __builtin_amdgcn_s_sleep(8);
    }
    return false;
}

template <typename T>
__device__ __forceinline__ void add8(float (&acc)[8], const u32x4& v) {
    using TR = Traits<T>;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        acc[2 * i] += TR::lo(v[i]);
        acc[2 * i + 1] += TR::hi(v[i]);
    }
}

// Stage the input where the peers can read it, and announce it (phase-0 flags) from the LAST workgroup to finish:
// every workgroup writes back its own XCD's L2 (system-scope release) before it takes its ticket, so when the last
// ticket is drawn the whole staged tensor is in memory -- the announcement does not lean on what the boundary between
// two kernels of one queue flushes.
__global__ __launch_bounds__(256) void ar_stage_kernel(const ArArgs a, int elt_bytes) {
    char* mine = a.block[a.rank];
    __shared__ uint32_t s_last;
    const size_t bytes = (size_t)a.count * elt_bytes;
    const u32x4* src = static_cast<const u32x4*>(a.in);
    u32x4* dst = reinterpret_cast<u32x4*>(mine + kArHeaderPad);
    const size_t n16 = bytes / 16;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) dst[i] = src[i];
    if (blockIdx.x == 0 && threadIdx.x < (bytes & 15)) {  // tail bytes (count is not a multiple of 16 bytes)
        mine[kArHeaderPad + n16 * 16 + threadIdx.x] = static_cast<const char*>(a.in)[n16 * 16 + threadIdx.x];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t t = __hip_atomic_fetch_add(local_ptr(mine, 3), 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        s_last = (t == gridDim.x - 1) ? 1u : 0u;
    }
    __syncthreads();
    if (s_last) {
        const uint32_t epoch = __hip_atomic_load(local_ptr(mine, 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
        if ((int)threadIdx.x < a.world) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
            __hip_atomic_store(flag_ptr(a.block[threadIdx.x], 0, a.rank), epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        if (threadIdx.x == 0) __hip_atomic_store(local_ptr(mine, 3), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void ar_reduce_kernel(const ArArgs a) {
    constexpr int E = sizeof(typename std::conditional<std::is_same<T, float>::value, float, uint16_t>::type);
    char* mine = a.block[a.rank];
    uint32_t* epoch_w = local_ptr(mine, 0);
    uint32_t* arrive_w = local_ptr(mine, 1);
    uint32_t* status_w = local_ptr(mine, 2);
    __shared__ uint32_t s_epoch, s_ok, s_last;
    const int tid = threadIdx.x;
    if (tid == 0) {
        // every rank runs the same sequence of calls: epochs agree without communication.  The block's local words are
        // only ever touched by agent-scope atomics (a plain store next to atomic adds on the same word is not
        // coherent across the XCDs' L2s).
        s_epoch = __hip_atomic_load(epoch_w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
        s_ok = 1;
    }
    __syncthreads();
    const uint32_t epoch = s_epoch;
    // ---- wait for every rank's staged input (announced by the stage kernels) ---------------------------------
    if (tid < a.world) {
        if (!wait_flag(flag_ptr(mine, 0, tid), epoch, a.spin_limit)) s_ok = 0;
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    const bool ok1 = s_ok != 0;

    // ---- shot 1: reduce slice `rank` over all ranks' staged inputs ------------------------------------------
    const int64_t s0 = (int64_t)a.rank * a.slice;
    const int64_t s1 = a.count < s0 + a.slice ? a.count : s0 + a.slice;
    const int64_t nvec = s1 > s0 ? (s1 - s0 + 7) / 8 : 0;  // 8-element groups (slice is a multiple of 8)
    char* red = mine + kArHeaderPad + a.stage_bytes;       // this rank's reduced slice, read by the peers in shot 2
    if (ok1) {
        for (int64_t i = (int64_t)blockIdx.x * 256 + tid; i < nvec; i += (int64_t)gridDim.x * 256) {
            const int64_t e0 = s0 + i * 8;
            if constexpr (std::is_same<T, float>::value) {
                f32x4 x0 = {0.f, 0.f, 0.f, 0.f}, x1 = x0;
                f32x4 w0[kArMaxWorld], w1[kArMaxWorld];
#pragma unroll
                for (int p = 0; p < kArMaxWorld; ++p)  // all peers' loads (2 x 16 B each) in flight together
                    if (p < a.world) {
                        const float* src = reinterpret_cast<const float*>(a.block[p] + kArHeaderPad) + e0;
                        w0[p] = *reinterpret_cast<const f32x4*>(src);
                        w1[p] = *reinterpret_cast<const f32x4*>(src + 4);
                    }
#pragma unroll
                for (int p = 0; p < kArMaxWorld; ++p)
                    if (p < a.world) {
                        x0 += w0[p];
                        x1 += w1[p];
                    }
                float* r = reinterpret_cast<float*>(red) + i * 8;
                *reinterpret_cast<f32x4*>(r) = x0;
                *reinterpret_cast<f32x4*>(r + 4) = x1;
                if (e0 + 8 <= a.count) {
                    float* o = static_cast<float*>(a.out) + e0;
                    *reinterpret_cast<f32x4*>(o) = x0;
                    *reinterpret_cast<f32x4*>(o + 4) = x1;
                } else {
                    for (int j = 0; e0 + j < a.count; ++j) static_cast<float*>(a.out)[e0 + j] = j < 4 ? x0[j] : x1[j - 4];
                }
            } else {
                using TR = Traits<T>;
                float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                u32x4 v[kArMaxWorld];
#pragma unroll
                for (int p = 0; p < kArMaxWorld; ++p)  // all peers' loads in flight together
                    if (p < a.world) v[p] = *reinterpret_cast<const u32x4*>(a.block[p] + kArHeaderPad + e0 * 2);
#pragma unroll
                for (int p = 0; p < kArMaxWorld; ++p)
                    if (p < a.world) add8<T>(acc, v[p]);
                const u32x4 pk = {TR::pack2(acc[0], acc[1]), TR::pack2(acc[2], acc[3]), TR::pack2(acc[4], acc[5]),
                                  TR::pack2(acc[6], acc[7])};
                *reinterpret_cast<u32x4*>(red + i * 16) = pk;
                if (e0 + 8 <= a.count) {
                    *reinterpret_cast<u32x4*>(static_cast<uint16_t*>(a.out) + e0) = pk;
                } else {
                    for (int j = 0; e0 + j < a.count; ++j)
                        static_cast<uint16_t*>(a.out)[e0 + j] = (uint16_t)(pk[j >> 1] >> (16 * (j & 1)));
                }
            }
        }
    }
    // ---- the last workgroup to finish shot 1 announces the reduced slice (phase-1 flags) --------------------
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    __syncthreads();
    if (tid == 0) {
        const uint32_t t = __hip_atomic_fetch_add(arrive_w, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        s_last = (t == gridDim.x - 1) ? 1u : 0u;
    }
    __syncthreads();
    if (s_last && tid < a.world) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
        __hip_atomic_store(flag_ptr(a.block[tid], 1, a.rank), epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (s_last && tid == 0) {
        // ready for the next call (every workgroup of this launch has arrived and read the epoch)
        __hip_atomic_store(arrive_w, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(epoch_w, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (!ok1 && tid == 0) __hip_atomic_store(status_w, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // ---- shot 2: gather every peer's reduced slice ----------------------------------------------------------
    // The peers are dealt over the workgroups (the launch's grid is a multiple of world - 1): all of a rank's inbound
    // links carry data at the same time, and a workgroup waits for ITS peer only.  A lane keeps kGather 16-byte loads in
    // flight before its first store: grid x 256 lanes x 16 B x kGather per rank (Little's law against the
    // ~3 us a remote read takes).
    if (a.world > 1) {
        constexpr int kGather = 8;
        const int npeer = a.world - 1;
        const int q = 1 + (int)(blockIdx.x % npeer);
        const int p = (a.rank + q) % a.world;  // a different first peer on every rank
        if (tid == 0) s_ok = wait_flag(flag_ptr(mine, 1, p), epoch, a.spin_limit) ? 1u : 0u;
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
        const bool ok2 = s_ok != 0;
        if (!ok2 && tid == 0) __hip_atomic_store(status_w, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int64_t p0 = (int64_t)p * a.slice, p1 = a.count < p0 + a.slice ? a.count : p0 + a.slice;
        const int64_t nbytes = p1 > p0 ? (p1 - p0) * E : 0;  // the slice starts 16-byte aligned (slice % 8 == 0)
        const int64_t n16 = nbytes / 16;
        const char* src = a.block[p] + kArHeaderPad + a.stage_bytes;
        char* dst = static_cast<char*>(a.out) + p0 * E;
        if (ok2) {
            const int64_t lanes = (int64_t)((gridDim.x - (q - 1) + npeer - 1) / npeer) * 256;  // lanes serving this peer
            for (int64_t i = (int64_t)(blockIdx.x / npeer) * 256 + tid; i < n16; i += lanes * kGather) {
                u32x4 v[kGather];
#pragma unroll
                for (int u = 0; u < kGather; ++u) {
                    const int64_t j = i + u * lanes;
                    v[u] = *reinterpret_cast<const u32x4*>(src + (j < n16 ? j : n16 - 1) * 16);
                }
#pragma unroll
                for (int u = 0; u < kGather; ++u) {
                    const int64_t j = i + u * lanes;
                    if (j < n16) *reinterpret_cast<u32x4*>(dst + j * 16) = v[u];
                }
            }
            if (blockIdx.x / npeer == 0 && tid < (nbytes & 15)) dst[n16 * 16 + tid] = src[n16 * 16 + tid];  // count % 8 elements
        }
    }
}

}  // namespace

size_t allreduce_block_bytes(int world, size_t max_bytes) {
    const size_t stage = (max_bytes + 255) / 256 * 256;
    const size_t red = ((max_bytes + world - 1) / world + 16 * 8 + 255) / 256 * 256;
    return kArHeaderPad + stage + red;
}

int launch_allreduce(char* const* blocks, size_t block_bytes, const void* in, void* out, int64_t count, int dtype,
                     int rank, int world, size_t max_bytes, int timeout_log2_polls, hipStream_t s) {
    ArArgs a;
    memset(&a, 0, sizeof(a));
    for (int i = 0; i < world; ++i) a.block[i] = blocks[i];
    a.in = in;
    a.out = out;
    a.count = count;
    const int elt = dtype == HYD_F32 ? 4 : 2;
    const int64_t per = (count + world - 1) / world;
    a.slice = (per + 7) / 8 * 8;
    a.stage_bytes = (max_bytes + 255) / 256 * 256;
    a.rank = rank;
    a.world = world;
    a.dtype = dtype;
    a.spin_limit = 1u << (timeout_log2_polls > 0 ? timeout_log2_polls : kArSpinLog2Default);
    (void)block_bytes;
    const size_t bytes = (size_t)count * elt;
    // The kernel waits on its peers, so every workgroup must be resident while it spins -- also when all 8 ranks share
    // one device, as in the single-GPU tests: 8 x 133 four-wave workgroups of <= 64 registers are half of what 256 CUs hold.
    // A multiple of world - 1: shot 2 deals the peers over the workgroups.
    const int want = (int)(bytes >= (8u << 20) ? 128 : bytes >= (1u << 20) ? 32 : 8);
    const int npeer = world > 1 ? world - 1 : 1;
    const int grid = (want + npeer - 1) / npeer * npeer;
    hipLaunchKernelGGL(ar_stage_kernel, dim3(grid), dim3(256), 0, s, a, elt);
    if (dtype == HYD_F32) hipLaunchKernelGGL((ar_reduce_kernel<float>), dim3(grid), dim3(256), 0, s, a);
    else if (dtype == HYD_BF16) hipLaunchKernelGGL((ar_reduce_kernel<BF16>), dim3(grid), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((ar_reduce_kernel<F16>), dim3(grid), dim3(256), 0, s, a);
    return (int)hipGetLastError();
}

}  // namespace hyd
