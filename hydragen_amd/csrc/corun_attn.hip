// Co-run kernel: ONE launch whose persistent workgroups (one per CU, one wave per SIMD) run the decode step's two
// passes SIDE BY SIDE -- the prefix pass (MFMA-bound, prefix_unit_w64.h) on a part of the chip while the suffix pass
// (HBM-bound, suffix_stream.h) streams on the rest -- instead of one after the other as
// /root/reference/hydragen/attention.py:250-352 issues them.  Measured motivation (MI355X, C2): the prefix pass alone
// on all 256 CUs is clock-throttled (1.8 GHz, 46.7 us) and leaves HBM idle, the suffix pass leaves the matrix cores
// idle; on two HIP streams the passes do overlap (suffix 351 us beside the prefix pass vs 347 us alone) but the
// fork + join costs ~25 us per step, more than the overlap returns.  Inside one launch there is no fork and no join.
//
// Work distribution: two queues in a caller-provided, zeroed 1 KiB block.  Prefix units (one per (head, row block))
// are dealt per XCD (8 counters: the row blocks that share a head's K/V stay on one L2; an XCD that runs dry steals
// from the next); suffix items (suffix_stream.h) come from one counter.  A workgroup starts in the role its slot
// gives it (np_of8 of every 8 workgroups of an XCD start on prefix units) and, when its queue is empty, moves to the
// other one -- so any split of the chip between the roles is only a starting point.  Both roles write fp32 partials
// (normalised out + natural-log LSE); combine.hip merges them after the launch.
#include "prefix_unit_w64.h"
#include "suffix_stream.h"
#include "suffix_gqa_stream.h"

namespace hyd {

constexpr int kCorunQueueWords = 256;  // 1 KiB: prefix counter x at word 16 x (x < 8), suffix counter at word 128

struct CorunKArgs {
    PrefixArgs pa;
    SuffixArgs sa;
    StreamGeom g;
    unsigned* queue;
    int32_t np_of8;
    int32_t pad_;
};
typedef const __attribute__((address_space(4))) CorunKArgs* corun_kargs_p;

// Both roles are REAL function calls (noinline), not inlined into the kernel: the prefix unit alone fills the scalar and
// the vector register file and owns a[0:191] through its asm statements; with anything else inlined next to it hipcc
// parks values in exactly those AGPRs (seen in the assembly: v_accvgpr_write a128 inside the prefix loop).  As callees
// the roles get their own register allocation (tests/test_build_quality.py checks the prefix role's); they take no
// arguments beyond the LDS pointers and read their block of the kernel arguments from the kernel-argument segment.
typedef __attribute__((address_space(3))) char* lds_char_p;
typedef __attribute__((address_space(3))) unsigned* lds_u32_p;

template <int D, int KG>
constexpr int corun_prefix_lds() { return (KG == 2 ? 2 : 1) * 256 * (D * 2) + 4 * 2 * 128 * (int)sizeof(float); }

template <typename T, int D, int CKI, int NBUF>
__device__ __attribute__((noinline)) void corun_suffix_role() {
    corun_kargs_p p = (corun_kargs_p)__builtin_amdgcn_kernarg_segment_ptr();
    const SuffixArgs sa = *(const SuffixArgs*)&p->sa;
    const StreamGeom g = *(const StreamGeom*)&p->g;
    const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    suffix_stream_wave<T, D, CKI, NBUF, true>(sa, g, p->queue + 128, wave * (unsigned)stream_wave_lds_bytes<CKI, NBUF>());  // LDS base 0
}

// Prefix role: units of this XCD's queue first (the row blocks that share a head's K/V stay on one L2), then the others'.
// The kernel has no static LDS, so its dynamic LDS starts at LDS address 0 (the kernel traps if it ever does not): the
// callee addresses the rings with compile-time constants exactly as the stand-alone prefix kernel does, instead of
// carrying a base pointer in a vector register.  (LDS address 0 is a valid address: the null pointer of that address
// space is ~0.)  The two unit-index words sit behind the prefix pass's own region.
template <typename T, int D, int KG>
__device__ __attribute__((noinline)) void corun_prefix_role() {
    char* smem = (char*)(lds_char_p)(uintptr_t)0u;
    lds_u32_p s_unit = (lds_u32_p)(uintptr_t)(unsigned)corun_prefix_lds<D, KG>();
    const int xcd = blockIdx.x & 7;
    int par = 0;
#pragma unroll 1
    for (int t = 0; t < 8; ++t) {
        const int xq = (xcd + t) & 7;
#pragma unroll 1
        for (;;) {
            corun_kargs_p p = (corun_kargs_p)__builtin_amdgcn_kernarg_segment_ptr();
            asm volatile("" : "+s"(p));  // re-read per unit: nothing of the argument block stays live across units
            const int vgrid = p->pa.vgrid;
            if (threadIdx.x == 0) s_unit[par] = __hip_atomic_fetch_add(p->queue + 16 * xq, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            const int j = (int)s_unit[par];
            par ^= 1;  // the next pull writes the other word: no second barrier before it
            const int vb = j * 8 + xq;
            if (vb >= vgrid) break;
            const PrefixArgs pa = *(const PrefixArgs*)&p->pa;
            prefix_unit_w64<T, D, false, KG, 0>(pa, vb, vgrid, smem);
            __syncthreads();  // the unit's LDS merge buffers are the next unit's rings
        }
    }
}

template <typename T, int D, int KG, int CKI, int NBUF>
__global__ __launch_bounds__(256) void decode_corun_kernel(const CorunKArgs ka_unused) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if ((unsigned)(uintptr_t)(lds_char_p)smem != 0u) __builtin_trap();
    corun_kargs_p kp = (corun_kargs_p)__builtin_amdgcn_kernarg_segment_ptr();
    const bool prefix_first = (int)((blockIdx.x >> 3) & 7) < kp->np_of8;
#pragma unroll 1
    for (int ph = 0; ph < 2; ++ph) {
        if ((ph == 0) == prefix_first) corun_prefix_role<T, D, KG>();
        else corun_suffix_role<T, D, CKI, NBUF>();
        __syncthreads();
    }
}

// Development kernel (ablation builds): the streaming role alone, as a drop-in for suffix_attn_kernel (R = 1).
template <typename T, int D, int CKI, int NBUF>
__global__ __launch_bounds__(256) void suffix_stream_kernel(const SuffixArgs sa, const StreamGeom g, unsigned* queue) {
    extern __shared__ __attribute__((aligned(16))) char smem[];  // the four waves' rings (one workgroup per CU)
    const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    suffix_stream_wave<T, D, CKI, NBUF, false>(sa, g, queue + 128,
                                               (unsigned)(uintptr_t)(lds_char_p)smem + wave * (unsigned)stream_wave_lds_bytes<CKI, NBUF>());
}

StreamGeom stream_geom(const SuffixArgs& a, int upi) {
    StreamGeom g;
    g.upi = upi;
    g.nbr = (a.B + upi - 1) / upi;
    g.n_items = g.nbr * 4 * ((a.Hkv + 3) / 4);
    return g;
}

bool corun_eligible(const PrefixArgs& pa, const SuffixArgs& sa, int D, bool causal) {
    return D == 128 && !causal && sa.rows == 1 && sa.nq == 1 && sa.g == 1 && pa.nsplit == 1 && pa.wg_rows == 128 && !pa.cu_k && !pa.cu_q && sa.kv_len > 0 &&
           (int64_t)sa.kv_len * (sa.k_ts > sa.v_ts ? sa.k_ts : sa.v_ts) * 2 < ((int64_t)1 << 31);
}

template <typename T, int KG>
static int launch_corun_t(const PrefixArgs& pa, const SuffixArgs& sa, unsigned* queue, int grid, int np_of8, int upi, hipStream_t s) {
    constexpr int D = 128;
    constexpr int CKI = 8, NBUF = 2;
    constexpr size_t lds_p = corun_prefix_lds<D, KG>() + 16, lds_s = 4 * stream_wave_lds_bytes<CKI, NBUF>();
    constexpr size_t lds = lds_p > lds_s ? lds_p : lds_s;
    static_assert(lds <= 160 * 1024, "LDS");
    auto kern = decode_corun_kernel<T, D, KG, CKI, NBUF>;
    static const hipError_t attr_rc = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (attr_rc != hipSuccess) return (int)attr_rc;
    hipError_t e = hipMemsetAsync(queue, 0, kCorunQueueWords * sizeof(unsigned), s);
    if (e != hipSuccess) return (int)e;
    CorunKArgs ka;
    ka.pa = pa;
    ka.sa = sa;
    ka.g = stream_geom(sa, upi);
    ka.queue = queue;
    ka.np_of8 = np_of8;
    ka.pad_ = 0;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, ka);
    return (int)hipGetLastError();
}

int launch_corun(const PrefixArgs& pa, const SuffixArgs& sa, int dtype, unsigned* queue, int grid, int np_of8, int upi, hipStream_t s) {
    if (pa.wg_rows != 128) return (int)hipErrorInvalidValue;  // corun_eligible
    return dtype == HYD_F16 ? launch_corun_t<F16, 2>(pa, sa, queue, grid, np_of8, upi, s)
                            : launch_corun_t<BF16, 2>(pa, sa, queue, grid, np_of8, upi, s);
}

// ---- the matrix-core suffix pass as persistent one-wave workgroups (suffix_gqa_stream.h) ---------------------------------------
template <typename T, int D, int MODE>
__global__ __launch_bounds__(64) void suffix_gqa_stream_kernel(const SuffixArgs a, const GqaStreamGeom g,
                                                                                                         unsigned* queue) {
    __shared__ __attribute__((aligned(1024))) char vtiles[2][32 * D * 2];
    gqa_stream_wave<T, D, MODE>(a, g, queue, (unsigned)(uintptr_t)(lds_char_p)vtiles[0]);
}

GqaStreamGeom gqa_stream_geom(const SuffixArgs& a, int upi) {
    GqaStreamGeom g;
    g.upi = upi < 1 ? 1 : upi > 64 ? 64 : upi;
    g.nbr = (a.B + g.upi - 1) / g.upi;
    g.chunks = (a.rows + 15) / 16;
    g.n_items = g.nbr * g.chunks * a.Hkv;
    return g;
}

bool gqa_stream_eligible(const SuffixArgs& a, int D) {
    const int64_t span = (int64_t)a.kv_len * (a.k_ts > a.v_ts ? a.k_ts : a.v_ts) * 2;
    const int64_t qbytes = (int64_t)a.B * a.nq * a.Hq * D * 2;
    return D == 128 && a.kv_len > 0 && span < ((int64_t)1 << 31) && qbytes < ((int64_t)1 << 32) - 4096;
}

// queue: one zeroed counter word the launch may use (the caller's workspace)
int launch_suffix_gqa_stream(const SuffixArgs& a, int dtype, unsigned* queue, int waves, int upi, hipStream_t s) {
    hipError_t e = hipMemsetAsync(queue, 0, 64, s);
    if (e != hipSuccess) return (int)e;
    const GqaStreamGeom g = gqa_stream_geom(a, upi);
    if (dtype == HYD_F16) hipLaunchKernelGGL((suffix_gqa_stream_kernel<F16, 128, 0>), dim3(waves), dim3(64), 0, s, a, g, queue);
    else hipLaunchKernelGGL((suffix_gqa_stream_kernel<BF16, 128, 0>), dim3(waves), dim3(64), 0, s, a, g, queue);
    return (int)hipGetLastError();
}

#ifdef HYD_ABLATION_BUILD
template <typename T, int CKI, int NBUF>
static int launch_stream_t(const SuffixArgs& sa, unsigned* queue, int grid, int upi, hipStream_t s) {
    constexpr size_t lds = 4 * stream_wave_lds_bytes<CKI, NBUF>() > 100 * 1024 ? 4 * stream_wave_lds_bytes<CKI, NBUF>() : 100 * 1024;
    static_assert(lds <= 160 * 1024, "LDS");
    auto kern = suffix_stream_kernel<T, 128, CKI, NBUF>;
    static const hipError_t attr_rc = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (attr_rc != hipSuccess) return (int)attr_rc;
    hipError_t e = hipMemsetAsync(queue, 0, kCorunQueueWords * sizeof(unsigned), s);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, sa, stream_geom(sa, upi), queue);
    return (int)hipGetLastError();
}

int launch_suffix_stream_dev(const SuffixArgs& sa, int dtype, int grid, int upi, int nbuf, hipStream_t s) {
    static unsigned* queue = [] {
        void* p = nullptr;
        (void)hipMalloc(&p, kCorunQueueWords * sizeof(unsigned));
        return static_cast<unsigned*>(p);
    }();
    if (nbuf == 100) return launch_suffix_gqa_stream(sa, dtype, queue, grid, upi, s);  // matrix-core persistent form, `grid` one-wave workgroups
    if (dtype == HYD_F16) return launch_stream_t<F16, 8, 2>(sa, queue, grid, upi, s);
    switch (nbuf) {
        case 3: return launch_stream_t<BF16, 4, 3>(sa, queue, grid, upi, s);
        case 4: return launch_stream_t<BF16, 4, 4>(sa, queue, grid, upi, s);
        case 8: return launch_stream_t<BF16, 2, 8>(sa, queue, grid, upi, s);
        default: return launch_stream_t<BF16, 8, 2>(sa, queue, grid, upi, s);
    }
}
#endif

}  // namespace hyd
