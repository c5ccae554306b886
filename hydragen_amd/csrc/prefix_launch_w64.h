#pragma once
// Prefix pass kernel template + launch helpers, shared by the two translation units that instantiate it (one per storage
// dtype: prefix_attn_w64.hip = bf16 + the dtype dispatch, prefix_attn_w64_f16.hip = fp16 -- two files only so that the 30
// instantiations compile in parallel).  The unit body lives in prefix_unit_w64.h.
#include "prefix_unit_w64.h"

namespace hyd {

// One workgroup per unit (grid == a.vgrid: PERSIST = false, the unit body runs straight through), or -- when the caller
// asks for fewer workgroups than units (hyd_decode_params.shared_max_workgroups: the two-stream form keeps the prefix pass
// to a part of the chip) -- persistent workgroups that walk the units with a stride of the grid.  Two instantiations on
// purpose: wrapped in the unit loop, the body's loop-invariant scalars are hoisted and spilled to VGPR lanes, which
// costs the one-unit launch 1-3 % (A/B against the round-2 library, tests/probes/ab_r2_prefix.py).
// The kernel's 248 bytes of arguments are fetched in one scalar-cache miss (hyd_common.h, warm_kernargs_256).
__device__ __forceinline__ void warm_kernargs() {
    static_assert(sizeof(PrefixArgs) <= 256, "four scalar-cache lines");
    warm_kernargs_256();
}

template <typename T, int D, bool CAUSAL, int KG, int ABL = 0, bool PERSIST = false>
__global__ __launch_bounds__(256) void prefix_attn_w64_kernel(const PrefixArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    warm_kernargs();
    if constexpr (!PERSIST) {
        prefix_unit_w64<T, D, CAUSAL, KG, ABL, false, 4>(a, blockIdx.x, a.vgrid, smem);
    } else {
        for (int vb = blockIdx.x; vb < a.vgrid; vb += gridDim.x) {
            prefix_unit_w64<T, D, CAUSAL, KG, ABL, true, 4>(a, vb, a.vgrid, smem);
            if (vb + (int)gridDim.x < a.vgrid) __syncthreads();  // the unit's LDS merge buffers are the next unit's rings
        }
    }
}

// The 8-wave kernel (two waves per SIMD, one workgroup per unit) is held to 160 architectural VGPRs -- amdgpu_num_vgpr(80)
// counts VGPRs and accumulator registers alike and the kernel uses no accumulator register -- so that v[160:255] stay the
// unit's own (prefix_unit_w64.h, RegsV).
template <typename T, int D, bool CAUSAL, int KG, int ABL = 0>
__global__ __launch_bounds__(512) __attribute__((amdgpu_num_vgpr(80))) void prefix_attn_w64x8_kernel(const PrefixArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    warm_kernargs();
    prefix_unit_w64<T, D, CAUSAL, KG, ABL, false, 8>(a, blockIdx.x, a.vgrid, smem);
}

template <typename T, int D, bool CAUSAL, int KG, int ABL = 0, bool PERSIST = false, int NW = 4>
inline int launch_prefix_w64_k(const PrefixArgs& a, int grid, hipStream_t s) {
    constexpr size_t lds = (KG == 2 ? 2 : 1) * 256 * (D * 2) + 4 * 2 * 128 * sizeof(float);
    auto kern = [] {
        if constexpr (NW == 8) return prefix_attn_w64x8_kernel<T, D, CAUSAL, KG, ABL>;
        else return prefix_attn_w64_kernel<T, D, CAUSAL, KG, ABL, PERSIST>;
    }();
    static_assert(NW == 4 || !PERSIST, "the 8-wave kernel runs one unit per workgroup");
    // once per instantiation, thread-safe (C++11 static initialisation); the value never changes afterwards
    static const hipError_t attr_rc = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (attr_rc != hipSuccess) return (int)attr_rc;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NW), lds, s, a);
    return (int)hipGetLastError();
}

template <typename T, int D, bool CAUSAL, int KG, int ABL = 0>
inline int launch_prefix_w64_t(const PrefixArgs& a, int grid, hipStream_t s) {
    if (grid >= a.vgrid) {
        if constexpr (D == 128) {
            if (a.waves == 8) return launch_prefix_w64_k<T, D, CAUSAL, KG, ABL, false, 8>(a, a.vgrid, s);
        }
        return launch_prefix_w64_k<T, D, CAUSAL, KG, ABL, false>(a, a.vgrid, s);
    }
    // fewer workgroups than units: only the decode entry asks for it, and its levels are never causal
    if constexpr (!CAUSAL && ABL == 0) return launch_prefix_w64_k<T, D, CAUSAL, KG, ABL, true>(a, grid, s);
    else return (int)hipErrorInvalidValue;
}

// Shapes -> instantiation for one storage dtype (TT); hipErrorInvalidValue for a combination that is not built.
template <typename TT>
inline int launch_prefix_w64_dtype(const PrefixArgs& a, int D, bool causal, int grid, hipStream_t s) {
#define HYD_DISPATCH(DD, KK) \
    return causal ? launch_prefix_w64_t<TT, DD, true, KK>(a, grid, s) : launch_prefix_w64_t<TT, DD, false, KK>(a, grid, s)
    if (D == 256) {  // one query block per wave: 128 rows per workgroup, every wave walks all keys (KG = 1)
        if (a.wg_rows != 128) return (int)hipErrorInvalidValue;
        HYD_DISPATCH(256, 1);
    }
    if (a.wg_rows == 256) {
        if (D == 128) { HYD_DISPATCH(128, 1); }
        if (D == 64) { HYD_DISPATCH(64, 1); }
        return (int)hipErrorInvalidValue;
    }
    if (D == 128) { HYD_DISPATCH(128, 2); }
    if (D == 64) { HYD_DISPATCH(64, 2); }
#undef HYD_DISPATCH
    return (int)hipErrorInvalidValue;
}

}  // namespace hyd
