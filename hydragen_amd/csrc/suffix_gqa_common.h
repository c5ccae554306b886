// Device-side pieces of the matrix-core suffix kernel (suffix_attn_gqa.hip): asm MFMAs,
// transposing LDS reads, buffer resources, LDS-DMA, cross-lane reductions.
#pragma once
#include <type_traits>

#include "hyd_kernels.h"

namespace hyd {

namespace {

// acc += a . b on the matrix cores, operands and accumulator in VGPRs (asm: the kernel then uses no accumulator register at
// all and hipcc may allocate the whole 256-register wave as VGPRs).  hipcc's hazard recogniser does not look inside: callers
// drain before a VALU read of `acc`.
template <typename T>
__device__ __forceinline__ void mfma16_acc(f32x4& acc, const u32x4& a, const u32x4& b) {
    // s_nop 1: the operands were just written by plain VALU code (the rescale of acc, the packed P); a VALU write
    // followed by a matrix-core read of the same VGPR wants two wait states, which hipcc inserts only for its own MFMAs
    if constexpr (std::is_same<T, BF16>::value) asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
    else asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}

typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr_g;
__device__ __forceinline__ u32x2 lds_tr16_g(unsigned lds_byte_addr) {
    s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr_g)(uintptr_t)lds_byte_addr);
    return __builtin_bit_cast(u32x2, t);
}

__device__ __forceinline__ u32x4 make_rsrc_g(const void* base, unsigned bytes) {
    const uint64_t b = (uint64_t)(uintptr_t)base;
    u32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((unsigned)b);
    r[1] = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32) & 0xffffu);
    r[2] = __builtin_amdgcn_readfirstlane(bytes);
    r[3] = 0x00020000u;
    return r;
}
// 64 lanes x 16 B, global -> LDS [lds_dst, +1 KiB) (lane-linear image), zero fill past the resource's end
// NT: non-temporal hint for K/V that is read once (a sequence's own cache: the unique phase).  Measured at C5 (277 MB per
// launch): 101.5 -> 96.5 us cold and 140 -> 97 us under the reference's write-flush protocol (the stream no longer competes
// with the flush's dirty lines for the memory-side cache); keys that several workgroups share (a small shared level run on
// this kernel) keep the default policy (C4 cold: 146 us against 149 with the hint).
template <bool NT>
__device__ __forceinline__ void dma16_g(u32x4 rsrc, unsigned voff, unsigned soff, unsigned lds_dst) {
    unsigned keep;
    if constexpr (NT)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen nt lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(voff), "s"(rsrc), "s"(soff), "s"(lds_dst)
                     : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(voff), "s"(rsrc), "s"(soff), "s"(lds_dst)
                     : "memory");
}
template <int N, typename F>
__device__ __forceinline__ void static_for_g(F&& f) {
    if constexpr (N > 0) {
        static_for_g<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

// symmetric all-reduce over the 4 lanes {l, l^16, l^32, l^48} (same query row, the 4 key groups)
__device__ __forceinline__ float quad_max(float x) {
    int xi = __builtin_bit_cast(int, x);
    auto p = __builtin_amdgcn_permlane32_swap(xi, xi, false, false);
    x = fmaxf(__builtin_bit_cast(float, (int)p[0]), __builtin_bit_cast(float, (int)p[1]));
    xi = __builtin_bit_cast(int, x);
    auto q = __builtin_amdgcn_permlane16_swap(xi, xi, false, false);
    return fmaxf(__builtin_bit_cast(float, (int)q[0]), __builtin_bit_cast(float, (int)q[1]));
}
__device__ __forceinline__ float quad_sum(float x) {
    int xi = __builtin_bit_cast(int, x);
    auto p = __builtin_amdgcn_permlane32_swap(xi, xi, false, false);
    x = __builtin_bit_cast(float, (int)p[0]) + __builtin_bit_cast(float, (int)p[1]);
    xi = __builtin_bit_cast(int, x);
    auto q = __builtin_amdgcn_permlane16_swap(xi, xi, false, false);
    return __builtin_bit_cast(float, (int)q[0]) + __builtin_bit_cast(float, (int)q[1]);
}

}  // namespace

}  // namespace hyd
