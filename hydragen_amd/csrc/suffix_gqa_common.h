// Device-side pieces of the matrix-core suffix kernel (suffix_attn_gqa.hip): asm MFMAs,
// transposing LDS reads, buffer resources, LDS-DMA, the asm-owned K register sets a[0:63], cross-lane reductions.
#pragma once
#include <type_traits>

#include "hyd_kernels.h"

namespace hyd {

namespace {

// acc += a . b on the matrix cores, operands and accumulator in VGPRs.  asm on purpose: with MFMA *builtins* in a kernel
// whose asm names AGPRs, hipcc moves the accumulators into AGPRs of its own choosing (the ones this file's loads are
// in flight to).  hipcc's hazard recogniser does not look inside: callers drain before a VALU read of `acc`.
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
// K fragment I of the two register sets (set = I / 8, 16-key half = (I / 4) & 1, 32-dim chunk = I & 3) in a[4I : 4I+3]
template <int I>
struct KReg;
#define HYD_KREG(I, A, B, C, E)                                                                                          \
    template <>                                                                                                          \
    struct KReg<I> {                                                                                                     \
        template <int OFF, bool NT> /* 16 B per lane, bounds-checked: rows past the resource's end read as zero */      \
        static __device__ __forceinline__ void load(u32x4 rsrc, unsigned voff, unsigned soff) {                          \
            if constexpr (NT)                                                                                            \
                asm volatile("buffer_load_dwordx4 a[" #A ":" #E "], %0, %1, %2 offen offset:%3 nt" ::"v"(voff),          \
                             "s"(rsrc), "s"(soff), "i"(OFF)                                                              \
                             : "memory", "a" #A, "a" #B, "a" #C, "a" #E);                                                \
            else                                                                                                         \
                asm volatile("buffer_load_dwordx4 a[" #A ":" #E "], %0, %1, %2 offen offset:%3" ::"v"(voff), "s"(rsrc),  \
                             "s"(soff), "i"(OFF)                                                                         \
                             : "memory", "a" #A, "a" #B, "a" #C, "a" #E);                                                \
        }                                                                                                                \
        template <typename T, bool FIRST> /* s (+)= K_frag . q^T */                                                      \
        static __device__ __forceinline__ void qk(f32x4& s, const u32x4& q) {                                            \
            constexpr bool BF = std::is_same<T, BF16>::value;                                                            \
            if constexpr (BF && FIRST) asm volatile("v_mfma_f32_16x16x32_bf16 %0, a[" #A ":" #E "], %1, 0" : "=&v"(s) : "v"(q)); \
            else if constexpr (BF) asm volatile("v_mfma_f32_16x16x32_bf16 %0, a[" #A ":" #E "], %1, %0" : "+v"(s) : "v"(q));     \
            else if constexpr (FIRST) asm volatile("v_mfma_f32_16x16x32_f16 %0, a[" #A ":" #E "], %1, 0" : "=&v"(s) : "v"(q));   \
            else asm volatile("v_mfma_f32_16x16x32_f16 %0, a[" #A ":" #E "], %1, %0" : "+v"(s) : "v"(q));                        \
        }                                                                                                                \
    };
HYD_KREG(0, 0, 1, 2, 3) HYD_KREG(1, 4, 5, 6, 7) HYD_KREG(2, 8, 9, 10, 11) HYD_KREG(3, 12, 13, 14, 15)
HYD_KREG(4, 16, 17, 18, 19) HYD_KREG(5, 20, 21, 22, 23) HYD_KREG(6, 24, 25, 26, 27) HYD_KREG(7, 28, 29, 30, 31)
HYD_KREG(8, 32, 33, 34, 35) HYD_KREG(9, 36, 37, 38, 39) HYD_KREG(10, 40, 41, 42, 43) HYD_KREG(11, 44, 45, 46, 47)
HYD_KREG(12, 48, 49, 50, 51) HYD_KREG(13, 52, 53, 54, 55) HYD_KREG(14, 56, 57, 58, 59) HYD_KREG(15, 60, 61, 62, 63)
#undef HYD_KREG

// Partial-prefetch buffer a[64:96], asm-owned like the K sets: one prefix partial's row piece of this lane -- dims
// [16 db + 4 g4, +4) in a[64 + 4 db : 67 + 4 db] (fp32; a 16-bit partial fills a[64 + 4 db : 65 + 4 db]) and its LSE in a[96] --
// loaded under the K/V stream and read back (v_accvgpr_read) after the counted wait that covers it.
__device__ __forceinline__ void claim_partial_buffer() {
    asm volatile("" ::: "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80",
                 "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96");
}
template <int DB>
struct PReg {
    static constexpr int R0 = 64 + 4 * DB;
    static __device__ __forceinline__ void load_f32(const float* p) {  // 16 B: four fp32 dims
        asm volatile("global_load_dwordx4 a[%1:%2], %0, off" ::"v"(p), "n"(R0), "n"(R0 + 3) : "memory");
    }
    static __device__ __forceinline__ void load_b16(const uint16_t* p) {  // 8 B: four 16-bit dims
        asm volatile("global_load_dwordx2 a[%1:%2], %0, off" ::"v"(p), "n"(R0), "n"(R0 + 1) : "memory");
    }
    template <int J>
    static __device__ __forceinline__ unsigned read() {
        unsigned x;
        asm volatile("v_accvgpr_read_b32 %0, a[%1]" : "=v"(x) : "n"(R0 + J));
        return x;
    }
};
__device__ __forceinline__ void preg_load_lse(const float* p) { asm volatile("global_load_dword a96, %0, off" ::"v"(p) : "memory"); }
__device__ __forceinline__ float preg_read_lse() {
    float x;
    asm volatile("v_accvgpr_read_b32 %0, a96" : "=v"(x));
    return x;
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
