// Shared device-side helpers for the gfx950 (CDNA4, wave64) kernels of libhydragen_hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/hydragen_hip.h"

namespace hyd {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;

struct F16 {};
struct BF16 {};

template <typename T>
struct Traits;

template <>
struct Traits<F16> {
    static constexpr int kDtype = HYD_F16;
    // D(32x32) += A(32x16) * B(16x32); each lane carries 8 consecutive k of row/col (lane & 31),
    // k-half (lane >> 5)
    static __device__ __forceinline__ f32x16 mfma32(u32x4 a, u32x4 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c,
                                                      0, 0, 0);
    }
    static __device__ __forceinline__ uint32_t pack2(float lo, float hi) {
        f16x2 v = {(_Float16)lo, (_Float16)hi};
        return __builtin_bit_cast(uint32_t, v);
    }
    static __device__ __forceinline__ float lo(uint32_t u) { return (float)__builtin_bit_cast(f16x2, u)[0]; }
    static __device__ __forceinline__ float hi(uint32_t u) { return (float)__builtin_bit_cast(f16x2, u)[1]; }
    static __device__ __forceinline__ float dot2(uint32_t a, uint32_t b, float c) {
        return __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2, a), __builtin_bit_cast(f16x2, b), c, false);
    }
};

template <>
struct Traits<BF16> {
    static constexpr int kDtype = HYD_BF16;
    static __device__ __forceinline__ f32x16 mfma32(u32x4 a, u32x4 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b),
                                                       c, 0, 0, 0);
    }
    static __device__ __forceinline__ uint32_t pack2(float lo, float hi) {
        bf16x2 v = {(__bf16)lo, (__bf16)hi};
        return __builtin_bit_cast(uint32_t, v);
    }
    static __device__ __forceinline__ float lo(uint32_t u) { return __builtin_bit_cast(float, u << 16); }
    static __device__ __forceinline__ float hi(uint32_t u) { return __builtin_bit_cast(float, u & 0xffff0000u); }
    static __device__ __forceinline__ float dot2(uint32_t a, uint32_t b, float c) {
        return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, a), __builtin_bit_cast(bf16x2, b), c,
                                               false);
    }
};

// A kernel's arguments span several 64-byte lines of the scalar cache, and hipcc fetches fields where it first needs them:
// the fetches that miss wait for one another before the first global load is issued.  Touch the first four lines at once,
// first thing in the kernel: one miss time instead of several (prefix pass: 1.1 k of 4.4 k prologue cycles).  The loads stay
// in flight beside hipcc's own first argument fetches; the four registers (fixed: not ones hipcc's fetches are landing in)
// stay claimed until the wait.
__device__ __forceinline__ void warm_kernargs_256() {
    const void* ka = (const void*)__builtin_amdgcn_kernarg_segment_ptr();
    unsigned t0, t1, t2, t3;
    asm volatile("s_load_dword %0, %4, 0x0\n\ts_load_dword %1, %4, 0x40\n\ts_load_dword %2, %4, 0x80\n\ts_load_dword %3, %4, 0xc0"
                 : "=&{s96}"(t0), "=&{s97}"(t1), "=&{s98}"(t2), "=&{s99}"(t3)  // early-clobber: the base may not share s[96:99]
                 : "s"(ka)
                 : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" : "+{s96}"(t0), "+{s97}"(t1), "+{s98}"(t2), "+{s99}"(t3));
}

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float fast_log2(float x) { return __builtin_amdgcn_logf(x); }

// reductions over the lane pair (l, l ^ 32); symmetric in the two results of permlane32_swap, so they
// do not depend on which of the two returned registers is "mine"
__device__ __forceinline__ float pair_max(float x) {
    int xi = __builtin_bit_cast(int, x);
    auto r = __builtin_amdgcn_permlane32_swap(xi, xi, false, false);
    return fmaxf(__builtin_bit_cast(float, (int)r[0]), __builtin_bit_cast(float, (int)r[1]));
}
__device__ __forceinline__ float pair_sum(float x) {
    int xi = __builtin_bit_cast(int, x);
    auto r = __builtin_amdgcn_permlane32_swap(xi, xi, false, false);
    return __builtin_bit_cast(float, (int)r[0]) + __builtin_bit_cast(float, (int)r[1]);
}

template <int CTRL>
__device__ __forceinline__ float dpp_f32(float x) {
    int xi = __builtin_bit_cast(int, x);
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(xi, xi, CTRL, 0xf, 0xf, false));
}

// all-reduce (sum) over aligned groups of LANES (8, 16 or 32) consecutive lanes: DPP inside a 16-lane row, one
// v_permlane16_swap (both lanes receive the ordered pair of the two rows' values) across the rows of a 32-lane group
template <int LANES>
__device__ __forceinline__ float group_sum(float x) {
    x += dpp_f32<0xB1>(x);   // quad_perm [1,0,3,2]
    x += dpp_f32<0x4E>(x);   // quad_perm [2,3,0,1]
    x += dpp_f32<0x141>(x);  // row_half_mirror
    if (LANES >= 16) x += dpp_f32<0x140>(x);  // row_mirror
    if (LANES == 32) {
        const int xi = __builtin_bit_cast(int, x);
        auto r = __builtin_amdgcn_permlane16_swap(xi, xi, false, false);
        x = __builtin_bit_cast(float, (int)r[0]) + __builtin_bit_cast(float, (int)r[1]);
    }
    return x;
}

// XCD-aware, bijective remap of the hardware block id: blocks land on XCD (id % 8); give each XCD a
// contiguous range of logical ids so blocks that share a K/V slice share an L2.
__device__ __forceinline__ int xcd_remap(int wid, int nwg) {
    const int xcd = wid & 7, slot = wid >> 3;
    const int qn = nwg >> 3, rn = nwg & 7;
    return (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + slot;
}

}  // namespace hyd
