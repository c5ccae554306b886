// LSE combine for N partials (K4/K5 of SURVEY.md).
// Replaces /root/reference/hydragen/attention.py:154-174 combine_lse: both the 2-partial Triton kernel
// (:46-151) and the N-partial torch fallback (:21-43).  Pure HBM traffic: one pass, 16-byte accesses
// when the head dim allows, any head dim otherwise (the reference test grid includes D = 63 and 129,
// tests/test_combine_lse.py:14).  fp32 maths, result in the output dtype (:75-98).
#include "hyd_kernels.h"

namespace hyd {

__device__ __forceinline__ float load_as_f32(const void* p, int64_t i, int dtype) {
    if (dtype == HYD_F32) return static_cast<const float*>(p)[i];
    const uint16_t u = static_cast<const uint16_t*>(p)[i];
    if (dtype == HYD_BF16) return __builtin_bit_cast(float, (uint32_t)u << 16);
    return (float)__builtin_bit_cast(_Float16, u);
}
__device__ __forceinline__ void store_from_f32(void* p, int64_t i, int dtype, float x) {
    if (dtype == HYD_F32) {
        static_cast<float*>(p)[i] = x;
    } else if (dtype == HYD_BF16) {
        static_cast<__bf16*>(p)[i] = (__bf16)x;
    } else {
        static_cast<_Float16*>(p)[i] = (_Float16)x;
    }
}

__device__ __forceinline__ int64_t lse_out_index(const CombineArgs& a, int64_t row) {
    if (a.lse_layout == HYD_LSE_BQH) return row;
    const int64_t tok = row / a.Hq;
    const int h = (int)(row % a.Hq);
    return ((tok / a.qpg) * a.Hq + h) * a.qpg + tok % a.qpg;
}

// generic: one thread per (row, element)
__global__ __launch_bounds__(256) void combine_scalar_kernel(const CombineArgs a) {
    const int64_t total = a.rows * a.D;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int64_t row = idx / a.D;
        float M = -INFINITY;
        for (int i = 0; i < a.n; ++i) M = fmaxf(M, a.lses[i][row]);
        const float Ms = (M == -INFINITY) ? 0.f : M;
        float num = 0.f, den = 0.f;
        for (int i = 0; i < a.n; ++i) {
            const float w = __expf(a.lses[i][row] - Ms);
            den += w;
            const int dt = a.dtype_in == HYD_MIXED ? (((a.f32_mask >> i) & 1) ? HYD_F32 : a.dtype_out) : a.dtype_in;
            num = __builtin_fmaf(w, load_as_f32(a.outs[i], idx, dt), num);
        }
        store_from_f32(a.out, idx, a.dtype_out, den > 0.f ? num / den : 0.f);
        if (a.out_lse && idx % a.D == 0) a.out_lse[lse_out_index(a, row)] = den > 0.f ? M + __logf(den) : -INFINITY;
    }
}

// vector path: D % 8 == 0; one thread per (row, 8 elements)
template <int DT_IN, int DT_OUT>
__global__ __launch_bounds__(256) void combine_vec_kernel(const CombineArgs a) {
    const int cpr = a.D >> 3;
    const int64_t total = a.rows * cpr;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int64_t row = idx / cpr;
        const int64_t e0 = idx * 8;
        float M = -INFINITY;
        for (int i = 0; i < a.n; ++i) M = fmaxf(M, a.lses[i][row]);
        const float Ms = (M == -INFINITY) ? 0.f : M;
        float num[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        float den = 0.f;
        for (int i = 0; i < a.n; ++i) {
            const float w = __expf(a.lses[i][row] - Ms);
            den += w;
            float x[8];
            if (DT_IN == HYD_F32 || (DT_IN == HYD_MIXED && ((a.f32_mask >> i) & 1))) {
                const f32x4 x0 = *reinterpret_cast<const f32x4*>(static_cast<const float*>(a.outs[i]) + e0);
                const f32x4 x1 = *reinterpret_cast<const f32x4*>(static_cast<const float*>(a.outs[i]) + e0 + 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    x[j] = x0[j];
                    x[4 + j] = x1[j];
                }
            } else {
                const u32x4 v = *reinterpret_cast<const u32x4*>(static_cast<const uint16_t*>(a.outs[i]) + e0);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (DT_IN == HYD_BF16 || (DT_IN == HYD_MIXED && DT_OUT == HYD_BF16)) {
                        x[2 * j] = Traits<BF16>::lo(v[j]);
                        x[2 * j + 1] = Traits<BF16>::hi(v[j]);
                    } else {
                        x[2 * j] = Traits<F16>::lo(v[j]);
                        x[2 * j + 1] = Traits<F16>::hi(v[j]);
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) num[j] = __builtin_fmaf(w, x[j], num[j]);
        }
        const float dinv = den > 0.f ? 1.0f / den : 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) num[j] *= dinv;
        if (DT_OUT == HYD_F32) {
            f32x4 y0 = {num[0], num[1], num[2], num[3]}, y1 = {num[4], num[5], num[6], num[7]};
            *reinterpret_cast<f32x4*>(static_cast<float*>(a.out) + e0) = y0;
            *reinterpret_cast<f32x4*>(static_cast<float*>(a.out) + e0 + 4) = y1;
        } else {
            u32x4 pk;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                pk[j] = DT_OUT == HYD_BF16 ? Traits<BF16>::pack2(num[2 * j], num[2 * j + 1])
                                           : Traits<F16>::pack2(num[2 * j], num[2 * j + 1]);
            *reinterpret_cast<u32x4*>(static_cast<uint16_t*>(a.out) + e0) = pk;
        }
        if (a.out_lse && idx % cpr == 0) a.out_lse[lse_out_index(a, row)] = den > 0.f ? M + __logf(den) : -INFINITY;
    }
}

int launch_combine(const CombineArgs& a, hipStream_t s) {
    if (a.rows == 0) return 0;
    const bool vec = (a.D % 8) == 0 && !a.scalar_only;
    const int64_t items = vec ? a.rows * (a.D / 8) : a.rows * a.D;
    int grid = (int)((items + 255) / 256);
    if (grid > 256 * 16) grid = 256 * 16;  // grid-stride beyond 16 blocks per CU
    if (!vec) {
        hipLaunchKernelGGL(combine_scalar_kernel, dim3(grid), dim3(256), 0, s, a);
        return (int)hipGetLastError();
    }
#define HYD_CMB(I, O)                                                                   \
    if (a.dtype_in == I && a.dtype_out == O) {                                          \
        hipLaunchKernelGGL((combine_vec_kernel<I, O>), dim3(grid), dim3(256), 0, s, a); \
        return (int)hipGetLastError();                                                  \
    }
    HYD_CMB(HYD_F16, HYD_F16)
    HYD_CMB(HYD_BF16, HYD_BF16)
    HYD_CMB(HYD_F32, HYD_F32)
    HYD_CMB(HYD_F32, HYD_F16)
    HYD_CMB(HYD_F32, HYD_BF16)
    HYD_CMB(HYD_MIXED, HYD_F16)
    HYD_CMB(HYD_MIXED, HYD_BF16)
#undef HYD_CMB
    hipLaunchKernelGGL(combine_scalar_kernel, dim3(grid), dim3(256), 0, s, a);
    return (int)hipGetLastError();
}

}  // namespace hyd
