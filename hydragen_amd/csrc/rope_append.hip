// Fused decode-step preamble of the attention block ("next" row 1 of SURVEY 8f): one kernel that
//   * applies rotary position embedding to this step's q and k at ABSOLUTE positions
//     (/root/reference/hydragen/llama.py:485-501, HF rotate-half convention),
//   * appends the rotated k and the v at index (position - shared length) of the unique KV caches
//     (llama.py:236-262, two scatter_ kernels in the reference; index rule llama.py:487-492),
//   * emits seq_lens[b] = index + 1 as int32 for the suffix kernel (llama.py:569, flash.py:220),
// replacing ~10 small torch kernels per layer per token.  HBM-bound and tiny: B*(Hq+2*Hkv)*D elements.
#include "hyd_kernels.h"

namespace hyd {

template <typename T>
__device__ __forceinline__ void rope8(const u32x4& lo, const u32x4& hi, const float* c, const float* s, u32x4& olo,
                                      u32x4& ohi) {
    using TR = Traits<T>;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float x0 = TR::lo(lo[i]), x1 = TR::hi(lo[i]);
        const float y0 = TR::lo(hi[i]), y1 = TR::hi(hi[i]);
        const float c0 = c[2 * i], c1 = c[2 * i + 1], s0 = s[2 * i], s1 = s[2 * i + 1];
        olo[i] = TR::pack2(x0 * c0 - y0 * s0, x1 * c1 - y1 * s1);
        ohi[i] = TR::pack2(y0 * c0 + x0 * s0, y1 * c1 + x1 * s1);
    }
}

template <typename T, int D>
__global__ __launch_bounds__(256) void rope_append_kernel(const RopeArgs a) {
    constexpr int TPR = D / 16;  // threads per row: each owns 8 dims of the first half + the matching 8 of the second
    const int rows_per_b = a.Hq + 2 * a.Hkv;
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t row = gid / TPR;
    const int sub = (int)(gid % TPR);
    if (row >= (int64_t)a.B * rows_per_b) return;
    const int b = (int)(row / rows_per_b);
    const int h = (int)(row % rows_per_b);
    const int64_t pos_raw = a.pos[(int64_t)b * a.pos_stride];
    const int64_t idx = pos_raw - (a.shared_len ? a.shared_len[b] : 0);
    // never read past the rotary tables (the host checks the range before launching; a kernel cannot raise)
    const int64_t pos = pos_raw < 0 ? 0 : (pos_raw >= a.max_pos ? (int64_t)a.max_pos - 1 : pos_raw);
    if (h == 0 && sub == 0) a.seq_lens[b] = (int32_t)(idx + 1);
    const int d0 = sub * 8;
    if (h < a.Hq + a.Hkv) {
        const bool isq = h < a.Hq;
        const uint16_t* src = isq ? static_cast<const uint16_t*>(a.q) + (int64_t)b * a.q_bs + (int64_t)h * D
                                  : static_cast<const uint16_t*>(a.k) + (int64_t)b * a.k_bs + (int64_t)(h - a.Hq) * D;
        const u32x4 lo = *reinterpret_cast<const u32x4*>(src + d0);
        const u32x4 hi = *reinterpret_cast<const u32x4*>(src + D / 2 + d0);
        float c[8], s[8];
        const float* cr = a.cos + pos * a.cs_stride + d0;
        const float* sr = a.sin + pos * a.cs_stride + d0;
        const f32x4 c0 = *reinterpret_cast<const f32x4*>(cr), c1 = *reinterpret_cast<const f32x4*>(cr + 4);
        const f32x4 s0 = *reinterpret_cast<const f32x4*>(sr), s1 = *reinterpret_cast<const f32x4*>(sr + 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            c[i] = c0[i];
            c[4 + i] = c1[i];
            s[i] = s0[i];
            s[4 + i] = s1[i];
        }
        u32x4 olo, ohi;
        rope8<T>(lo, hi, c, s, olo, ohi);
        uint16_t* dst;
        if (isq) {
            dst = static_cast<uint16_t*>(a.q_out) + ((int64_t)b * a.Hq + h) * D;
        } else {
            if (idx < 0 || idx >= a.cache_len) return;  // out of the allocated cache: never write out of bounds
            dst = static_cast<uint16_t*>(a.k_cache) + (int64_t)b * a.kc_bs + idx * a.kc_ts + (int64_t)(h - a.Hq) * a.kc_hs;
        }
        *reinterpret_cast<u32x4*>(dst + d0) = olo;
        *reinterpret_cast<u32x4*>(dst + D / 2 + d0) = ohi;
    } else {
        if (idx < 0 || idx >= a.cache_len) return;
        const int hv = h - a.Hq - a.Hkv;
        const uint16_t* src = static_cast<const uint16_t*>(a.v) + (int64_t)b * a.v_bs + (int64_t)hv * D;
        uint16_t* dst = static_cast<uint16_t*>(a.v_cache) + (int64_t)b * a.vc_bs + idx * a.vc_ts + (int64_t)hv * a.vc_hs;
        *reinterpret_cast<u32x4*>(dst + d0) = *reinterpret_cast<const u32x4*>(src + d0);
        *reinterpret_cast<u32x4*>(dst + D / 2 + d0) = *reinterpret_cast<const u32x4*>(src + D / 2 + d0);
    }
}

int launch_rope_append(const RopeArgs& a, int dtype, int D, hipStream_t s) {
    const int64_t threads = (int64_t)a.B * (a.Hq + 2 * a.Hkv) * (D / 16);
    const int grid = (int)((threads + 255) / 256);
    if (grid == 0) return 0;
#define HYD_ROPE(TT, DD) hipLaunchKernelGGL((rope_append_kernel<TT, DD>), dim3(grid), dim3(256), 0, s, a)
    if (dtype == HYD_F16) {
        if (D == 128) HYD_ROPE(F16, 128);
        else if (D == 64) HYD_ROPE(F16, 64);
        else if (D == 256) HYD_ROPE(F16, 256);
        else return (int)hipErrorInvalidValue;
    } else {
        if (D == 128) HYD_ROPE(BF16, 128);
        else if (D == 64) HYD_ROPE(BF16, 64);
        else if (D == 256) HYD_ROPE(BF16, 256);
        else return (int)hipErrorInvalidValue;
    }
#undef HYD_ROPE
    return (int)hipGetLastError();
}

}  // namespace hyd
