// Device-side pieces of the suffix-pass kernels (suffix_attn.hip): 16-bit widening, uniform pointers, online-softmax state merges, the output-row epilogue.
#pragma once
#include "hyd_kernels.h"

namespace hyd {

template <typename T>
__device__ __forceinline__ void widen8(const u32x4& v, float (&f)[8]) {
    using TR = Traits<T>;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f[2 * i] = TR::lo(v[i]);
        f[2 * i + 1] = TR::hi(v[i]);
    }
}

// tell hipcc a pointer is wave-uniform (it is: derived from blockIdx and the wave index) so that it
// lives in SGPRs and loads take the base + 32-bit-offset form
typedef const __attribute__((address_space(1))) char* gchar_p;
typedef const __attribute__((address_space(1))) u32x4* gu32x4_p;
__device__ __forceinline__ gchar_p uniform_ptr(const char* p) {
    const uint64_t v = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return (gchar_p)(((uint64_t)hi << 32) | lo);
}

// merge (m, l, acc) state pairs; all values in base-2 domain
__device__ __forceinline__ void merge_state(float& m, float& l, float (&acc)[8], float m2, float l2,
                                            const float (&acc2)[8]) {
    const float mf = fmaxf(m, m2);
    const float ms = (mf == -INFINITY) ? 0.f : mf;
    const float a1 = fast_exp2(m - ms), a2 = fast_exp2(m2 - ms);
    l = l * a1 + l2 * a2;
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = acc[j] * a1 + acc2[j] * a2;
    m = mf;
}

// Epilogue of one output row (8 dims per lane, the D/8 lanes of a row call it together): normalise the suffix pass's
// (m, l, acc), merge with the prefix partials (attention.py:21-43 semantics, N partials) and store.  The first npre
// partials already sit in registers (PrePartials: prefetched under the K/V stream).
// 16-bit partials fetched under the K/V stream: the first NPRE of a call (consecutive ones from partial 0).  One
// covers the usual single prefix level; the second is a two-level hierarchy's (BASELINE config 4): left to the epilogue,
// its two dependent round trips (LSE, then the row) cost 17 us of a 113 us suffix pass there.
// The kernels are instantiated for NPRE = 1 and 2 and picked by a.n_pre (counted on the host): carrying the second slot
// through the single-level call costs it 7 % at suffix 16 (registers and issue slots of a ~2 us wave).
template <int NPRE>
struct PrePartials {
    float lse[NPRE];
    u32x4 out[NPRE];
};
__device__ __forceinline__ int n_prefetched(const SuffixArgs& a) { return a.n_pre; }
template <int NPRE>
__device__ __forceinline__ void prefetch_partials(const SuffixArgs& a, int npre, int64_t ridx, int sub, int D, PrePartials<NPRE>& pp) {
#pragma unroll
    for (int i = 0; i < NPRE; ++i) {
        pp.lse[i] = 0.f;
        pp.out[i] = u32x4{0u, 0u, 0u, 0u};
        if (i < npre) {
            pp.lse[i] = a.partials[i].lse[ridx];
            pp.out[i] = *reinterpret_cast<const u32x4*>(static_cast<const uint16_t*>(a.partials[i].out) + ridx * D + sub * 8);
        }
    }
}

template <typename T, int D, int NBATCH = 4, int NPRE = 1>
__device__ __forceinline__ void finish_row(const SuffixArgs& a, int64_t ridx, int sub, float m, float l, const float (&acc)[8],
                                           int npre, const PrePartials<NPRE>& pp) {
    using TR = Traits<T>;
    const float inv = l > 0.f ? 1.0f / l : 0.f;
    const float lse_s = l > 0.f ? m * kLn2 + __logf(l) : -INFINITY;
    if (a.lse && sub == 0) a.lse[ridx] = lse_s;
    float num[8];
    if (a.n_partials == 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) num[j] = acc[j] * inv;
    } else {
        const int i0 = npre;  // partials [0, npre) already sit in registers
        // The remaining partials (split-KV slices, further levels) are read NBATCH at a time with clamped indices,
        // so that a batch's loads are all in flight together instead of one memory latency per partial.
        const int np = a.n_partials;
        float M = lse_s;
#pragma unroll
        for (int i = 0; i < NPRE; ++i)
            if (i < npre) M = fmaxf(M, pp.lse[i]);
        for (int i = i0; i < np; i += NBATCH) {
            float lv[NBATCH];
#pragma unroll
            for (int j = 0; j < NBATCH; ++j) lv[j] = a.partials[min(i + j, np - 1)].lse[ridx];
#pragma unroll
            for (int j = 0; j < NBATCH; ++j) M = fmaxf(M, lv[j]);
        }
        const float Ms = (M == -INFINITY) ? 0.f : M;
        const float ws = __expf(lse_s - Ms);
        float den = ws;
#pragma unroll
        for (int j = 0; j < 8; ++j) num[j] = acc[j] * (inv * ws);
#pragma unroll
        for (int i = 0; i < NPRE; ++i) {
            if (i < npre) {
                const float w = __expf(pp.lse[i] - Ms);
                den += w;
                float pv[8];
                widen8<T>(pp.out[i], pv);
#pragma unroll
                for (int j = 0; j < 8; ++j) num[j] = __builtin_fmaf(w, pv[j], num[j]);
            }
        }
        for (int i = i0; i < np;) {
            // a batch = up to NBATCH consecutive partials of the same element type (slices of one level are adjacent)
            const bool f32 = a.partials[i].is_f32 != 0;
            int cnt = 1;
            while (cnt < NBATCH && i + cnt < np && (a.partials[i + cnt].is_f32 != 0) == f32) ++cnt;
            float lw[NBATCH];
            float pv[NBATCH][8];
            if (f32) {
                f32x4 x0[NBATCH], x1[NBATCH];
#pragma unroll
                for (int j = 0; j < NBATCH; ++j) {
                    const PartialDev& pd = a.partials[i + min(j, cnt - 1)];
                    const float* po = static_cast<const float*>(pd.out) + ridx * D + sub * 8;
                    lw[j] = pd.lse[ridx];
                    x0[j] = *reinterpret_cast<const f32x4*>(po);
                    x1[j] = *reinterpret_cast<const f32x4*>(po + 4);
                }
#pragma unroll
                for (int j = 0; j < NBATCH; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        pv[j][e] = x0[j][e];
                        pv[j][4 + e] = x1[j][e];
                    }
            } else {
                u32x4 x[NBATCH];
#pragma unroll
                for (int j = 0; j < NBATCH; ++j) {
                    const PartialDev& pd = a.partials[i + min(j, cnt - 1)];
                    lw[j] = pd.lse[ridx];
                    x[j] = *reinterpret_cast<const u32x4*>(static_cast<const uint16_t*>(pd.out) + ridx * D + sub * 8);
                }
#pragma unroll
                for (int j = 0; j < NBATCH; ++j) widen8<T>(x[j], pv[j]);
            }
#pragma unroll
            for (int j = 0; j < NBATCH; ++j) {
                const float w = j < cnt ? __expf(lw[j] - Ms) : 0.f;
                den += w;
#pragma unroll
                for (int e = 0; e < 8; ++e) num[e] = __builtin_fmaf(w, pv[j][e], num[e]);
            }
            i += cnt;
        }
        const float dinv = den > 0.f ? 1.0f / den : 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) num[j] *= dinv;
    }
    const u32x4 pk = {TR::pack2(num[0], num[1]), TR::pack2(num[2], num[3]), TR::pack2(num[4], num[5]),
                      TR::pack2(num[6], num[7])};
    *reinterpret_cast<u32x4*>(static_cast<uint16_t*>(a.out) + ridx * D + sub * 8) = pk;
}

// Merge the 64 / LPK lane groups of a wave (butterfly: every lane ends with the total).  Cross-lane exchange on the
// VALU (no LDS round trips): lanes xor 32 / xor 16 through v_permlane32_swap / v_permlane16_swap, which hand BOTH lanes
// the ordered pair (even side, odd side), so the merge is computed identically on both; lanes xor 8 (D = 64 only)
// through a DPP row rotate.
template <int LPK>
__device__ __forceinline__ void merge_lane_groups(float& m, float& l, float (&acc)[8]) {
#pragma unroll
    for (int off = LPK; off < 64; off <<= 1) {
        float m1, m2, l1, l2, a1v[8], a2v[8];
        auto xchg = [&](float x, float& lo, float& hi) __attribute__((always_inline)) {
            const int xi = __builtin_bit_cast(int, x);
            if (off == 32) {
                auto p = __builtin_amdgcn_permlane32_swap(xi, xi, false, false);
                lo = __builtin_bit_cast(float, (int)p[0]);
                hi = __builtin_bit_cast(float, (int)p[1]);
            } else if (off == 16) {
                auto p = __builtin_amdgcn_permlane16_swap(xi, xi, false, false);
                lo = __builtin_bit_cast(float, (int)p[0]);
                hi = __builtin_bit_cast(float, (int)p[1]);
            } else {  // off == 8: partner inside the 16-lane DPP row
                lo = x;
                hi = dpp_f32<0x128>(x);  // row_ror:8
            }
        };
        xchg(m, m1, m2);
        xchg(l, l1, l2);
#pragma unroll
        for (int j = 0; j < 8; ++j) xchg(acc[j], a1v[j], a2v[j]);
        const float mf = fmaxf(m1, m2);
        const float ms = (mf == -INFINITY) ? 0.f : mf;
        const float w1 = fast_exp2(m1 - ms), w2 = fast_exp2(m2 - ms);
        m = mf;
        l = l1 * w1 + l2 * w2;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = a1v[j] * w1 + a2v[j] * w2;
    }
}

}  // namespace hyd
