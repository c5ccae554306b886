// Suffix pass (K2 + K3 of SURVEY.md, with K4/K5 fused into the epilogue): every query row of
// sequence b against the first seq_len[b] keys of b's own (unique) K/V.  ~1 flop/byte: this is an
// HBM-bandwidth kernel, so it is built around wide coalesced loads and many bytes in flight, not MFMA.
//
// Replaces /root/reference/hydragen/flash.py:163-281 flash_attention_seqlen
//   (= hydragen/xformers_stuff.py:189-428 _fwd_kernel_splitK + hydragen/flash.py:76-160 _splitK_reduce)
// and, when partials are passed, hydragen/attention.py:352 combine_lse (:21-43 semantics, N partials).
//
// Mapping (wave64): a "unit" is one (sequence b, kv head).  D/8 lanes cover one key row with a 16-byte
// load each, so one wave instruction fetches 64/(D/8) consecutive keys (4 at D=128).  WPU waves share a
// unit (key iterations interleaved across them); 4/WPU units per 256-thread workgroup, consecutive
// waves = consecutive kv heads of the same sequence, i.e. neighbouring 2*D-byte pieces of the same token
// rows.  q.k partial dot products use v_dot2 and are reduced across the D/8 lanes with DPP adds; each
// lane group runs its own online softmax over its keys; groups, then waves, are merged at the end.
#include <type_traits>

#include <cstdlib>
#include <cstring>

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

template <typename T, int D, int R, int WPU>
__global__ __launch_bounds__(256, (R == 1 ? 6 : 1)) void suffix_attn_kernel(const SuffixArgs a) {
    using TR = Traits<T>;
    constexpr int LPK = D / 8;    // lanes per key row
    constexpr int KPI = 64 / LPK; // keys per wave instruction
    constexpr int U = 4;  // key iterations in flight per wave (x2 tensors x 1 KiB); occupancy supplies the rest
    __shared__ float xbuf[WPU > 1 ? (WPU - 1) * R * (2 + D) : 1];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int sub = lane % LPK, ks = lane / LPK;
    const int wv = wave % WPU;
    // grid = (sequence, kv-head group, row chunk): no integer division on the way to (b, hk).  Sequence
    // is the fastest-varying index on purpose: measured A/B on MI355X (same run, S = 128..256), spreading
    // concurrently running workgroups over different sequences streams 5-10 % faster than walking the
    // head groups of one sequence (whose rows share HBM channels).
    const int b = blockIdx.x;
    const int hk = blockIdx.y * (4 / WPU) + wave / WPU;
    if (hk >= a.Hkv) return;  // uniform per unit (all WPU waves of a unit leave together)
    const int row0 = blockIdx.z * R;

    int len = a.kv_len;
    if (a.sl32) len = a.sl32[b];
    else if (a.sl64) len = (int)a.sl64[b];
    len = max(0, min(len, a.kv_len));

    // ---- query rows (packed 16-bit pairs, 8 dims per lane) -------------------------------------
    u32x4 qp[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int row = row0 + r;
        u32x4 z = {0u, 0u, 0u, 0u};
        if (row < a.rows) {
            const int iq = a.nq == 1 ? 0 : row / a.g, gq = a.nq == 1 ? row : row % a.g;
            const uint16_t* qr = static_cast<const uint16_t*>(a.q) +
                                 (((int64_t)b * a.nq + iq) * a.Hq + hk * a.g + gq) * D + sub * 8;
            qp[r] = *reinterpret_cast<const u32x4*>(qr);
        } else {
            qp[r] = z;
        }
    }
    // ---- prefetch the first partial (the usual single prefix level) for the row this lane group will
    // finish in the epilogue, so its HBM latency overlaps the K/V stream instead of following it -----
    constexpr int RPG = (R + KPI - 1) / KPI;  // epilogue rows per lane group
    float pl0[RPG];
    u32x4 po0[RPG];
    const bool pre0 = a.n_partials > 0 && !a.partials[0].is_f32;
#pragma unroll
    for (int j = 0; j < RPG; ++j) {
        const int r = ks + j * KPI;
        const int row = row0 + r;
        pl0[j] = 0.f;
        po0[j] = u32x4{0u, 0u, 0u, 0u};
        if (pre0 && r < R && row < a.rows) {
            const int iq = a.nq == 1 ? 0 : row / a.g, gq = a.nq == 1 ? row : row % a.g;
            const int64_t ridx = ((int64_t)b * a.nq + iq) * a.Hq + hk * a.g + gq;
            pl0[j] = a.partials[0].lse[ridx];
            po0[j] = *reinterpret_cast<const u32x4*>(static_cast<const uint16_t*>(a.partials[0].out) + ridx * D + sub * 8);
        }
    }

    // wave-uniform unit base (scalar registers) + 32-bit per-lane byte offsets -> SADDR-form loads
    const char* kb_ = reinterpret_cast<const char*>(static_cast<const uint16_t*>(a.k) + (int64_t)b * a.k_bs + (int64_t)hk * a.k_hs);
    const char* vb_ = reinterpret_cast<const char*>(static_cast<const uint16_t*>(a.v) + (int64_t)b * a.v_bs + (int64_t)hk * a.v_hs);
    const gchar_p kbu = uniform_ptr(kb_), vbu = uniform_ptr(vb_);
    const unsigned krs = (unsigned)(a.k_ts * 2), vrs = (unsigned)(a.v_ts * 2);  // token stride in bytes

    float m[R], l[R], acc[R][8];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        m[r] = -INFINITY;
        l[r] = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[r][j] = 0.f;
    }

    const int niter = (len + KPI * WPU - 1) / (KPI * WPU);
    const float sc = a.scale_log2e;
    // One chunk = UU key iterations (UU wave instructions per tensor in flight): loads first, then scores,
    // one online-softmax update, then the P.V accumulation.  Full chunks use UU = U; the tail (and every
    // short sequence) runs one iteration at a time, so a wave never issues loads or arithmetic for
    // iterations beyond ceil(len / keys-per-iteration) -- at small S the kernel is bound by exactly this
    // per-wave instruction overhead, not by HBM.
    auto chunk = [&](auto UU_C, int it) __attribute__((always_inline)) {
        constexpr int UU = decltype(UU_C)::value;
        u32x4 kreg[UU], vreg[UU];
        bool valid[UU];
#pragma unroll
        for (int u = 0; u < UU; ++u) {
            const int key = ((it + u) * WPU + wv) * KPI + ks;
            valid[u] = key < len;
            // never predicate the loads (a branch per load serialises them): clamp to the last valid key,
            // its score is forced to -inf below so it contributes exactly 0
            const int kc = min(key, len - 1);
            kreg[u] = __builtin_nontemporal_load((gu32x4_p)(kbu + ((unsigned)kc * krs + sub * 16)));
            vreg[u] = __builtin_nontemporal_load((gu32x4_p)(vbu + ((unsigned)kc * vrs + sub * 16)));
        }
        float s[R][UU];
#pragma unroll
        for (int u = 0; u < UU; ++u) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                float d = 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) d = TR::dot2(qp[r][i], kreg[u][i], d);
                d = group_sum<LPK>(d);
                s[r][u] = valid[u] ? d * sc : -INFINITY;
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float cmax = s[r][0];
#pragma unroll
            for (int u = 1; u < UU; ++u) cmax = fmaxf(cmax, s[r][u]);
            const float mnew = fmaxf(m[r], cmax);
            const float ms = (mnew == -INFINITY) ? 0.f : mnew;
            const float alpha = fast_exp2(m[r] - ms);
            float ps = 0.f;
#pragma unroll
            for (int u = 0; u < UU; ++u) {
                s[r][u] = fast_exp2(s[r][u] - ms);  // p
                ps += s[r][u];
            }
            l[r] = l[r] * alpha + ps;
            m[r] = mnew;
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[r][j] *= alpha;
        }
        // V is widened one key at a time, right where it is consumed (keeps the register footprint,
        // hence the occupancy that hides HBM latency, independent of UU)
#pragma unroll
        for (int u = 0; u < UU; ++u) {
            float vf[8];
            widen8<T>(vreg[u], vf);
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[r][j] = __builtin_fmaf(s[r][u], vf[j], acc[r][j]);
        }
    };
    int it = 0;
    for (; it + U <= niter; it += U) chunk(std::integral_constant<int, U>{}, it);
    for (; it < niter; ++it) chunk(std::integral_constant<int, 1>{}, it);

    // ---- merge the KPI lane groups of the wave (butterfly: every lane ends with the total) -----
    // Cross-lane exchange on the VALU (no LDS round trips): lanes xor 32 / xor 16 through
    // v_permlane32_swap / v_permlane16_swap, which hand BOTH lanes the ordered pair (even side, odd side),
    // so the merge is computed identically on both; lanes xor 8 (D = 64 only) through a DPP row rotate.
#pragma unroll
    for (int off = LPK; off < 64; off <<= 1) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
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
            xchg(m[r], m1, m2);
            xchg(l[r], l1, l2);
#pragma unroll
            for (int j = 0; j < 8; ++j) xchg(acc[r][j], a1v[j], a2v[j]);
            const float mf = fmaxf(m1, m2);
            const float ms = (mf == -INFINITY) ? 0.f : mf;
            const float w1 = fast_exp2(m1 - ms), w2 = fast_exp2(m2 - ms);
            m[r] = mf;
            l[r] = l1 * w1 + l2 * w2;
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[r][j] = a1v[j] * w1 + a2v[j] * w2;
        }
    }

    // ---- merge the WPU waves of the unit through LDS -------------------------------------------
    if (WPU > 1) {
        float* xb = xbuf;  // [(WPU-1)][R][2 + D]
        if (wv > 0 && ks == 0) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                float* p = xb + ((wv - 1) * R + r) * (2 + D);
                if (sub == 0) {
                    p[0] = m[r];
                    p[1] = l[r];
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) p[2 + sub * 8 + j] = acc[r][j];
            }
        }
        __syncthreads();
        if (wv > 0) return;
#pragma unroll
        for (int w = 1; w < WPU; ++w)
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const float* p = xb + ((w - 1) * R + r) * (2 + D);
                float a2[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) a2[j] = p[2 + sub * 8 + j];
                merge_state(m[r], l[r], acc[r], p[0], p[1], a2);
            }
    }

    // ---- epilogue: normalise, merge with the prefix partials (attention.py:21-43), store -------
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int row = row0 + r;
        if (row >= a.rows || ks != (r % KPI)) continue;
        const int iq = a.nq == 1 ? 0 : row / a.g, gq = a.nq == 1 ? row : row % a.g;
        const int64_t ridx = ((int64_t)b * a.nq + iq) * a.Hq + hk * a.g + gq;  // [B, nq, Hq]
        const float inv = l[r] > 0.f ? 1.0f / l[r] : 0.f;
        const float lse_s = l[r] > 0.f ? m[r] * kLn2 + __logf(l[r]) : -INFINITY;
        if (a.lse && sub == 0) a.lse[ridx] = lse_s;
        float num[8];
        if (a.n_partials == 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) num[j] = acc[r][j] * inv;
        } else {
            const int i0 = pre0 ? 1 : 0;  // partial 0 already sits in registers
            const float l0 = pl0[r / KPI];
            // The remaining partials (split-KV slices, further levels) are read four at a time with clamped indices,
            // so that a batch's loads are all in flight together instead of one memory latency per partial.
            const int np = a.n_partials;
            float M = pre0 ? fmaxf(lse_s, l0) : lse_s;
            for (int i = i0; i < np; i += 4) {
                float lv[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) lv[j] = a.partials[min(i + j, np - 1)].lse[ridx];
                M = fmaxf(fmaxf(M, fmaxf(lv[0], lv[1])), fmaxf(lv[2], lv[3]));
            }
            const float Ms = (M == -INFINITY) ? 0.f : M;
            const float ws = __expf(lse_s - Ms);
            float den = ws;
#pragma unroll
            for (int j = 0; j < 8; ++j) num[j] = acc[r][j] * (inv * ws);
            if (pre0) {
                const float w = __expf(l0 - Ms);
                den += w;
                float pv[8];
                widen8<T>(po0[r / KPI], pv);
#pragma unroll
                for (int j = 0; j < 8; ++j) num[j] = __builtin_fmaf(w, pv[j], num[j]);
            }
            for (int i = i0; i < np;) {
                // a batch = up to 4 consecutive partials of the same element type (slices of one level are adjacent)
                const bool f32 = a.partials[i].is_f32 != 0;
                int cnt = 1;
                while (cnt < 4 && i + cnt < np && (a.partials[i + cnt].is_f32 != 0) == f32) ++cnt;
                float lw[4];
                float pv[4][8];
                if (f32) {
                    f32x4 x0[4], x1[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const PartialDev& pd = a.partials[i + min(j, cnt - 1)];
                        const float* po = static_cast<const float*>(pd.out) + ridx * D + sub * 8;
                        lw[j] = pd.lse[ridx];
                        x0[j] = *reinterpret_cast<const f32x4*>(po);
                        x1[j] = *reinterpret_cast<const f32x4*>(po + 4);
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            pv[j][e] = x0[j][e];
                            pv[j][4 + e] = x1[j][e];
                        }
                } else {
                    u32x4 x[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const PartialDev& pd = a.partials[i + min(j, cnt - 1)];
                        lw[j] = pd.lse[ridx];
                        x[j] = *reinterpret_cast<const u32x4*>(static_cast<const uint16_t*>(pd.out) + ridx * D + sub * 8);
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) widen8<T>(x[j], pv[j]);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
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
        u32x4 pk = {TR::pack2(num[0], num[1]), TR::pack2(num[2], num[3]), TR::pack2(num[4], num[5]),
                    TR::pack2(num[6], num[7])};
        *reinterpret_cast<u32x4*>(static_cast<uint16_t*>(a.out) + ridx * D + sub * 8) = pk;
    }
}

template <typename T, int D, int R>
static int launch_suffix_r(const SuffixArgs& a, hipStream_t s) {
    // Shapes-only choice: spread one unit over the 4 waves of a workgroup when there are too few
    // units to fill 256 CUs with one wave each (C3-like shapes).
    const int row_chunks = (a.rows + R - 1) / R;
    const bool few_units = (int64_t)a.units * row_chunks < 2 * 256 * 4 && a.kv_len >= 64;
    if (few_units) {
        dim3 grid(a.B, a.Hkv, row_chunks);
        hipLaunchKernelGGL((suffix_attn_kernel<T, D, R, 4>), grid, dim3(256), 0, s, a);
    } else {
        dim3 grid(a.B, (a.Hkv + 3) / 4, row_chunks);
        hipLaunchKernelGGL((suffix_attn_kernel<T, D, R, 1>), grid, dim3(256), 0, s, a);
    }
    return (int)hipGetLastError();
}

template <typename T, int D>
static int launch_suffix_t(const SuffixArgs& a, hipStream_t s) {
    if (a.rows <= 1) return launch_suffix_r<T, D, 1>(a, s);
    if (a.rows <= 2) return launch_suffix_r<T, D, 2>(a, s);
    if (a.rows <= 4) return launch_suffix_r<T, D, 4>(a, s);
    return launch_suffix_r<T, D, 8>(a, s);
}

int launch_suffix(const SuffixArgs& a, int dtype, int D, hipStream_t s) {
    // grouped-query shapes go to the matrix-core kernel (suffix_attn_gqa.hip).  Development builds only
    // (HYD_ABLATION_BUILD): HYD_SUFFIX_IMPL=valu keeps them here, =gqa sends every addressable shape there.
#ifdef HYD_ABLATION_BUILD
    static const int force = [] {
        const char* e = getenv("HYD_SUFFIX_IMPL");
        return !e ? 0 : !strcmp(e, "valu") ? 1 : !strcmp(e, "gqa") ? 2 : 0;
    }();
#else
    constexpr int force = 0;
#endif
    if (force != 1 && suffix_gqa_eligible(a, D, force == 2)) return launch_suffix_gqa(a, dtype, D, s);
    if (dtype == HYD_F16) {
        if (D == 128) return launch_suffix_t<F16, 128>(a, s);
        if (D == 64) return launch_suffix_t<F16, 64>(a, s);
    } else {
        if (D == 128) return launch_suffix_t<BF16, 128>(a, s);
        if (D == 64) return launch_suffix_t<BF16, 64>(a, s);
    }
    return (int)hipErrorInvalidValue;
}

}  // namespace hyd
