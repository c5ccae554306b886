// The matrix-core suffix pass (suffix_attn_gqa.hip: flash.py:163-281 + xformers_stuff.py:189-428 + attention.py:21-43 of the
// reference) as a PERSISTENT wave loop: one wave walks many (sequence, kv head, 16-row chunk) units, and the software
// pipeline over 32-key steps runs ACROSS unit boundaries -- the loads of the next unit's first step (its K fragments, its
// V tile, its query rows and its prefix partial) are issued before the current unit's last step is computed.  Two uses:
//   * stand-alone (suffix_gqa_stream_kernel): grouped-query shapes with many units (C5: 2048 units in a single round of
//     one-wave workgroups paid launch ramp + first round trip + epilogue per unit, 56 us cold for 277 MB);
//   * the streaming role of the co-run kernel (corun_attn.hip), where a workgroup owns a whole CU with one wave per SIMD
//     and nothing but the wave's own pipeline hides latency.  (Measured before writing it: the one-wave-per-unit kernel at
//     4 waves per CU streams within 0-3 % of its 10-waves-per-CU rate; a v_dot2 + DPP wave at one per SIMD does not --
//     every dependent VALU chain is exposed, 5000 cycles per 32 keys.)
//
// Everything a step needs arrives through asm-issued loads hipcc cannot see (so it never waits for them): K fragments
// into a[0:63] (two sets), the query rows into a[64:95] (two sets, one per unit parity), the first prefix partial into
// a[96:129] (two sets), the V tile by LDS-DMA into one of two LDS tiles.  The ONE wait per step is counted:
// vmcnt(<vector-memory instructions issued after the loads of the step about to be computed>) = the next step's loads
// (+ its unit's query / partial loads when it opens a unit) + the previous unit's epilogue stores (asm as well, so the
// count is exact; CDNA4's vmcnt counts stores).  Work comes from an atomic item counter; an item = UPI consecutive
// sequences of one (kv head, row chunk); item boundaries drain the pipeline (lengths and the next item id are ordinary
// loads).
#pragma once
#include "suffix_gqa_common.h"
#include "suffix_stream.h"

namespace hyd {

namespace {

// Query fragment c of staging set SET in a[64 + 16 SET + 4 c : +3]; partial staging set SET: out pieces db in
// a[96 + 18 SET + 2 db : +1] (even-aligned pairs), lse in a[112 + SET].
template <int SET, int C>
struct QStage;
#define HYD_QSTAGE(SET, C, A, B, G, E)                                                                                    \
    template <>                                                                                                           \
    struct QStage<SET, C> {                                                                                               \
        static __device__ __forceinline__ void load(u32x4 rsrc, unsigned voff) {                                          \
            asm volatile("buffer_load_dwordx4 a[" #A ":" #E "], %0, %1, 0 offen offset:%2" ::"v"(voff), "s"(rsrc), "i"(64 * C) \
                         : "memory", "a" #A, "a" #B, "a" #G, "a" #E);                                                     \
        }                                                                                                                 \
        static __device__ __forceinline__ u32x4 read() {                                                                  \
            u32x4 r;                                                                                                      \
            asm volatile("v_accvgpr_read_b32 %0, a" #A "\n\tv_accvgpr_read_b32 %1, a" #B "\n\tv_accvgpr_read_b32 %2, a" #G "\n\tv_accvgpr_read_b32 %3, a" #E \
                         : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]));                                               \
            return r;                                                                                                     \
        }                                                                                                                 \
    };
// clang-format off
HYD_QSTAGE(0, 0, 64, 65, 66, 67) HYD_QSTAGE(0, 1, 68, 69, 70, 71) HYD_QSTAGE(0, 2, 72, 73, 74, 75) HYD_QSTAGE(0, 3, 76, 77, 78, 79)
HYD_QSTAGE(1, 0, 80, 81, 82, 83) HYD_QSTAGE(1, 1, 84, 85, 86, 87) HYD_QSTAGE(1, 2, 88, 89, 90, 91) HYD_QSTAGE(1, 3, 92, 93, 94, 95)
// clang-format on
#undef HYD_QSTAGE

template <int SET, int DB>
struct PStage;
#define HYD_PSTAGE(SET, DB, A, B)                                                                                         \
    template <>                                                                                                           \
    struct PStage<SET, DB> {                                                                                              \
        static __device__ __forceinline__ void load(u32x4 rsrc, unsigned voff) {                                          \
            asm volatile("buffer_load_dwordx2 a[" #A ":" #B "], %0, %1, 0 offen offset:%2" ::"v"(voff), "s"(rsrc), "i"(32 * DB) \
                         : "memory", "a" #A, "a" #B);                                                                     \
        }                                                                                                                 \
        static __device__ __forceinline__ u32x2 read() {                                                                  \
            u32x2 r;                                                                                                      \
            asm volatile("v_accvgpr_read_b32 %0, a" #A "\n\tv_accvgpr_read_b32 %1, a" #B : "=v"(r[0]), "=v"(r[1]));     \
            return r;                                                                                                     \
        }                                                                                                                 \
    };
// clang-format off
HYD_PSTAGE(0, 0, 96, 97) HYD_PSTAGE(0, 1, 98, 99) HYD_PSTAGE(0, 2, 100, 101) HYD_PSTAGE(0, 3, 102, 103)
HYD_PSTAGE(0, 4, 104, 105) HYD_PSTAGE(0, 5, 106, 107) HYD_PSTAGE(0, 6, 108, 109) HYD_PSTAGE(0, 7, 110, 111)
HYD_PSTAGE(1, 0, 114, 115) HYD_PSTAGE(1, 1, 116, 117) HYD_PSTAGE(1, 2, 118, 119) HYD_PSTAGE(1, 3, 120, 121)
HYD_PSTAGE(1, 4, 122, 123) HYD_PSTAGE(1, 5, 124, 125) HYD_PSTAGE(1, 6, 126, 127) HYD_PSTAGE(1, 7, 128, 129)
// clang-format on
#undef HYD_PSTAGE
template <int SET>
struct PLse;
template <>
struct PLse<0> {
    static __device__ __forceinline__ void load(u32x4 rsrc, unsigned voff) {
        asm volatile("buffer_load_dword a112, %0, %1, 0 offen" ::"v"(voff), "s"(rsrc) : "memory", "a112");
    }
    static __device__ __forceinline__ float read() {
        float r;
        asm volatile("v_accvgpr_read_b32 %0, a112" : "=v"(r));
        return r;
    }
};
template <>
struct PLse<1> {
    static __device__ __forceinline__ void load(u32x4 rsrc, unsigned voff) {
        asm volatile("buffer_load_dword a113, %0, %1, 0 offen" ::"v"(voff), "s"(rsrc) : "memory", "a113");
    }
    static __device__ __forceinline__ float read() {
        float r;
        asm volatile("v_accvgpr_read_b32 %0, a113" : "=v"(r));
        return r;
    }
};

template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory"); }

// epilogue stores as asm: the step waits count them
__device__ __forceinline__ void st_f32x4(float* p, f32x4 x) { asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(p), "v"(x) : "memory"); }
__device__ __forceinline__ void st_u32x2(void* p, u32x2 x) { asm volatile("global_store_dwordx2 %0, %1, off\n\ts_nop 1" ::"v"(p), "v"(x) : "memory"); }
__device__ __forceinline__ void st_f32(float* p, float x) { asm volatile("global_store_dword %0, %1, off\n\ts_nop 1" ::"v"(p), "v"(x) : "memory"); }

}  // namespace

struct GqaStreamGeom {
    int32_t upi;      // units (sequences) per item
    int32_t nbr;      // batch ranges = ceil(B / upi)
    int32_t chunks;   // 16-row chunks per (sequence, kv head)
    int32_t n_items;  // nbr * chunks * Hkv
};

// MODE 0: final output (16-bit out, optional suffix-only LSE, prefix partials folded); MODE 1: the normalised fp32 partial
// + natural-log LSE of the unique keys alone (co-run: merged after the launch).  vt0: LDS byte address of this wave's two
// 32-key V tiles (2 * 32 * D * 2 bytes).
template <typename T, int D, int MODE>
__device__ __forceinline__ void gqa_stream_wave(const SuffixArgs& a, const GqaStreamGeom g, unsigned* next_item, const unsigned vt0) {
    using TR = Traits<T>;
    static_assert(D == 128, "head_dim 128");
    constexpr int RB = D * 2, NCH = D / 32, NDB = D / 16;
    constexpr int RPI = 1024 / RB, NVD = 32 / RPI, TILE = 32 * RB;
    constexpr int NKV = 2 * NCH + NVD;                           // K fragments + V DMAs of a step
    constexpr int NOPEN = NCH + (MODE == 0 ? NDB + 1 : 0);       // loads that open a unit: query rows (+ the first partial)
    int lane_ = threadIdx.x & 63;
    asm volatile("" : "+v"(lane_));
    const int lane = lane_;
    const int l15 = lane & 15, g4 = lane >> 4;
    const unsigned k_ts2 = (unsigned)(a.k_ts * 2), v_ts2 = (unsigned)(a.v_ts * 2);
    const unsigned kvoff = (unsigned)l15 * k_ts2 + 16u * g4;
    const int drow = (lane * 16) / RB, dcp = ((lane * 16) % RB) >> 4;
    unsigned vvoff[NVD];
#pragma unroll
    for (int i = 0; i < NVD; ++i) {
        const int r_ = i * RPI + drow;
        const int sw = r_ & 3;
        const int vch = (((dcp >> 2) ^ sw) << 2) | (dcp & 3);
        vvoff[i] = (unsigned)r_ * v_ts2 + (unsigned)vch * 16u;
    }
    const int trow = 4 * g4 + (l15 >> 2);
    const int tsw = trow & 3;
    unsigned vaddr[NDB];
#pragma unroll
    for (int db = 0; db < NDB; ++db) vaddr[db] = vt0 + trow * RB + (((db >> 1) ^ tsw) << 6) + 32 * (db & 1) + 8 * (l15 & 3);
    const float sc = a.scale_log2e;
    const int np = a.n_partials;
    const bool pre = MODE == 0 && np > 0 && !a.partials[0].is_f32;  // the first partial travels with the unit's opening loads
    // whole-tensor resources of q and of the first partial (per-lane row offsets; invalid rows point past the end)
    const size_t nrows = (size_t)a.B * a.nq * a.Hq;
    const u32x4 qrs = make_rsrc_g(a.q, (unsigned)(nrows * RB));
    const u32x4 prs = make_rsrc_g(pre ? a.partials[0].out : a.q, pre ? (unsigned)(nrows * RB) : 0u);
    const u32x4 lrs = make_rsrc_g(pre ? (const void*)a.partials[0].lse : a.q, pre ? (unsigned)(nrows * 4) : 0u);
    float* lse_out = a.lse;  // MODE 0: may be null (then the epilogue has one store fewer: wait_for counts it)

    f32x4 o[NDB];
    float m_run = -INFINITY, l_run = 0.f;
    u32x4 qf[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) qf[c] = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
    for (int db = 0; db < NDB; ++db) o[db] = f32x4{0.f, 0.f, 0.f, 0.f};

    unsigned item = __builtin_amdgcn_readfirstlane(stream_pull(next_item, lane));
    while (item < (unsigned)g.n_items) {
        const unsigned nxt_item = stream_pull(next_item, lane);  // consumed at the end of the item
        const int hk = item % (unsigned)a.Hkv, rest = item / (unsigned)a.Hkv;
        const int rc = rest % g.chunks, br = rest / g.chunks;
        const int b0 = br * g.upi, nb = min(g.upi, a.B - b0), row0 = rc * 16;
        int lv = a.kv_len;
        {
            const int bb = b0 + min(lane, nb - 1);
            if (a.sl32) lv = a.sl32[bb];
            else if (a.sl64) lv = (int)a.sl64[bb];
            lv = max(0, min(lv, a.kv_len));
        }
        auto len_of = [&](int j) -> int { return __builtin_amdgcn_readlane(lv, j); };
        // this lane's query row inside a unit, and its row index relative to the sequence
        const int row = row0 + l15;
        const bool rvalid = row < a.rows;
        const int iq = a.nq == 1 ? 0 : (rvalid ? row / a.g : 0), gq = a.nq == 1 ? (rvalid ? row : 0) : (rvalid ? row % a.g : 0);
        const int64_t rrel = (int64_t)iq * a.Hq + hk * a.g + gq;  // ridx = b * nq * Hq + rrel
        auto ridx_of = [&](int b) -> int64_t { return (int64_t)b * a.nq * a.Hq + rrel; };

        // ---- producer: issue the loads of the next step; returns what the consumer needs to know about it -------------
        struct Desc {
            int j, key0, len;
            bool valid, first, last;
        };
        int pj = 0, pkey = 0;
        u32x4 krs = qrs, vrs = qrs;
        auto produce = [&](auto BUF_) __attribute__((always_inline)) -> Desc {
            constexpr int BUF = decltype(BUF_)::value;
            Desc d;
            d.valid = pj < nb;
            d.j = pj;
            d.key0 = pkey;
            d.len = 0;
            d.first = pkey == 0;
            d.last = true;
            if (d.valid) {
                const int len = len_of(pj);
                const int b = b0 + pj;
                d.len = len;
                d.last = pkey + 32 >= len;
                if (d.first) {
                    krs = make_rsrc_g(static_cast<const uint16_t*>(a.k) + (int64_t)b * a.k_bs + (int64_t)hk * a.k_hs, (unsigned)len * k_ts2);
                    vrs = make_rsrc_g(static_cast<const uint16_t*>(a.v) + (int64_t)b * a.v_bs + (int64_t)hk * a.v_hs, (unsigned)len * v_ts2);
                    const int64_t ridx = ridx_of(b);
                    const unsigned qv = rvalid ? (unsigned)(ridx * RB) + 16u * g4 : 0xfffffff0u;
                    const unsigned pv = rvalid ? (unsigned)(ridx * RB) + 8u * g4 : 0xfffffff0u;
                    const unsigned lvo = rvalid ? (unsigned)(ridx * 4) : 0xfffffff0u;
                    if (pj & 1) {
                        static_for_g<NCH>([&](auto C_) { QStage<1, decltype(C_)::value>::load(qrs, qv); });
                        if constexpr (MODE == 0) {
                            static_for_g<NDB>([&](auto B_) { PStage<1, decltype(B_)::value>::load(prs, pv); });
                            PLse<1>::load(lrs, lvo);
                        }
                    } else {
                        static_for_g<NCH>([&](auto C_) { QStage<0, decltype(C_)::value>::load(qrs, qv); });
                        if constexpr (MODE == 0) {
                            static_for_g<NDB>([&](auto B_) { PStage<0, decltype(B_)::value>::load(prs, pv); });
                            PLse<0>::load(lrs, lvo);
                        }
                    }
                }
                const unsigned ks0 = (unsigned)pkey * k_ts2, ks1 = ks0 + 16u * k_ts2, vsoff = (unsigned)pkey * v_ts2;
                static_for_g<NCH>([&](auto C_) {
                    constexpr int c = decltype(C_)::value;
                    KReg<BUF * 8 + c>::template load<64 * c>(krs, kvoff, __builtin_amdgcn_readfirstlane(ks0));
                    KReg<BUF * 8 + 4 + c>::template load<64 * c>(krs, kvoff, __builtin_amdgcn_readfirstlane(ks1));
                });
#pragma unroll
                for (int i = 0; i < NVD; ++i) dma16_g(vrs, vvoff[i], __builtin_amdgcn_readfirstlane(vsoff), vt0 + BUF * TILE + i * 1024);
                if (d.last) { ++pj; pkey = 0; }
                else pkey += 32;
            }
            return d;
        };

        // ---- consumer: one 32-key step of the current unit out of register set / LDS tile BUF ----------------------------
        float pre_lse = -INFINITY;
        u32x2 pre_u[NDB];
#pragma unroll
        for (int db = 0; db < NDB; ++db) pre_u[db] = u32x2{0u, 0u};
        auto consume = [&](auto BUF_, const Desc& d) __attribute__((always_inline)) -> bool {  // returns: epilogue stores were issued
            constexpr int BUF = decltype(BUF_)::value;
            if (d.first) {
                m_run = -INFINITY;
                l_run = 0.f;
#pragma unroll
                for (int db = 0; db < NDB; ++db) o[db] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (d.j & 1) {
                    static_for_g<NCH>([&](auto C_) { qf[decltype(C_)::value] = QStage<1, decltype(C_)::value>::read(); });
                    if constexpr (MODE == 0) {
                        static_for_g<NDB>([&](auto B_) { pre_u[decltype(B_)::value] = PStage<1, decltype(B_)::value>::read(); });
                        pre_lse = PLse<1>::read();
                    }
                } else {
                    static_for_g<NCH>([&](auto C_) { qf[decltype(C_)::value] = QStage<0, decltype(C_)::value>::read(); });
                    if constexpr (MODE == 0) {
                        static_for_g<NDB>([&](auto B_) { pre_u[decltype(B_)::value] = PStage<0, decltype(B_)::value>::read(); });
                        pre_lse = PLse<0>::read();
                    }
                }
            }
            if (d.len > 0) {
                f32x4 s0, s1;
                // the query fragments may have just been written by v_accvgpr_read (VALU): two wait states before a
                // matrix-core read of the same VGPRs, which hipcc does not insert in front of asm
                asm volatile("s_nop 1" ::: "memory");
                static_for_g<NCH>([&](auto C_) {
                    constexpr int c = decltype(C_)::value;
                    KReg<BUF * 8 + c>::template qk<T, c == 0>(s0, qf[c]);
                    KReg<BUF * 8 + 4 + c>::template qk<T, c == 0>(s1, qf[c]);
                });
                asm volatile("s_nop 7\n\ts_nop 7" : "+v"(s0), "+v"(s1));
                float p[8];
                const int kb = d.key0 + 4 * g4;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    p[i] = (kb + i < d.len) ? s0[i] * sc : -INFINITY;
                    p[4 + i] = (kb + 16 + i < d.len) ? s1[i] * sc : -INFINITY;
                }
                float tmax = fmaxf(fmaxf(fmaxf(p[0], p[1]), fmaxf(p[2], p[3])), fmaxf(fmaxf(p[4], p[5]), fmaxf(p[6], p[7])));
                tmax = quad_max(tmax);
                const float m_new = fmaxf(m_run, tmax);  // finite: key0 < len guarantees one valid key per row
                const float alpha = fast_exp2(m_run - m_new);
                float ps = 0.f;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    p[i] = fast_exp2(p[i] - m_new);
                    ps += p[i];
                }
                ps = quad_sum(ps);
                l_run = l_run * alpha + ps;
                m_run = m_new;
                const u32x4 pf = {TR::pack2(p[0], p[1]), TR::pack2(p[2], p[3]), TR::pack2(p[4], p[5]), TR::pack2(p[6], p[7])};
#pragma unroll
                for (int db = 0; db < NDB; ++db) {
                    const u32x2 t0 = lds_tr16_g(vaddr[db] + BUF * TILE);
                    const u32x2 t1 = lds_tr16_g(vaddr[db] + BUF * TILE + 16 * RB);
                    const u32x4 vf = {t0[0], t0[1], t1[0], t1[1]};
                    o[db] *= alpha;
                    mfma16_acc<T>(o[db], vf, pf);
                }
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 7\n\ts_nop 7" ::: "memory");
            }
            if (!d.last) return false;
            // ---- the unit is complete: fold the prefix partials (MODE 0), normalise, store ---------------------------------
            const int64_t ridx = ridx_of(b0 + d.j);
            const float m_s = m_run, l_s = l_run;  // the suffix-only state: the LSE output is the suffix pass's own
            if constexpr (MODE == 0) {
                auto fold = [&](float lse_p, const f32x4(&x)[NDB]) __attribute__((always_inline)) {
                    const float m_p = lse_p * kLog2e;
                    const float mf = fmaxf(m_run, m_p);
                    const float ms = (mf == -INFINITY) ? 0.f : mf;
                    const float a1 = fast_exp2(m_run - ms), a2 = fast_exp2(m_p - ms);
                    l_run = l_run * a1 + a2;
                    m_run = mf;
#pragma unroll
                    for (int db = 0; db < NDB; ++db) o[db] = o[db] * a1 + x[db] * a2;
                };
                if (pre) {
                    f32x4 x[NDB];
#pragma unroll
                    for (int db = 0; db < NDB; ++db)
                        x[db] = f32x4{TR::lo(pre_u[db][0]), TR::hi(pre_u[db][0]), TR::lo(pre_u[db][1]), TR::hi(pre_u[db][1])};
                    fold(rvalid ? pre_lse : -INFINITY, x);
                }
                // further partials (split-KV slices, more levels): ordinary loads -- hipcc waits for them with vmcnt(0),
                // which also drains the next step's loads; such shapes pay one round trip per unit here
                for (int i = pre ? 1 : 0; i < np; ++i) {
                    const PartialDev& pd = a.partials[i];
                    f32x4 x[NDB];
                    float lse_p = -INFINITY;
                    if (rvalid) {
                        lse_p = pd.lse[ridx];
                        if (pd.is_f32) {
#pragma unroll
                            for (int db = 0; db < NDB; ++db)
                                x[db] = *reinterpret_cast<const f32x4*>(static_cast<const float*>(pd.out) + ridx * D + 16 * db + 4 * g4);
                        } else {
#pragma unroll
                            for (int db = 0; db < NDB; ++db) {
                                const u32x2 u = *reinterpret_cast<const u32x2*>(static_cast<const uint16_t*>(pd.out) + ridx * D + 16 * db + 4 * g4);
                                x[db] = f32x4{TR::lo(u[0]), TR::hi(u[0]), TR::lo(u[1]), TR::hi(u[1])};
                            }
                        }
                    } else {
#pragma unroll
                        for (int db = 0; db < NDB; ++db) x[db] = f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                    fold(lse_p, x);
                }
            }
            const float inv = l_run > 0.f ? 1.0f / l_run : 0.f;
            const float lse_s = l_s > 0.f ? m_s * kLn2 + __logf(l_s) : -INFINITY;
            if (rvalid) {
#pragma unroll
                for (int db = 0; db < NDB; ++db) {
                    const f32x4 x = o[db] * inv;
                    if constexpr (MODE == 1) {
                        st_f32x4(static_cast<float*>(a.out) + ridx * D + 16 * db + 4 * g4, x);
                    } else {
                        const u32x2 pk = {TR::pack2(x[0], x[1]), TR::pack2(x[2], x[3])};
                        st_u32x2(static_cast<uint16_t*>(a.out) + ridx * D + 16 * db + 4 * g4, pk);
                    }
                }
                if (lse_out && g4 == 0) st_f32(lse_out + ridx, lse_s);  // one store instruction (lanes of key group 0)
            }
            return true;
        };

        // ---- the pipeline over the item's steps: step s is computed while step s+1 is in flight ---------------------------
        const bool has_lse_store = lse_out != nullptr;
        auto wait_for = [&](const Desc& nx, bool stores_before) __attribute__((always_inline)) {
            // instructions issued after the loads of the step about to be computed: the previous step's epilogue stores
            // (if it closed a unit), then the next step's loads (+ its unit's opening loads)
            const int ns = stores_before ? NDB + (has_lse_store ? 1 : 0) : 0;
            if (!nx.valid) {
                // nothing was issued for a next step: only the stores are younger
                if (ns == NDB + 1) wait_vm<NDB + 1>();
                else if (ns == NDB) wait_vm<NDB>();
                else wait_vm<0>();
                return;
            }
            const bool open = nx.first;
            if (open) {
                if (ns == NDB + 1) wait_vm<NKV + NOPEN + NDB + 1>();
                else if (ns == NDB) wait_vm<NKV + NOPEN + NDB>();
                else wait_vm<NKV + NOPEN>();
            } else {
                if (ns == NDB + 1) wait_vm<NKV + NDB + 1>();
                else if (ns == NDB) wait_vm<NKV + NDB>();
                else wait_vm<NKV>();
            }
        };
        {
            using std::integral_constant;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // lengths are in; nothing of the previous item is in flight
            Desc cur = produce(integral_constant<int, 0>{});
            bool stores = false;
            while (cur.valid) {
                Desc nx = produce(integral_constant<int, 1>{});
                wait_for(nx, stores);
                stores = consume(integral_constant<int, 0>{}, cur);
                cur = nx;
                if (!cur.valid) break;
                nx = produce(integral_constant<int, 0>{});
                wait_for(nx, stores);
                stores = consume(integral_constant<int, 1>{}, cur);
                cur = nx;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        item = __builtin_amdgcn_readfirstlane(nxt_item);
    }
}

}  // namespace hyd
