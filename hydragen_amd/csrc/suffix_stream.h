// The suffix pass as a STREAMING ROLE for persistent, low-occupancy workgroups (one wave per SIMD): what the co-run
// kernel (corun_attn.hip) runs on the CUs the prefix role does not hold.  Same operator as suffix_attn.hip for one query
// row per unit (nq == 1, Hq == Hkv): flash.py:163-281 of the reference, one wave per (sequence, kv head) unit, D/8 lanes
// per key row, v_dot2 + DPP scores, per-lane-group online softmax, lane groups merged at the end of the unit.
//
// suffix_attn.hip hides HBM latency with occupancy (24 waves per CU, 8 KiB in flight each).  A workgroup that shares its
// kernel with the prefix role owns a whole CU with 4 waves, so here the bytes in flight come from the wave itself: a unit's
// keys are cut into chunks of CKI x 4 keys; a chunk's K rows, V rows and the query row travel by LDS-DMA
// (buffer_load ... lds, 1 KiB per wave instruction, 2 CKI + 1 instructions per chunk) into one of NBUF slots of the wave's
// own LDS ring, and the DMAs of chunk c + NBUF - 1 are issued before chunk c is computed -- across unit boundaries, so
// the stream never drains inside an item.  Why LDS and not registers: loads hipcc can see are waited for by hipcc, and
// its wait-count pass put `s_waitcnt vmcnt(3..7)` in front of the address arithmetic of the NEXT chunk's loads (a false
// write-after-write on the rotating register sets), which serialised every chunk behind a full HBM round trip (S = 16:
// 98 us against 47 us).  DMAs issued from asm are invisible to that pass, have no register destination that could be
// copied or reused under them, and are waited for by ONE counted `s_waitcnt vmcnt((NBUF - 1)(2 CKI + 1))` per chunk.
// Buffer resources cover exactly the unit's keys: rows past seq_len zero-fill without touching memory.
// Work is handed out by an atomic counter: an item = UPI consecutive sequences of one kv head; the four waves of a
// workgroup pull neighbouring heads of the same sequences (one contiguous 1 KiB piece per token), and the NEXT item id
// is fetched while the current one is processed.
#pragma once
#include "suffix_common.h"

namespace hyd {

struct StreamGeom {
    int32_t upi;      // units (sequences) per item
    int32_t nbr;      // batch ranges = ceil(B / upi)
    int32_t n_items;  // nbr * 4 * ceil(Hkv / 4)
};

template <int CKI, int NBUF>
constexpr int stream_wave_lds_bytes() { return NBUF * (2 * CKI + 1) * 1024; }

__device__ __forceinline__ unsigned stream_pull(unsigned* ctr, int lane) {
    unsigned v = 0;
    if (lane == 0) v = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return v;  // valid in lane 0; readfirstlane it where it is consumed
}

// one LDS-DMA instruction (see prefix_unit_w64.h: dma16w); M0 is written inside the statement that reads it
__device__ __forceinline__ void stream_dma(u32x4 rsrc, unsigned voff, unsigned soff, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds"
                 :
                 : "v"(voff), "s"(rsrc), "s"(soff), "s"(lds_dst)
                 : "memory");
}
__device__ __forceinline__ u32x4 stream_rsrc(const char* base, unsigned bytes) {
    const uint64_t b = (uint64_t)(uintptr_t)base;
    u32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((unsigned)b);
    r[1] = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32) & 0xffffu);
    r[2] = __builtin_amdgcn_readfirstlane(bytes);
    r[3] = 0x00020000u;
    return r;
}

typedef __attribute__((address_space(1))) float* gf32_p;
typedef __attribute__((address_space(1))) f32x4* gf32x4_p;
typedef const __attribute__((address_space(3))) u32x4* lds_u32x4_cp;

// One wave: pull items until the counter runs past n_items.  `ring` = LDS byte address of this wave's own
// stream_wave_lds_bytes<CKI, NBUF>() bytes.  OUT_F32: write the normalised fp32 partial + natural-log LSE of the unique
// keys alone (merged later); otherwise finish the row like suffix_attn.hip (partials merged, 16-bit output).
template <typename T, int D, int CKI, int NBUF, bool OUT_F32>
__device__ __forceinline__ void suffix_stream_wave(const SuffixArgs& a, const StreamGeom g, unsigned* next_item, const unsigned ring) {
    using TR = Traits<T>;
    static_assert(D == 128, "one 16-lane group per key row");
    constexpr int LPK = D / 8, KPI = 64 / LPK;
    constexpr int NDMA = 2 * CKI + 1;          // wave instructions per chunk
    constexpr unsigned SLOT = NDMA * 1024u;    // K rows | V rows | query row (x4)
    int lane_ = threadIdx.x & 63;
    asm volatile("" : "+v"(lane_));
    const int lane = lane_;
    const int sub = lane % LPK, ks = lane / LPK;
    const unsigned krs = (unsigned)(a.k_ts * 2), vrs = (unsigned)(a.v_ts * 2);  // token strides in bytes
    const unsigned klane = (unsigned)ks * krs + sub * 16, vlane = (unsigned)ks * vrs + sub * 16, qlane = sub * 16;
    const float sc = a.scale_log2e;
    const unsigned lds_lane = ring + lane * 16;

    float m = -INFINITY, l = 0.f, acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;

    unsigned item = __builtin_amdgcn_readfirstlane(stream_pull(next_item, lane));
    while (item < (unsigned)g.n_items) {
        const unsigned nxt = stream_pull(next_item, lane);  // consumed at the end of the item
        const int hlo = item & 3, rest = item >> 2;
        const int hgrp = rest / g.nbr, br = rest - hgrp * g.nbr;
        const int h = hgrp * 4 + hlo;
        const int b0 = br * g.upi;
        const int nb = min(g.upi, a.B - b0);
        if (h < a.Hkv) {
            // lengths of the item's units, one per lane
            int lv = a.kv_len;
            {
                const int bb = b0 + min(lane, nb - 1);
                if (a.sl32) lv = a.sl32[bb];
                else if (a.sl64) lv = (int)a.sl64[bb];
                lv = max(0, min(lv, a.kv_len));
            }
            auto len_of = [&](int j) -> int { return __builtin_amdgcn_readlane(lv, j); };
            // Producer and consumer walk the same chunk sequence: unit j of the item, first key iteration `it` of the
            // chunk.  A unit has max(1, ceil(len / (CKI * KPI))) chunks (an empty unit still gets its epilogue).
            int pj = 0, pit = 0, cj = 0, cit = 0;
            unsigned pslot = 0, cslot = 0;  // slot = chunk count mod NBUF
            auto issue = [&]() __attribute__((always_inline)) {
                const bool has = pj < nb;
                const int j = has ? pj : 0;
                const int b = b0 + j;
                const int len = has ? len_of(j) : 0;
                const char* kp = reinterpret_cast<const char*>(static_cast<const uint16_t*>(a.k) + (int64_t)b * a.k_bs + (int64_t)h * a.k_hs);
                const char* vp = reinterpret_cast<const char*>(static_cast<const uint16_t*>(a.v) + (int64_t)b * a.v_bs + (int64_t)h * a.v_hs);
                const char* qp = reinterpret_cast<const char*>(static_cast<const uint16_t*>(a.q) + ((int64_t)b * a.Hq + h) * D);
                // resources end at the unit's last key: rows past it (and every row once the item is exhausted) zero-fill
                const u32x4 kr = stream_rsrc(kp, len > 0 ? (unsigned)(len - 1) * krs + D * 2 : 0u);
                const u32x4 vr = stream_rsrc(vp, len > 0 ? (unsigned)(len - 1) * vrs + D * 2 : 0u);
                const u32x4 qr = stream_rsrc(qp, has ? D * 2 : 0u);
                const unsigned dst = __builtin_amdgcn_readfirstlane(ring + pslot * SLOT);
                const unsigned k0 = (unsigned)(pit * KPI) * krs, v0 = (unsigned)(pit * KPI) * vrs;
#pragma unroll
                for (int u = 0; u < CKI; ++u) {
                    stream_dma(kr, klane, __builtin_amdgcn_readfirstlane(k0 + (unsigned)(u * KPI) * krs), dst + u * 1024);
                    stream_dma(vr, vlane, __builtin_amdgcn_readfirstlane(v0 + (unsigned)(u * KPI) * vrs), dst + (CKI + u) * 1024);
                }
                stream_dma(qr, qlane, 0u, dst + 2 * CKI * 1024);
                pslot = pslot + 1 == NBUF ? 0 : pslot + 1;
                if (has) {
                    if ((pit + CKI) * KPI >= len) { ++pj; pit = 0; }
                    else pit += CKI;
                }
            };
            auto compute = [&]() __attribute__((always_inline)) {
                const int b = b0 + cj;
                const int len = len_of(cj);
                const bool last = (cit + CKI) * KPI >= len;
                if (cit == 0) {
                    m = -INFINITY;
                    l = 0.f;
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
                }
                const unsigned src = lds_lane + cslot * SLOT;
                auto lds16 = [&](int piece) -> u32x4 { return *(lds_u32x4_cp)(uintptr_t)(src + piece * 1024); };
                const u32x4 qv = lds16(2 * CKI);
                float s[CKI];
#pragma unroll
                for (int u = 0; u < CKI; ++u) {
                    const u32x4 kv = lds16(u);
                    float d = 0.f;
#pragma unroll
                    for (int i = 0; i < 4; ++i) d = TR::dot2(qv[i], kv[i], d);
                    d = group_sum<LPK>(d);
                    s[u] = ((cit + u) * KPI + ks < len) ? d * sc : -INFINITY;
                }
                float cmax = s[0];
#pragma unroll
                for (int u = 1; u < CKI; ++u) cmax = fmaxf(cmax, s[u]);
                const float mnew = fmaxf(m, cmax);
                const float ms = (mnew == -INFINITY) ? 0.f : mnew;
                const float alpha = fast_exp2(m - ms);
                float ps = 0.f;
#pragma unroll
                for (int u = 0; u < CKI; ++u) {
                    s[u] = fast_exp2(s[u] - ms);
                    ps += s[u];
                }
                l = l * alpha + ps;
                m = mnew;
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] *= alpha;
#pragma unroll
                for (int u = 0; u < CKI; ++u) {
                    float vf[8];
                    widen8<T>(lds16(CKI + u), vf);
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[j] = __builtin_fmaf(s[u], vf[j], acc[j]);
                }
                if (last) {
                    merge_lane_groups<LPK>(m, l, acc);
                    const int64_t ridx = (int64_t)b * a.Hq + h;
                    if (ks == 0) {
                        if constexpr (OUT_F32) {
                            const float inv = l > 0.f ? 1.0f / l : 0.f;
                            gf32x4_p po = (gf32x4_p)(uintptr_t)(static_cast<float*>(a.out) + ridx * D + sub * 8);
                            po[0] = f32x4{acc[0] * inv, acc[1] * inv, acc[2] * inv, acc[3] * inv};
                            po[1] = f32x4{acc[4] * inv, acc[5] * inv, acc[6] * inv, acc[7] * inv};
                            if (sub == 0) *(gf32_p)(uintptr_t)(a.lse + ridx) = l > 0.f ? m * kLn2 + __logf(l) : -INFINITY;
                        } else {
                            finish_row<T, D>(a, ridx, sub, m, l, acc, false, 0.f, u32x4{0u, 0u, 0u, 0u});
                        }
                    }
                    ++cj;
                    cit = 0;
                } else {
                    cit += CKI;
                }
                cslot = cslot + 1 == NBUF ? 0 : cslot + 1;
            };
            // chunk c is computed while chunks c+1 .. c+NBUF-1 are in flight
#pragma unroll
            for (int i = 0; i < NBUF - 1; ++i) issue();
            while (cj < nb) {
                issue();
                asm volatile("s_waitcnt vmcnt(%0)" ::"i"((NBUF - 1) * NDMA) : "memory");
                compute();
            }
            // the NBUF - 1 DMAs still in flight were issued past the item's end (null resources, zero fill): let them
            // land before the ring is reused
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        item = __builtin_amdgcn_readfirstlane(nxt);
    }
}

}  // namespace hyd
