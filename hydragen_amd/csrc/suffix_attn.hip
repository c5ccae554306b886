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

#include <hip/hip_ext.h>

#include "suffix_common.h"

namespace hyd {

// Packed path for SHORT sequences with one query row per unit (decode with Hq == Hkv, nq == 1 -- C2): a LANE GROUP of
// D/8 lanes owns one (sequence, kv head) unit, so a wave works on 64 / (D/8) consecutive kv heads of one sequence (4 at
// D = 128) and walks their keys together: one wave instruction fetches one token's K (or V) rows of those heads = 1 KiB
// contiguous.  The one-unit-per-wave path spends ~300 instructions per unit outside its key loop (three quarters of a
// load instruction idle at short lengths, two rounds of cross-lane-group merges, a per-wave epilogue); at C2 that is
// 32768 waves and 16 us of pure issue time, the whole cost of a step with a short suffix (22 us at S = 1 for 6 us of
// HBM traffic).  Here a lane group keeps its own (m, l, acc): no cross-group merge at all, a quarter of the waves
// (measured at C2, fused entry: S = 1 22.2 -> 7.4 us, S = 4 22.4 -> 16.0 us, S = 8 29.9 -> 27.0 us; from S = 16 on the
// one-unit-per-wave path streams 2-4 % faster -- more loads in flight -- so the choice is made per sequence at RUN time
// from its length, inside the one kernel the shapes select: capture-safe, no device read on the host).
constexpr int kPackedMaxLen = 12;
template <typename T, int D, int NPRE>
__device__ __forceinline__ void suffix_packed_body(const SuffixArgs& a, int b, int ygroup, int len) {
    using TR = Traits<T>;
    constexpr int LPK = D / 8;     // lanes per unit
    constexpr int HPW = 64 / LPK;  // units (kv heads) per wave
    constexpr int U = 2;           // keys in flight per wave and tensor (x 1 KiB); sequences here are at most kPackedMaxLen long
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int sub = lane % LPK, hg = lane / LPK;
    const int h0 = (ygroup * 4 + wave) * HPW;  // first head of this wave
    if (h0 >= a.Hkv) return;
    const int hk = h0 + hg;
    const bool hvalid = hk < a.Hkv;
    const int hkc = hvalid ? hk : a.Hkv - 1;  // idle lane groups shadow the last head and never store

    const int64_t ridx = (int64_t)b * a.Hq + hkc;  // nq == 1, g == 1: [B, 1, Hq]
    const u32x4 qp = *reinterpret_cast<const u32x4*>(static_cast<const uint16_t*>(a.q) + ridx * D + sub * 8);
    // the first 16-bit partials (the usual single prefix level; a second level), fetched under the K/V stream
    const int npre = min(n_prefetched(a), NPRE);
    PrePartials<NPRE> pp;
    prefetch_partials(a, npre, ridx, sub, D, pp);

    // wave-uniform sequence base (scalar registers) + per-lane 32-bit byte offset (head, dims) -> SADDR-form loads
    const gchar_p kbu = uniform_ptr(reinterpret_cast<const char*>(static_cast<const uint16_t*>(a.k) + (int64_t)b * a.k_bs));
    const gchar_p vbu = uniform_ptr(reinterpret_cast<const char*>(static_cast<const uint16_t*>(a.v) + (int64_t)b * a.v_bs));
    const unsigned klane = (unsigned)(hkc * a.k_hs * 2 + sub * 16), vlane = (unsigned)(hkc * a.v_hs * 2 + sub * 16);
    const unsigned krs = (unsigned)(a.k_ts * 2), vrs = (unsigned)(a.v_ts * 2);  // token stride in bytes

    float m = -INFINITY, l = 0.f, acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    const float sc = a.scale_log2e;
    auto chunk = [&](auto UU_C, int t0) __attribute__((always_inline)) {
        constexpr int UU = decltype(UU_C)::value;
        u32x4 kreg[UU], vreg[UU];
#pragma unroll
        for (int u = 0; u < UU; ++u) {
            // never predicate the loads: clamp to the last valid key, its score is forced to -inf below
            const unsigned tc = (unsigned)min(t0 + u, len - 1);
            kreg[u] = __builtin_nontemporal_load((gu32x4_p)(kbu + (tc * krs + klane)));
            vreg[u] = __builtin_nontemporal_load((gu32x4_p)(vbu + (tc * vrs + vlane)));
        }
        float s[UU];
#pragma unroll
        for (int u = 0; u < UU; ++u) {
            float d = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) d = TR::dot2(qp[i], kreg[u][i], d);
            d = group_sum<LPK>(d);
            s[u] = (t0 + u < len) ? d * sc : -INFINITY;  // wave-uniform condition
        }
        float cmax = s[0];
#pragma unroll
        for (int u = 1; u < UU; ++u) cmax = fmaxf(cmax, s[u]);
        const float mnew = fmaxf(m, cmax);  // finite: t0 < len
        const float alpha = fast_exp2(m - mnew);
        float ps = 0.f;
#pragma unroll
        for (int u = 0; u < UU; ++u) {
            s[u] = fast_exp2(s[u] - mnew);
            ps += s[u];
        }
        l = l * alpha + ps;
        m = mnew;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] *= alpha;
#pragma unroll
        for (int u = 0; u < UU; ++u) {
            float vf[8];
            widen8<T>(vreg[u], vf);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = __builtin_fmaf(s[u], vf[j], acc[j]);
        }
    };
    int t = 0;
    for (; t + U <= len; t += U) chunk(std::integral_constant<int, U>{}, t);
    for (; t < len; ++t) chunk(std::integral_constant<int, 1>{}, t);

    if (hvalid) finish_row<T, D, 2, NPRE>(a, ridx, sub, m, l, acc, npre, pp);
}

// Token-row form of the one-query-row decode shape (nq == 1, Hq == Hkv, Hkv a multiple of the 64 / (D / 8) heads one wave
// instruction covers -- C2 and its tensor-parallel shards): a wave walks the token rows of ONE sequence for HPI neighbouring
// heads.  A lane group of D / 8 lanes owns one head outright (all of its keys arrive in the same lanes): one wave instruction
// fetches 1 KB contiguous, the waves of a workgroup sit side by side on the token row (4 KB contiguous per token and tensor),
// UT tokens x 2 tensors (16 KB at UT = 8) per wave.  No merge across lane groups or waves at the end, and a quarter
// of the waves (wave starts, page touches, epilogues, q / partial / output rows of 256 B) of the one-unit-per-wave kernel
// further down, which splits the keys of ONE head over a wave's four lane groups.
// Measured at C2 (profiles/r06_suffix_rows_*.txt; same box, same arena, alternating), first form (all of a chunk's loads,
// then all of its arithmetic): 165 vs 173 us at S = 64, 318 vs 337 at S = 128, 87.6 vs 90.8 at S = 32, equal at S <= 16; UT = 8
// beats 4 / 12 / 16, two heads per lane group, a second buffer, eight-wave workgroups and head-major launch order all measured
// equal or worse.
// Shapes with fewer than 4 waves per sequence put 4 / wps sequences into one workgroup (a.rows_wps_log2).
//
// The product form rotates the two register sets instead of doubling them: K of chunk c + 1 is requested as soon as the scores
// of chunk c are out of the K registers, V of chunk c + 1 as soon as P.V of chunk c is out of the V registers -- the registers
// of the single-buffer form (116 VGPRs, 4 waves per SIMD), but a wave always has 8 KB in flight while it computes.  The last,
// partial chunk rides the same pipeline with clamped token indices (never a predicated load).  The sequence's length travels
// as a VECTOR load in front of q and the prefetched partial (a scalar load's lgkmcnt(0) would serialise it with every later
// kernel-argument fetch), the argument block's four scalar-cache lines are touched at once, and the lane offsets are computed
// so that the first K request does not wait for the partial's LSE (see `khg`).  Measured against the first form, one process,
// alternating (profiles/r06_suffix_rows_pipelined_ab.txt): S = 8 28.5 -> 27.9 us, S = 16 50.3 -> 48.5, S = 32 92.0 -> 90.2,
// S = 64 175.8 -> 172.8 (-1.7 %), S = 128 340 -> 337; equal at S <= 4.  Requesting (half of) chunk 0's K BEFORE the length is
// known (HS > 0, development builds) adds nothing on top (the launch is throughput-bound, not start-latency-bound) and costs
// 1.5 us at S = 1..2 (rows past the length are fetched for nothing): not shipped.
template <typename T, int D, int UT, int NPRE, int HS = 0, int TS = 1>
__global__ __launch_bounds__(256, 4) void suffix_attn_rows_kernel(const SuffixArgs a) {
    using TR = Traits<T>;
    warm_kernargs_256();  // the fields in front of partials[1] span four scalar-cache lines: one miss time instead of five in a row
    constexpr int LPK = D / 8, HPI = 64 / LPK;  // lanes per head row, heads per wave instruction
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int sub = lane % LPK, hg = lane / LPK;
    static_assert(TS == 1 || HS == 0, "token split: no blind requests");
    // waves per sequence inside a workgroup: 4 (then blockIdx.y walks further head slices), 2 or 1
    // TS > 1 (shapes with too few waves to fill the chip: a TP rank's shard, a small batch): TS waves share a (sequence, head slice)
    // and deal its 8-token chunks round-robin -- wave t takes chunks t, t + TS, ... -- then hand their (m, l, acc) to wave 0 through LDS.
    constexpr int TL = TS == 4 ? 2 : TS == 2 ? 1 : 0;
    const int ts_id = wave & (TS - 1), wrest = wave >> TL;
    const int wl = a.rows_wps_log2;
    const int bslot = (int)(blockIdx.x << (2 - wl - TL)) + (wrest >> wl);
    const int h0 = (int)((blockIdx.y << wl) + (wrest & ((1 << wl) - 1))) * HPI;  // first head of this wave
    if (bslot >= a.B || h0 >= a.Hkv) return;  // (all TS waves of a group leave together: a barrier counts the waves that are left)

    // the length as a vector load: every lane the same address; an opaque zero keeps hipcc from making it a scalar load
    int zero = 0;
    asm volatile("" : "+v"(zero));
    // dispatch slot -> sequence: the caller's schedule (hyd_suffix_params.seq_order: longest first when lengths are ragged) or the index
    const int b = a.order ? __builtin_amdgcn_readfirstlane(a.order[bslot + zero]) : bslot;
    int lenv = a.kv_len;
    if (a.sl32) lenv = a.sl32[b + zero];
    else if (a.sl64) lenv = (int)a.sl64[b + zero];

    const int64_t ridx = (int64_t)b * a.Hq + h0 + hg;  // nq == 1, g == 1: [B, 1, Hq]
    const u32x4 qp = *reinterpret_cast<const u32x4*>(static_cast<const uint16_t*>(a.q) + ridx * D + sub * 8);
    PrePartials<NPRE> pp;
    const int npre = min(n_prefetched(a), NPRE);
    prefetch_partials(a, npre, ridx, sub, D, pp);

    // wave-uniform base (scalar registers) + per-lane 32-bit byte offset (head, dims) -> SADDR-form loads
    const gchar_p kbu = uniform_ptr(reinterpret_cast<const char*>(static_cast<const uint16_t*>(a.k) + (int64_t)b * a.k_bs + (int64_t)h0 * a.k_hs));
    const gchar_p vbu = uniform_ptr(reinterpret_cast<const char*>(static_cast<const uint16_t*>(a.v) + (int64_t)b * a.v_bs + (int64_t)h0 * a.v_hs));
    // (product and sum kept apart: fused, hipcc emits a 64-bit multiply-add whose unused high addend lands in the register the
    // partial's LSE is being loaded into, and the first K request waits for that load)
    unsigned khg = (unsigned)hg * (unsigned)(a.k_hs * 2), vhg = (unsigned)hg * (unsigned)(a.v_hs * 2);
    asm volatile("" : "+v"(khg), "+v"(vhg));
    const unsigned klane = khg + sub * 16, vlane = vhg + sub * 16;
    const unsigned krs = (unsigned)(a.k_ts * 2), vrs = (unsigned)(a.v_ts * 2);  // token stride in bytes

    u32x4 kreg[UT], vreg[UT];
    if constexpr (HS > 0) {  // development builds: K of the first HS token rows before the length is known (the launcher checks the cache holds them)
#pragma unroll
        for (int u = 0; u < HS; ++u) kreg[u] = __builtin_nontemporal_load((gu32x4_p)(kbu + ((unsigned)u * krs + klane)));
        __builtin_amdgcn_sched_barrier(0);
    }

    float m = -INFINITY, l = 0.f, acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    const float sc = a.scale_log2e;

    const int len = max(0, min(__builtin_amdgcn_readfirstlane(lenv), a.kv_len));
    const int nch = (len + UT - 1) / UT;  // chunks with at least one key
    const int last = max(len - 1, 0);

    // requests of a chunk: token indices clamped to the last valid key (never a predicated load); its score is masked below
    auto issue_k = [&](int c, int u0) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < UT; ++u) {
            if (u < u0) continue;
            const unsigned tc = (unsigned)min(c * UT + u, last);
            kreg[u] = __builtin_nontemporal_load((gu32x4_p)(kbu + (tc * krs + klane)));
        }
    };
    auto issue_v = [&](int c) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < UT; ++u) {
            const unsigned tc = (unsigned)min(c * UT + u, last);
            vreg[u] = __builtin_nontemporal_load((gu32x4_p)(vbu + (tc * vrs + vlane)));
        }
    };
    // one chunk out of the registers.  LAST = false: a full chunk that is not the sequence's last one (all UT keys valid; the
    // next chunk's K / V are requested as soon as this one's are out of their registers).  LAST = true: the sequence's final
    // chunk, masked by the length, nothing requested behind it.
    auto chunk = [&](int c, auto LAST) __attribute__((always_inline)) {
        constexpr bool is_last = decltype(LAST)::value;
        float sv[UT];
#pragma unroll
        for (int u = 0; u < UT; ++u) {
            float d = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) d = TR::dot2(qp[e], kreg[u][e], d);
            d = group_sum<LPK>(d) * sc;
            sv[u] = (!is_last || c * UT + u < len) ? d : -INFINITY;  // wave-uniform condition
        }
        if constexpr (!is_last) {
            __builtin_amdgcn_sched_barrier(0);
            issue_k(c + TS, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        float cmax = sv[0];
#pragma unroll
        for (int u = 1; u < UT; ++u) cmax = fmaxf(cmax, sv[u]);
        const float mnew = fmaxf(m, cmax);  // finite: every chunk that is processed starts with a valid key
        const float alpha = fast_exp2(m - mnew);
        float ps = 0.f;
#pragma unroll
        for (int u = 0; u < UT; ++u) {
            sv[u] = fast_exp2(sv[u] - mnew);
            ps += sv[u];
        }
        l = l * alpha + ps;
        m = mnew;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] *= alpha;
#pragma unroll
        for (int u = 0; u < UT; ++u) {
            float vf[8];
            widen8<T>(vreg[u], vf);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = __builtin_fmaf(sv[u], vf[j], acc[j]);
        }
        if constexpr (!is_last) {
            __builtin_amdgcn_sched_barrier(0);
            issue_v(c + TS);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    if (nch > ts_id || HS > 0) {  // (a wave without a chunk requests nothing: an empty sequence's cache may have no rows at all)
        issue_k(ts_id, HS);  // this wave's first chunk in the steady state's order: K, then V
        issue_v(ts_id);
        __builtin_amdgcn_sched_barrier(0);
        int c = ts_id;
        for (; c + TS < nch; c += TS) chunk(c, std::integral_constant<bool, false>{});
        if (nch > ts_id) chunk(c, std::integral_constant<bool, true>{});  // this wave's last chunk: masked (it may be the sequence's last)
    }
    if constexpr (TS > 1) {
        __shared__ float xch[4][10][64];  // [wave of the workgroup][m, l, acc[8]][lane]
        if (ts_id > 0) {
            xch[wave][0][lane] = m;
            xch[wave][1][lane] = l;
#pragma unroll
            for (int j = 0; j < 8; ++j) xch[wave][2 + j][lane] = acc[j];
        }
        __syncthreads();
        if (ts_id > 0) return;
#pragma unroll
        for (int t = 1; t < TS; ++t) {
            float a2[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) a2[j] = xch[wave + t][2 + j][lane];
            merge_state(m, l, acc, xch[wave + t][0][lane], xch[wave + t][1][lane], a2);
        }
    }
    finish_row<T, D, 2, NPRE>(a, ridx, sub, m, l, acc, npre, pp);
}

#ifdef HYD_ABLATION_BUILD
// The first form of the token-row kernel (all of a chunk's loads, then all of its arithmetic), kept in development builds for the
// A/B tables of profiles/r06_suffix_rows_*.txt (HYD_ROWS_PIPE=0).  ROT: the full chunks of a sequence are walked from a
// per-sequence starting chunk (softmax does not care about the order), so that workgroups that start together do not touch the
// same token offsets of caches that sit a pathological distance apart.  Measured (profiles/r06_suffix_stride_sweep.txt): it
// rescues the bad strides (129 rows between sequences: 193 vs 212 us at S = 64, the one-unit-per-wave kernel 243) and costs
// 2-10 % on the good ones (128 / 256 / 512 / 2048 rows, S = 128: 345 vs 332, 350 vs 318), which are the ones cache allocations
// have (capacities are multiples of 16 rows): not shipped.
template <typename T, int D, int UT, int NPRE, int ROT>
__global__ __launch_bounds__(256, 4) void suffix_attn_rows1_kernel(const SuffixArgs a) {
    using TR = Traits<T>;
    constexpr int LPK = D / 8, HPI = 64 / LPK;  // lanes per head row, heads per wave instruction
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int sub = lane % LPK, hg = lane / LPK;
    // waves per sequence inside a workgroup: 4 (then blockIdx.y walks further head slices), 2 or 1
    const int wl = a.rows_wps_log2;
    const int b = (int)(blockIdx.x << (2 - wl)) + (wave >> wl);
    const int h0 = (int)((blockIdx.y << wl) + (wave & ((1 << wl) - 1))) * HPI;  // first head of this wave
    if (b >= a.B || h0 >= a.Hkv) return;

    int len = a.kv_len;
    if (a.sl32) len = a.sl32[b];
    else if (a.sl64) len = (int)a.sl64[b];
    len = max(0, min(len, a.kv_len));

    const int64_t ridx = (int64_t)b * a.Hq + h0 + hg;  // nq == 1, g == 1: [B, 1, Hq]
    const u32x4 qp = *reinterpret_cast<const u32x4*>(static_cast<const uint16_t*>(a.q) + ridx * D + sub * 8);
    PrePartials<NPRE> pp;
    const int npre = min(n_prefetched(a), NPRE);
    prefetch_partials(a, npre, ridx, sub, D, pp);

    // wave-uniform base (scalar registers) + per-lane 32-bit byte offset (head, dims) -> SADDR-form loads
    const gchar_p kbu = uniform_ptr(reinterpret_cast<const char*>(static_cast<const uint16_t*>(a.k) + (int64_t)b * a.k_bs + (int64_t)h0 * a.k_hs));
    const gchar_p vbu = uniform_ptr(reinterpret_cast<const char*>(static_cast<const uint16_t*>(a.v) + (int64_t)b * a.v_bs + (int64_t)h0 * a.v_hs));
    const unsigned klane = (unsigned)(hg * a.k_hs * 2 + sub * 16), vlane = (unsigned)(hg * a.v_hs * 2 + sub * 16);
    const unsigned krs = (unsigned)(a.k_ts * 2), vrs = (unsigned)(a.v_ts * 2);  // token stride in bytes

    float m = -INFINITY, l = 0.f, acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    const float sc = a.scale_log2e;

    // one chunk = UT tokens: all loads first, then the scores, one online-softmax update, then P.V
    auto chunk = [&](int t0, auto MASKED) __attribute__((always_inline)) {
        u32x4 kreg[UT], vreg[UT];
#pragma unroll
        for (int u = 0; u < UT; ++u) {
            // never predicate a load: clamp to the last valid key, its score is forced to -inf below
            const unsigned tc = (unsigned)(decltype(MASKED)::value ? min(t0 + u, len - 1) : t0 + u);
            kreg[u] = __builtin_nontemporal_load((gu32x4_p)(kbu + (tc * krs + klane)));
            vreg[u] = __builtin_nontemporal_load((gu32x4_p)(vbu + (tc * vrs + vlane)));
        }
        float sv[UT];
#pragma unroll
        for (int u = 0; u < UT; ++u) {
            float d = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) d = TR::dot2(qp[e], kreg[u][e], d);
            d = group_sum<LPK>(d) * sc;
            sv[u] = (!decltype(MASKED)::value || t0 + u < len) ? d : -INFINITY;  // wave-uniform condition
        }
        float cmax = sv[0];
#pragma unroll
        for (int u = 1; u < UT; ++u) cmax = fmaxf(cmax, sv[u]);
        const float mnew = fmaxf(m, cmax);  // finite: t0 < len
        const float alpha = fast_exp2(m - mnew);
        float ps = 0.f;
#pragma unroll
        for (int u = 0; u < UT; ++u) {
            sv[u] = fast_exp2(sv[u] - mnew);
            ps += sv[u];
        }
        l = l * alpha + ps;
        m = mnew;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] *= alpha;
#pragma unroll
        for (int u = 0; u < UT; ++u) {
            float vf[8];
            widen8<T>(vreg[u], vf);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = __builtin_fmaf(sv[u], vf[j], acc[j]);
        }
    };
    using Full = std::integral_constant<bool, false>;
    using Masked = std::integral_constant<bool, true>;
    const int nfull = len / UT;
    int c = 0;
    if constexpr (ROT != 0) {
        // starting chunk = a 16-bit hash of the sequence index scaled to [0, nfull)
        c = (int)(((((unsigned)b * 0x9E3779B1u) >> 16) * (unsigned)nfull) >> 16);
    }
    for (int i = 0; i < nfull; ++i) {
        chunk(c * UT, Full{});
        c = c + 1 == nfull ? 0 : c + 1;
    }
    if (nfull * UT < len) chunk(nfull * UT, Masked{});
    finish_row<T, D, 2, NPRE>(a, ridx, sub, m, l, acc, npre, pp);
}
#endif  // HYD_ABLATION_BUILD

// shapes only (capture-safe): one query row per unit, whole lane groups, 32-bit offsets inside a sequence's cache, and
// enough waves to fill the chip (below that the keys of a unit are spread over four waves by launch_suffix_r)
template <int D>
static bool suffix_rows_eligible(const SuffixArgs& a) {
    constexpr int HPI = 64 / (D / 8);
    const int64_t span = (int64_t)a.kv_len * (a.k_ts > a.v_ts ? a.k_ts : a.v_ts) * 2 +
                         (int64_t)a.Hkv * (a.k_hs > a.v_hs ? a.k_hs : a.v_hs) * 2;
    return a.rows == 1 && a.nq == 1 && a.g == 1 && a.Hkv % HPI == 0 && span < ((int64_t)1 << 31) && a.n_pre <= 2;
}

template <typename T, int D>
static int launch_suffix_rows(const SuffixArgs& a0, hipStream_t s, int ut, int rot, int pipe) {
    constexpr int HPI = 64 / (D / 8);
    SuffixArgs a = a0;
    const int wps = a.Hkv / HPI;  // waves per sequence
    a.rows_wps_log2 = wps >= 3 ? 2 : wps == 2 ? 1 : 0;
    const int wl = a.rows_wps_log2;
    // Token split (shapes only).  When one wave covers all heads of a token (Hkv = the 64 / (D / 8) heads of a wave instruction: a
    // 1 KB token row, a TP = 8 shard of C2), the 4 waves of a workgroup used to walk 4 different sequences, 8 KB of each at a time;
    // sharing ONE sequence between 2 (4) of them -- 16 (32) KB of the same contiguous cache requested together -- streams 6-13 %
    // faster from S = 16 on at every batch size (profiles/r06_suffix_rows_token_split_ab.txt: B = 1024, S = 64 28.5 -> 25.8 us,
    // S = 128 51.4 -> 44.9; B = 8192, S = 64 183 -> 167); 2 is the better split up to 2048 sequences, 4 above (and the cheaper one at
    // S = 8: + 0.5 us).  With two or more waves per sequence already (8 or more kv heads at D = 128) it changes nothing: not used.
    int ts = 1;
    if (wl == 0 && a.n_pre < 2 && a.kv_len >= 32) ts = a.B <= 2048 ? 2 : 4;
#ifdef HYD_ABLATION_BUILD
    if (const char* e = getenv("HYD_ROWS_TS")) { ts = atoi(e); if (wl + (ts == 4 ? 2 : ts == 2 ? 1 : 0) > 2 || a.n_pre >= 2) ts = 1; }
#endif
    const int tl = ts == 4 ? 2 : ts == 2 ? 1 : 0;
    const dim3 grid((unsigned)((a.B + (4 >> (wl + tl)) - 1) >> (2 - wl - tl)), (unsigned)((wps + (1 << wl) - 1) >> wl), 1);
#define HYD_ROWS_LAUNCH(KERNEL) \
    do { hipLaunchKernelGGL((KERNEL), grid, dim3(256), 0, s, a); return (int)hipGetLastError(); } while (0)
#ifdef HYD_ABLATION_BUILD
    if constexpr (D == 128) {
        if (a.n_pre < 2) {
            if (pipe == 0 && ut == 4) HYD_ROWS_LAUNCH((suffix_attn_rows1_kernel<T, D, 4, 1, 0>));
            if (pipe == 0 && rot) HYD_ROWS_LAUNCH((suffix_attn_rows1_kernel<T, D, 8, 1, 1>));
            if (pipe == 0) HYD_ROWS_LAUNCH((suffix_attn_rows1_kernel<T, D, 8, 1, 0>));
            if (pipe == 1 && a.kv_len >= 4) HYD_ROWS_LAUNCH((suffix_attn_rows_kernel<T, D, 8, 1, 4>));
            if (pipe == 2 && a.kv_len >= 2) HYD_ROWS_LAUNCH((suffix_attn_rows_kernel<T, D, 8, 1, 2>));
        }
    }
#endif
    (void)ut;
    (void)rot;
    (void)pipe;
    if (a.n_pre == 2) HYD_ROWS_LAUNCH((suffix_attn_rows_kernel<T, D, 8, 2>));
    if (ts == 4) HYD_ROWS_LAUNCH((suffix_attn_rows_kernel<T, D, 8, 1, 0, 4>));
    if (ts == 2) HYD_ROWS_LAUNCH((suffix_attn_rows_kernel<T, D, 8, 1, 0, 2>));
    HYD_ROWS_LAUNCH((suffix_attn_rows_kernel<T, D, 8, 1>));
#undef HYD_ROWS_LAUNCH
}

// NPRE: 16-bit partials fetched under the K/V stream (suffix_common.h); 2 is instantiated for the decode shape only
// (R = 1, one wave per unit) and launched when the call has two such partials (a two-level hierarchy).
template <typename T, int D, int R, int WPU, int NPRE = 1, int U_ = 4, int OCC = (R == 1 ? 6 : 1), int PIPE = 0>
__global__ __launch_bounds__(256, OCC) void suffix_attn_kernel(const SuffixArgs a) {
    using TR = Traits<T>;
    constexpr int LPK = D / 8;    // lanes per key row
    constexpr int KPI = 64 / LPK; // keys per wave instruction
    constexpr int U = U_;  // key iterations in flight per wave (x2 tensors x 1 KiB); occupancy supplies the rest
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
    const int b = a.order ? a.order[blockIdx.x] : (int)blockIdx.x;  // dispatch slot -> sequence (hyd_suffix_params.seq_order)
    const int hk = blockIdx.y * (4 / WPU) + wave / WPU;
    if (hk >= a.Hkv) return;  // uniform per unit (all WPU waves of a unit leave together)
    const int row0 = blockIdx.z * R;

    int len = a.kv_len;
    if (a.sl32) len = a.sl32[b];
    else if (a.sl64) len = (int)a.sl64[b];
    len = max(0, min(len, a.kv_len));
    if constexpr (R == 1 && WPU == 1 && D == 128) {
        // short sequence and a packable shape (a.packed, shapes only): the first of every 4 workgroups of this
        // sequence serves the 16 (D = 128) kv heads of all four with the lane-group layout, the other three leave
        if (a.packed && len <= kPackedMaxLen) {
            if ((blockIdx.y & 3) == 0) suffix_packed_body<T, D, NPRE>(a, b, blockIdx.y >> 2, len);
            return;
        }
    }

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
    // ---- prefetch the first 16-bit partials (the usual single prefix level; a second level) for the rows this lane
    // group will finish in the epilogue, so their HBM latency overlaps the K/V stream instead of following it -----
    constexpr int RPG = (R + KPI - 1) / KPI;  // epilogue rows per lane group
    PrePartials<NPRE> pp[RPG];
    const int npre = min(n_prefetched(a), NPRE);
#pragma unroll
    for (int j = 0; j < RPG; ++j) {
        const int r = ks + j * KPI;
        const int row = row0 + r;
        const bool live = r < R && row < a.rows;
        const int rowc = live ? row : 0;
        const int iq = a.nq == 1 ? 0 : rowc / a.g, gq = a.nq == 1 ? rowc : rowc % a.g;
        const int64_t ridx = ((int64_t)b * a.nq + iq) * a.Hq + hk * a.g + gq;
        prefetch_partials(a, live ? npre : 0, ridx, sub, D, pp[j]);
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
    if constexpr (PIPE != 0) {
        // Long caches (more than 1024 rows: the token-row kernel does not take them): the full chunks with rotated requests, as in the
        // token-row kernel -- K of chunk c + 1 goes out as soon as the scores of chunk c have left the K registers, V of chunk c + 1
        // behind P.V of chunk c -- the same registers, never an empty queue.  2176-row caches, C2 heads, one process, alternating
        // (profiles/r06_unit_kernel_rotated_ab.txt): S = 64 188.7 -> 183.6 us, S = 512 1319 -> 1298, S = 2176 5401 -> 5337;
        // on 128- and 1024-row caches it loses 1-4 %, so only caches beyond 1024 rows take it.
        const int nfull = niter / U;
        if (nfull > 0) {
            u32x4 kreg[U], vreg[U];
            auto key_of = [&](int c, int u) __attribute__((always_inline)) { return ((c * U + u) * WPU + wv) * KPI + ks; };
            auto issue_k = [&](int c) __attribute__((always_inline)) {
#pragma unroll
                for (int u = 0; u < U; ++u)
                    kreg[u] = __builtin_nontemporal_load((gu32x4_p)(kbu + ((unsigned)min(key_of(c, u), len - 1) * krs + sub * 16)));
            };
            auto issue_v = [&](int c) __attribute__((always_inline)) {
#pragma unroll
                for (int u = 0; u < U; ++u)
                    vreg[u] = __builtin_nontemporal_load((gu32x4_p)(vbu + ((unsigned)min(key_of(c, u), len - 1) * vrs + sub * 16)));
            };
            auto body = [&](int c, auto MORE) __attribute__((always_inline)) {
                constexpr bool more = decltype(MORE)::value;
                float s[R][U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const bool valid = key_of(c, u) < len;
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        float d = 0.f;
#pragma unroll
                        for (int i = 0; i < 4; ++i) d = TR::dot2(qp[r][i], kreg[u][i], d);
                        d = group_sum<LPK>(d);
                        s[r][u] = valid ? d * sc : -INFINITY;
                    }
                }
                if constexpr (more) {
                    __builtin_amdgcn_sched_barrier(0);
                    issue_k(c + 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    float cmax = s[r][0];
#pragma unroll
                    for (int u = 1; u < U; ++u) cmax = fmaxf(cmax, s[r][u]);
                    const float mnew = fmaxf(m[r], cmax);
                    const float ms = (mnew == -INFINITY) ? 0.f : mnew;
                    const float alpha = fast_exp2(m[r] - ms);
                    float ps = 0.f;
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        s[r][u] = fast_exp2(s[r][u] - ms);
                        ps += s[r][u];
                    }
                    l[r] = l[r] * alpha + ps;
                    m[r] = mnew;
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[r][j] *= alpha;
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    float vf[8];
                    widen8<T>(vreg[u], vf);
#pragma unroll
                    for (int r = 0; r < R; ++r)
#pragma unroll
                        for (int j = 0; j < 8; ++j) acc[r][j] = __builtin_fmaf(s[r][u], vf[j], acc[r][j]);
                }
                if constexpr (more) {
                    __builtin_amdgcn_sched_barrier(0);
                    issue_v(c + 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            issue_k(0);
            issue_v(0);
            __builtin_amdgcn_sched_barrier(0);
            int c = 0;
            for (; c + 1 < nfull; ++c) body(c, std::integral_constant<bool, true>{});
            body(c, std::integral_constant<bool, false>{});
            it = nfull * U;
        }
    } else {
        for (; it + U <= niter; it += U) chunk(std::integral_constant<int, U>{}, it);
    }
    for (; it < niter; ++it) chunk(std::integral_constant<int, 1>{}, it);

    // ---- merge the KPI lane groups of the wave (butterfly: every lane ends with the total) -----
#pragma unroll
    for (int r = 0; r < R; ++r) merge_lane_groups<LPK>(m[r], l[r], acc[r]);

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
        finish_row<T, D, 4, NPRE>(a, ridx, sub, m[r], l[r], acc[r], npre, pp[r / KPI]);
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
#ifdef HYD_ABLATION_BUILD
        // timing probe only (results race with the prefix pass): launch without the in-queue barrier
        if (const char* e = getenv("HYD_ANYORDER"); e && atoi(e)) {
            hipExtLaunchKernelGGL((suffix_attn_kernel<T, D, R, 1>), grid, dim3(256), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, a);
            return (int)hipGetLastError();
        }
        // occupancy A/B (VERDICT r5 next #1c): key iterations in flight x resident waves per SIMD
        if constexpr (R == 1 && D == 128) {
            if (const char* e = getenv("HYD_SUFFIX_OCC"); e && a.n_pre < 2) {
                switch (atoi(e)) {
                    case 48: hipLaunchKernelGGL((suffix_attn_kernel<T, D, R, 1, 1, 4, 8>), grid, dim3(256), 0, s, a); return (int)hipGetLastError();
                    case 38: hipLaunchKernelGGL((suffix_attn_kernel<T, D, R, 1, 1, 3, 8>), grid, dim3(256), 0, s, a); return (int)hipGetLastError();
                    case 28: hipLaunchKernelGGL((suffix_attn_kernel<T, D, R, 1, 1, 2, 8>), grid, dim3(256), 0, s, a); return (int)hipGetLastError();
                    case 65: hipLaunchKernelGGL((suffix_attn_kernel<T, D, R, 1, 1, 6, 5>), grid, dim3(256), 0, s, a); return (int)hipGetLastError();
                    case 84: hipLaunchKernelGGL((suffix_attn_kernel<T, D, R, 1, 1, 8, 4>), grid, dim3(256), 0, s, a); return (int)hipGetLastError();
                    default: break;
                }
            }
        }
#endif
        if constexpr (R == 1) {
            if (a.n_pre == 2) {
                hipLaunchKernelGGL((suffix_attn_kernel<T, D, R, 1, 2>), grid, dim3(256), 0, s, a);
                return (int)hipGetLastError();
            }
        }
        if constexpr (R == 1) {
            if (a.kv_len > 1024) {  // long caches: rotated requests (see PIPE in the kernel)
                hipLaunchKernelGGL((suffix_attn_kernel<T, D, R, 1, 1, 4, 6, 1>), grid, dim3(256), 0, s, a);
                return (int)hipGetLastError();
            }
        }
        hipLaunchKernelGGL((suffix_attn_kernel<T, D, R, 1>), grid, dim3(256), 0, s, a);
    }
    return (int)hipGetLastError();
}

// packed path: one query row per unit (nq == 1, Hq == Hkv), 32-bit offsets inside a sequence's cache
static bool suffix_packed_eligible(const SuffixArgs& a, int D) {
    const int hpw = 64 / (D / 8);
    const int64_t span = (int64_t)a.kv_len * (a.k_ts > a.v_ts ? a.k_ts : a.v_ts) * 2 +
                         (int64_t)a.Hkv * (a.k_hs > a.v_hs ? a.k_hs : a.v_hs) * 2;
    return a.rows == 1 && a.nq == 1 && a.g == 1 && a.Hkv >= hpw && D == 128 && span < ((int64_t)1 << 31);
}

template <typename T, int D>
static int launch_suffix_t(const SuffixArgs& a0, hipStream_t s) {
    SuffixArgs a = a0;
    a.packed = suffix_packed_eligible(a, D) ? 1 : 0;
    {
        // token-row kernel: measured faster wherever the one-unit-per-wave kernel would run with one wave per unit
        // (launch_suffix_r's few_units rule spreads a unit over four waves below 2048 units)
        // ... and whose caches hold at most 1024 token rows: on longer rows a wave start costs little, and the one-unit-per-wave kernel
        // streams them as fast or faster (2176-row caches, profiles/r06_suffix_rows_capacity.txt)
        bool rows = suffix_rows_eligible<D>(a) && !((int64_t)a.units < 2 * 256 * 4 && a.kv_len >= 64) && a.kv_len <= 1024;
        int ut = 8, rot = 0, pipe = 3;  // pipe (development builds): 0 = the first form, 1 / 2 = blind K requests, 3 = the product form
#ifdef HYD_ABLATION_BUILD
        if (const char* e = getenv("HYD_ROWS_PIPE")) pipe = atoi(e);
        if (const char* e = getenv("HYD_SUFFIX_ROWS")) rows = atoi(e) != 0 && suffix_rows_eligible<D>(a);
        if (const char* e = getenv("HYD_ROWS_UT")) ut = atoi(e);
        if (const char* e = getenv("HYD_ROWS_ROT")) rot = atoi(e);
#endif
        if (rows) return launch_suffix_rows<T, D>(a, s, ut, rot, pipe);
    }
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
        if (D == 256) return launch_suffix_t<F16, 256>(a, s);
    } else {
        if (D == 128) return launch_suffix_t<BF16, 128>(a, s);
        if (D == 64) return launch_suffix_t<BF16, 64>(a, s);
        if (D == 256) return launch_suffix_t<BF16, 256>(a, s);
    }
    return (int)hipErrorInvalidValue;
}

}  // namespace hyd
