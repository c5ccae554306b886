// Suffix pass for grouped-query shapes (nq * g >= 4 query rows per (sequence, kv head)): the same operator as
// suffix_attn.hip -- attention of the unit's rows over the first seq_len[b] keys of the sequence's own cache
// plus the fused LSE combine with the prefix partials -- but with the two contractions on the matrix cores.
//
// Replaces _fwd_kernel_splitK + _splitK_reduce (/root/reference/hydragen/flash.py:76-281,
// hydragen/xformers_stuff.py:189-428) and combine_lse (hydragen/attention.py:21-174) for GQA configurations
// such as the reference microbenchmark's default Hq = 8, Hkv = 1 (scripts/microbenchmark.py:136-138).
//
// Why a second kernel: with R rows per unit the wave-level dot-product kernel issues ~R * 12 VALU instructions
// per 4 keys and stops being HBM-bound at R = 8 (measured 2.1 TB/s at B = 2048, Hq/Hkv = 8/1, S = 256).  Here a
// wave owns one unit and walks its keys 32 at a time:
//   K, V: LDS-DMA (buffer_load ... lds: whole rows, zero fill past seq_len) into one swizzled 32-key landing tile each;
//        once a step has landed its tiles are emptied into registers -- K as MFMA A operands (ds_read_b128), V^T with
//        ds_read_b64_tr_b16 -- and handed straight to the next step's DMA
//   S^T = K . Q^T      2 * D/32 x v_mfma_f32_16x16x32 (Q rows padded to 16, kept in registers)
//   online softmax in base 2, one query row per lane & 15, 8 scores per lane, two cross-lane exchanges
//   O^T += V^T . P^T   D/16 x MFMA, P^T converted in registers (the contraction index is permuted to the
//        accumulator layout, keys {4j..4j+3, 16+4j..}, so no data moves)
// Producer and consumer of a wave's tiles are the same wave: no barriers.
//
// Software pipeline: collect(j) -> issue(j + 1) -> compute(j): one 16 KiB step in flight per wave while the previous one is
// computed out of registers, 16 KiB of LDS per wave, 8 waves per CU.  Round 4's form double-buffered in LDS (V) and in
// asm-owned AGPRs (K, loaded straight into operand layout: 16 rows x 64 B per instruction, every quarter wave 16 different
// cache lines); a timing experiment of this round showed that a coalesced route for K alone is worth 10 % on 8-kv-head shapes
// (whole-job C5, S = 128: 222 -> 198 us), which is what the landing tiles are for.
#include <cstdlib>
#include <type_traits>

#include "suffix_gqa_common.h"

namespace hyd {

// WPU: waves per unit.  1: a one-wave workgroup per unit (producer and consumer of the LDS tile are the same wave, no
// barriers).  4: the unit's 32-key steps are dealt round-robin to the 4 waves of a 256-thread workgroup and their
// (m, l, O) merged through LDS at the end -- for shapes with too few units to fill the chip with one wave each (C3: 1024
// units on 256 CUs, C5: 2048), where nothing but more waves hides the per-step load latency.
// NT: K/V loads carry the non-temporal hint (suffix_gqa_common.h): the unique phase, where every key is read once.
// HPW: kv heads per workgroup (WPU = 1 only).  The HPW one-wave units of a workgroup are the kv heads hk0 .. hk0 + HPW - 1 of ONE
// sequence: they start together, have the same length and walk the same tokens, so the 2 D-byte pieces they read of every
// token's [Hkv, D] row are requested together (the one-wave workgroups of a sequence's heads are B workgroups apart in
// dispatch order).  No wave talks to another.  Chosen from shapes in launch_gqa_t (measurements there).
template <typename T, int D, int WPU, bool NT, int HPW = 1>
__global__ __launch_bounds__(64 * WPU * HPW) __attribute__((amdgpu_waves_per_eu(D == 256 ? 1 : 2, D == 256 ? 1 : 2))) void suffix_attn_gqa_kernel(const SuffixArgs a) {
    static_assert(WPU == 1 || HPW == 1, "several heads per workgroup: one-wave units only");
    constexpr int NWV = WPU * HPW;  // waves per workgroup
    // every field a wave needs (and partials[0]) sits in the first 256 bytes of SuffixArgs: one scalar-cache miss at the start of
    // a unit's dependent chain (C5 slice 7.17 -> 7.00 us at S = 16, 25.6 -> 24.9 at S = 128 in an A/B within one run; the dot-product
    // kernel, whose 32768 waves all pay the four extra loads, lost 2-7 % at S <= 4 with it and does not have it)
    warm_kernargs_256();
    using TR = Traits<T>;
    constexpr int RB = D * 2;        // bytes per K/V row
    constexpr int NCH = D / 32;      // 32-dim chunks of the QK^T contraction
    constexpr int NDB = D / 16;      // 16-wide d blocks of O^T
    constexpr int RPI = 1024 / RB;   // rows per DMA instruction
    constexpr int NVD = 32 / RPI;    // DMA instructions per 32-key tile
    constexpr int TILE = 32 * RB;    // bytes of one 32-key tile (= 16 rows * D floats: reused by the merge)
    // NSET tile sets per wave = key steps in flight.  A 32-key step is 2 x 4 KB at D = 64 -- half of what a D = 128 wave keeps in
    // flight, and the kernel is bound by exactly that (profiles/r06_head_dim_rates.txt: 4.9-5.7 TB/s at D = 64 against 5.6-6.1 at
    // 128) -- so D = 64 keeps TWO steps in flight: step j + 2 is requested into the set step j has just been read out of.
    constexpr int NSET = D == 64 ? 2 : 1;
    constexpr unsigned SETB = 2 * TILE;  // bytes of one set (K tile, V tile)
    // dynamic LDS: per wave one K and one V landing tile, then the merge's [NWV][4][16] floats
    extern __shared__ __attribute__((aligned(1024))) char gqa_smem[];
    char(*tiles)[NSET][2][TILE] = reinterpret_cast<char(*)[NSET][2][TILE]>(gqa_smem);  // [wave][set][0 = K, 1 = V]
    float(*mlx)[4][16] = reinterpret_cast<float(*)[4][16]>(gqa_smem + NWV * NSET * 2 * TILE);
    const int wv = NWV == 1 ? 0 : __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // wave of the workgroup
    const int wave = WPU == 1 ? 0 : wv;                                                      // wave of the unit
    char* ktile = tiles[wv][0][0];
    char* vtile = tiles[wv][0][1];

    const int lane = threadIdx.x & 63;
    const int l15 = lane & 15, g4 = lane >> 4;
    // workgroup -> (sequence, kv-head group): blockIdx.x runs over B * (Hkv / HPW) pairs, sequences fastest
    const int hkg = (int)blockIdx.x / a.B;
    const int bslot = (int)blockIdx.x - hkg * a.B;
    const int b = a.order ? a.order[bslot] : bslot;  // dispatch slot -> sequence (hyd_suffix_params.seq_order)
    const int hk = hkg * HPW + (HPW == 1 ? 0 : wv), row0 = blockIdx.y * 16;

    // the sequence's length is requested first (a scalar load): q and the first partials are requested under its round trip, the first
    // key step goes out when it arrives
    int len_raw = a.kv_len;
    if (a.sl32) len_raw = a.sl32[b];
    else if (a.sl64) len_raw = (int)a.sl64[b];

    // ---- this lane's query row (B operand of S^T = K Q^T: column = row l15, 8 dims of every 32-dim chunk) ----
    const int row = row0 + l15;
    const bool rvalid = row < a.rows;
    const int iq = a.nq == 1 ? 0 : (rvalid ? row / a.g : 0), gq = a.nq == 1 ? (rvalid ? row : 0) : (rvalid ? row % a.g : 0);
    const int64_t ridx = ((int64_t)b * a.nq + iq) * a.Hq + hk * a.g + gq;  // [B, nq, Hq]
    u32x4 qf[NCH];
    {
        const uint16_t* qr = static_cast<const uint16_t*>(a.q) + ridx * D + 8 * g4;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const u32x4 z = {0u, 0u, 0u, 0u};
            qf[c] = rvalid ? *reinterpret_cast<const u32x4*>(qr + 32 * c) : z;
        }
    }

    // ---- prefix partials (attention.py:21-43) are dealt to the unit's waves: wave w folds partials w, w + WPU, ... into
    // its own (m, l, O) state -- a normalised partial (O_p, lse_p) IS a state (m = lse_p * log2 e, l = 1, O = O_p) -- and the
    // merge of the waves then combines everything; no partial is left for the final epilogue.  The wave's first TWO partials (16-bit,
    // fp32, or a split level's fp32 slices) are requested here, in front of the K/V stream, and folded when the first key step
    // has landed; the others are fetched behind the loop, two per round trip.  (hipcc counts only its own loads.  A plain load
    // that is still pending inside the loop makes it insert counted waits that also wait for the next step's DMAs, which it
    // does not know about -- one plain request before the stream, settled at the first wait, is what stays exact.  Round 5
    // tried one partial per key step through an asm-owned register buffer: 1 us at the paper default, nothing at C3 / C5.)
    // Not when the suffix pass's own LSE is asked for (a.lse: the unfused form), which needs the keys-only state at the end.
    const int np = a.n_partials;
    constexpr int NPRE = D == 256 ? 1 : 2;  // partials of this wave requested in front of the stream (a split level's two slices at C5; 2 of 4 at C3)
    const int npre = a.lse != nullptr ? 0 : min(NPRE, (np - wave + WPU - 1) / WPU);  // (np <= wave: 0)
    bool pre_folded = npre <= 0;
    bool pre_f32[NPRE];
    u32x4 pbuf[NPRE][NDB];  // a partial's row piece of this lane: dims [16 db + 4 g4, +4) as fp32, or as 16-bit in the low half
    float plse[NPRE];
#pragma unroll
    for (int k = 0; k < NPRE; ++k) {
        pre_f32[k] = false;
        plse[k] = -INFINITY;
#pragma unroll
        for (int db = 0; db < NDB; ++db) pbuf[k][db] = u32x4{0u, 0u, 0u, 0u};
        if (k < npre) {
            const PartialDev& pd = a.partials[wave + k * WPU];
            pre_f32[k] = pd.is_f32 != 0;
            plse[k] = pd.lse[ridx];
            if (pre_f32[k]) {
                const float* po = static_cast<const float*>(pd.out) + ridx * D + 4 * g4;
#pragma unroll
                for (int db = 0; db < NDB; ++db) pbuf[k][db] = *reinterpret_cast<const u32x4*>(po + 16 * db);
            } else {
                const uint16_t* po = static_cast<const uint16_t*>(pd.out) + ridx * D + 4 * g4;
#pragma unroll
                for (int db = 0; db < NDB; ++db) {  // (upper half undefined: no move, hence no wait, between the load and its register)
                    const u32x2 u = *reinterpret_cast<const u32x2*>(po + 16 * db);
                    pbuf[k][db] = __builtin_shufflevector(u, u, 0, 1, -1, -1);
                }
            }
        }
    }

    // ---- K / V windows of this unit: rows [0, len) are in range, everything else reads as zero ------------------
    // The windows span the cache's kv_len rows, NOT the sequence's length: the first step is requested before the length has
    // arrived (below), so rows in [len, kv_len) may land in the tiles -- whatever they hold (the tests poison them with NaN): their
    // scores are replaced by -inf (a select, not arithmetic) and their V rows are zeroed in registers (sanitize, masked steps only).
    const unsigned k_ts2 = (unsigned)(a.k_ts * 2), v_ts2 = (unsigned)(a.v_ts * 2);
    const u32x4 krs = make_rsrc_g(static_cast<const uint16_t*>(a.k) + (int64_t)b * a.k_bs + (int64_t)hk * a.k_hs, (unsigned)a.kv_len * k_ts2);
    const u32x4 vrs = make_rsrc_g(static_cast<const uint16_t*>(a.v) + (int64_t)b * a.v_bs + (int64_t)hk * a.v_hs, (unsigned)a.kv_len * v_ts2);
    // Tiny problems (SuffixArgs::pk): the group's shared prefix is walked first, through its own resources, then the
    // unit's own keys -- the (m, l, O) state simply carries over, and the whole operator is this one launch.
    const bool has_pre = a.pk != nullptr;
    const int gi = has_pre ? b / a.p_per : 0;
    const u32x4 krs_p = make_rsrc_g(static_cast<const uint16_t*>(has_pre ? a.pk : a.k) + (int64_t)gi * a.pk_gs + (int64_t)hk * a.pk_hs,
                                    has_pre ? (unsigned)a.p_len * k_ts2 : 0u);
    const u32x4 vrs_p = make_rsrc_g(static_cast<const uint16_t*>(has_pre ? a.pv : a.v) + (int64_t)gi * a.pv_gs + (int64_t)hk * a.pv_hs,
                                    has_pre ? (unsigned)a.p_len * v_ts2 : 0u);
    u32x4 krs_c = krs, vrs_c = vrs;  // the segment being walked
    int seg_len = 0, seg_cap = 0;  // keys of the segment being walked; rows its window addresses
    // K and V DMA: instruction i covers tile rows [i * RPI, +RPI) -- whole rows, 1 KiB contiguous when the heads of a token are
    // (Hkv = 1), RPI row pieces of 2 D bytes otherwise; the LDS image of an instruction is lane-linear, so the XOR swizzles of
    // the two tiles are applied to the per-lane SOURCE chunk (involutions inside a row).
    //   K tile: 16-byte chunk j of row r sits at position j ^ sw_k(r) -- the A-operand reads below (16 rows, one chunk
    //           column per quarter wave) then touch every bank once;  sw_k(r) = r & 15 (D >= 128), (r >> 1) & 7 (D = 64)
    //   V tile: 64-byte groups swizzled for the transposing reads (as before)
    const int drow = (lane * 16) / RB, dcp = ((lane * 16) % RB) >> 4;
    unsigned kvoff[NVD], vvoff[NVD];
#pragma unroll
    for (int i = 0; i < NVD; ++i) {
        const int r_ = i * RPI + drow;
        const int swk = D >= 128 ? (r_ & 15) : ((r_ >> 1) & 7);
        kvoff[i] = (unsigned)r_ * k_ts2 + (unsigned)(dcp ^ swk) * 16u;
        const int sw = D >= 128 ? (r_ & 3) : ((r_ >> 1) & 1);
        const int vch = (((dcp >> 2) ^ sw) << 2) | (dcp & 3);
        vvoff[i] = (unsigned)r_ * v_ts2 + (unsigned)vch * 16u;
    }
    typedef const __attribute__((address_space(3))) char* lptr_c;
    const unsigned kt0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lptr_c)ktile);
    const unsigned vt0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lptr_c)vtile);
    // K fragment (A operand of S^T = K Q^T): key = 16 h + l15, dims 32 c + 8 g4 .. + 8 = chunk 4 c + g4 of the row
    unsigned kaddr[2];  // byte address of chunk position 0 of this lane's row, key half h; the chunk's position is XORed in per read
    const int kswz = D >= 128 ? l15 : ((l15 >> 1) & 7);  // sw_k of rows l15 and 16 + l15 alike
#pragma unroll
    for (int h = 0; h < 2; ++h) kaddr[h] = (unsigned)(uintptr_t)(lptr_c)(ktile + (16 * h + l15) * RB);
    // V^T fragment (A operand of O^T += V^T P^T): d = 16 db + l15, keys {4 g4 + j} (half 0) / {16 + 4 g4 + j} (half 1);
    // the 16-lane group reads the 4 x 16 block, lane l15 supplies row l15 >> 2, columns 4 (l15 & 3) .. + 4
    const int trow = 4 * g4 + (l15 >> 2);
    const int tsw = D >= 128 ? (trow & 3) : ((trow >> 1) & 1);
    unsigned vaddr[NDB];
#pragma unroll
    for (int db = 0; db < NDB; ++db)
        vaddr[db] = (unsigned)(uintptr_t)(lptr_c)(vtile + trow * RB + (((db >> 1) ^ tsw) << 6) + 32 * (db & 1) + 8 * (l15 & 3));

    f32x4 o[NDB];
#pragma unroll
    for (int db = 0; db < NDB; ++db) o[db] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;
    const float sc = a.scale_log2e;

    // ---- one 32-key step --------------------------------------------------------------------------------------------------
    // issue:   2 NVD LDS-DMA instructions, K rows -> K tile, V rows -> V tile (zero fill past the segment's end)
    // collect: the landed tiles -> registers (K: 2 NCH ds_read_b128 in the A-operand layout; V^T: 2 NDB transposing reads), after
    //          which both tiles are free for the next step's DMA
    // compute: S^T = K Q^T, online softmax, O^T += V^T P^T, all from registers
    // The loop keeps ONE step in flight per wave while the previous one is computed (collect(j); issue(j + 1); compute(j); head dim 64: TWO, NSET):
    // 16 KiB per wave, 8 waves per CU.  (Round 4's form loaded K straight into MFMA operand layout -- 16 rows x 64 B per
    // instruction, every quarter wave 16 different cache lines -- and ran 10 % below this one on 8-kv-head shapes.)
    u32x4 kf[2][NCH];
    u32x4 vf[NDB];
    // (so: byte offset of the tile set, 0 when there is one)
    auto issue = [&](int key0, unsigned so) __attribute__((always_inline)) {
        const unsigned ksoff = (unsigned)key0 * k_ts2, vsoff = (unsigned)key0 * v_ts2;
#pragma unroll
        for (int i = 0; i < NVD; ++i) dma16_g<NT>(krs_c, kvoff[i], ksoff, kt0 + so + i * 1024);
#pragma unroll
        for (int i = 0; i < NVD; ++i) dma16_g<NT>(vrs_c, vvoff[i], vsoff, vt0 + so + i * 1024);
    };
    auto collect = [&](unsigned so) __attribute__((always_inline)) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int c = 0; c < NCH; ++c)
                kf[h][c] = *reinterpret_cast<const __attribute__((address_space(3))) u32x4*>((uintptr_t)(kaddr[h] + so + (unsigned)(((4 * c + g4) ^ kswz) << 4)));
#pragma unroll
        for (int db = 0; db < NDB; ++db) {
            const u32x2 t0 = lds_tr16_g(vaddr[db] + so);
            const u32x2 t1 = lds_tr16_g(vaddr[db] + so + 16 * RB);
            vf[db] = u32x4{t0[0], t0[1], t1[0], t1[1]};
        }
        // every read has returned before the tiles are handed to the next DMA (asm: hipcc does not order it against them otherwise)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    // a step that reaches past the segment's keys: zero the V rows of keys >= seg_len (dword w of a V^T fragment holds keys
    // key0 + 4 g4 + {0,1} / {2,3} / 16 + {0,1} / 16 + {2,3}, the first in its low half)
    auto sanitize = [&](int key0) __attribute__((always_inline)) {
        const int kb = key0 + 4 * g4;
        unsigned msk[4];
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const int ka = kb + (w >> 1) * 16 + (w & 1) * 2;
            msk[w] = (ka < seg_len ? 0x0000ffffu : 0u) | (ka + 1 < seg_len ? 0xffff0000u : 0u);
        }
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
            for (int w = 0; w < 4; ++w) vf[db][w] &= msk[w];
    };
    auto compute = [&](int key0) __attribute__((always_inline)) {
        f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            mfma16_acc<T>(s0, kf[0][c], qf[c]);
            mfma16_acc<T>(s1, kf[1][c], qf[c]);
        }
        // the MFMAs above are asm: hipcc's hazard recogniser does not see that their results need the pipeline drained
        asm volatile("s_nop 7\n\ts_nop 7" : "+v"(s0), "+v"(s1));
        // scores of this lane: keys key0 + 4 g4 + i (s0) and key0 + 16 + 4 g4 + i (s1), query row l15
        float p[8];
        const int kb = key0 + 4 * g4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            p[i] = (kb + i < seg_len) ? s0[i] * sc : -INFINITY;
            p[4 + i] = (kb + 16 + i < seg_len) ? s1[i] * sc : -INFINITY;
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
            o[db] *= alpha;
            mfma16_acc<T>(o[db], vf[db], pf);
        }
        // the accumulators are read by plain VALU code next (the rescale of the next step, a fold, the epilogue): drain the matrix pipeline
        asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
    };

    // ---- folding a prefix partial into this wave's state -----------------------------------------------------------------
    auto fold = [&](float lse_p, const f32x4(&x)[NDB]) __attribute__((always_inline)) {
        const float m_p = lse_p * 1.4426950408889634f;
        const float mf = fmaxf(m_run, m_p);
        const float ms = (mf == -INFINITY) ? 0.f : mf;
        const float a1 = fast_exp2(m_run - ms), a2 = fast_exp2(m_p - ms);
        l_run = l_run * a1 + a2;
        m_run = mf;
#pragma unroll
        for (int db = 0; db < NDB; ++db) o[db] = o[db] * a1 + x[db] * a2;
    };
    auto widen = [&](const u32x2(&u)[NDB], f32x4(&x)[NDB]) __attribute__((always_inline)) {
#pragma unroll
        for (int db = 0; db < NDB; ++db) x[db] = f32x4{TR::lo(u[db][0]), TR::hi(u[db][0]), TR::lo(u[db][1]), TR::hi(u[db][1])};
    };
    auto fetch = [&](const PartialDev& pd, f32x4(&x)[NDB]) __attribute__((always_inline)) {
        if (pd.is_f32) {
#pragma unroll
            for (int db = 0; db < NDB; ++db)
                x[db] = *reinterpret_cast<const f32x4*>(static_cast<const float*>(pd.out) + ridx * D + 16 * db + 4 * g4);
        } else {
            u32x2 u[NDB];
#pragma unroll
            for (int db = 0; db < NDB; ++db)
                u[db] = *reinterpret_cast<const u32x2*>(static_cast<const uint16_t*>(pd.out) + ridx * D + 16 * db + 4 * g4);
            widen(u, x);
        }
    };
    // the pre-requested partials have landed (the caller waited for every outstanding load): fold them
    auto fold_pre = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < NPRE; ++k) {
            // opaque: hipcc must not hoist the conversions below (and with them its wait for the request) in front of the stream
            asm volatile("" : "+v"(plse[k]));
#pragma unroll
            for (int db = 0; db < NDB; ++db) asm volatile("" : "+v"(pbuf[k][db]));
            if (k < npre) {
                f32x4 x[NDB];
#pragma unroll
                for (int db = 0; db < NDB; ++db) {
                    const f32x4 xf = __builtin_bit_cast(f32x4, pbuf[k][db]);
                    const f32x4 xh = f32x4{TR::lo(pbuf[k][db][0]), TR::hi(pbuf[k][db][0]), TR::lo(pbuf[k][db][1]), TR::hi(pbuf[k][db][1])};
                    x[db] = pre_f32[k] ? xf : xh;
                }
                fold(plse[k], x);
            }
        }
        pre_folded = true;
    };
    int len = 0;
    for (int seg = has_pre ? 0 : 1; seg < 2; ++seg) {  // every segment drains its own pipeline
        constexpr int stride = 32 * WPU;
        const int k_first = wave * 32;
        if (seg == 0) { krs_c = krs_p; vrs_c = vrs_p; seg_cap = a.p_len; }
        else { krs_c = krs; vrs_c = vrs; seg_cap = a.kv_len; }
        // The first step goes out as soon as the sequence's length has arrived (a scalar load: its wait does not touch the vector
        // queue), in front of the wait for q and the first partials, and its windows END at the length: a step that reaches past
        // it is zero-filled by the address check instead of being fetched.  Until round 6 the first step was requested blind --
        // before the length, for whatever the cache's kv_len rows held -- and the windows stayed that wide: every unit read up to 31
        // rows behind its length in its last step (1.166 x the algorithmic bytes at C5's mean suffix of 128.5, FETCH_SIZE in
        // profiles/r06_v4_c5_whole_job.json) and a full 32 rows in the first 31 decode steps of every generation (C5 at S = 20:
        // 368 MB moved for 268 MB).  The round trip the blind step saved is 0.5 us per launch (profiles/r06_gqa_blind_step_ab.txt).
        bool blind = false;
#ifdef HYD_ABLATION_BUILD
        blind = a.dbg_blind != 0;  // HYD_GQA_BLIND=1: the blind first step, windows still cut at the length afterwards (A/B)
#endif
        auto learn_len = [&]() __attribute__((always_inline)) {
            if (seg == 1) {
                len = max(0, min(len_raw, a.kv_len));
                krs_c[2] = __builtin_amdgcn_readfirstlane((unsigned)len * k_ts2);
                vrs_c[2] = __builtin_amdgcn_readfirstlane((unsigned)len * v_ts2);
            }
            seg_len = seg == 0 ? a.p_len : len;
        };
        int nst = 0;  // 32-key steps of this wave
        if (blind) {
            if (seg_cap > k_first) issue(k_first, 0u);
        } else {
            learn_len();
            nst = seg_len > k_first ? (seg_len - k_first + stride - 1) / stride : 0;
            if (nst > 0) issue(k_first, 0u);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // ... all landed
        // hipcc counts only its own loads (q, the first partials): make it settle them HERE, where nothing is in flight -- a
        // counted wait of its own further down would also wait for the next step's DMAs, which it does not know about
#pragma unroll
        for (int c = 0; c < NCH; ++c) asm volatile("" ::"v"(qf[c]));
        if (blind) {
            if (seg == 1) asm volatile("" : "+v"(len_raw));
            learn_len();
            nst = seg_len > k_first ? (seg_len - k_first + stride - 1) / stride : 0;
        }
        if (!pre_folded) fold_pre();
        // the second set's first step goes out BEHIND the drain: hipcc's own wait for q and the partials (it lands at the settle point
        // above) is a full one and must not find a key step it does not know about in the queue
        if (NSET == 2 && nst > 1) issue(k_first + stride, SETB);
        for (int j = 0; j < nst; ++j) {
            const int key0 = k_first + j * stride;
            const unsigned so = NSET == 2 ? (unsigned)(j & 1) * SETB : 0u;
            if (j > 0) {  // step j's tiles have landed (requests return in order: with two sets, step j + 1's 2 NVD may stay out)
                if (NSET == 2 && j + 1 < nst) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NVD) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            collect(so);
            if (key0 + 32 > seg_len) sanitize(key0);
            if (j + NSET < nst) issue(key0 + NSET * stride, so);  // in flight while steps j (and j + 1) are computed
            compute(key0);
        }
    }
    // the keys-only state (the LSE output is the suffix pass's own; when it is asked for, no partial was folded above)
    const float m_s = m_run, l_s = l_run;

    // ---- this wave's other partials (all of them when the LSE output is asked for) ---------------------------------------
    for (int i = wave + max(npre, 0) * WPU; i < np; i += 2 * WPU) {
        // two partials per round trip: their lse values and 2 * D/16 row pieces are all requested before the first use
        const bool two = i + WPU < np;
        const int i1 = two ? i + WPU : i;
        const float lse0 = a.partials[i].lse[ridx];
        const float lse1 = two ? a.partials[i1].lse[ridx] : -INFINITY;
        f32x4 x0[NDB], x1[NDB];
        fetch(a.partials[i], x0);
        fetch(a.partials[i1], x1);
        fold(lse0, x0);
        fold(lse1, x1);
    }

    // ---- merge the WPU waves of the unit: every wave leaves (m, l, O^T) in its own V tile, wave 0 folds them -------
    float ms_run = m_s, ls_run = l_s;
    if constexpr (WPU > 1) {
        float* mine = reinterpret_cast<float*>(vtile);  // [16 rows][D]: O (unnormalised)
        // lane (row l15, key group g4) holds O^T[d = 16 db + 4 g4 + i][row l15] in o[db][i]; m / l are equal over g4
#pragma unroll
        for (int db = 0; db < NDB; ++db)
            *reinterpret_cast<f32x4*>(mine + l15 * D + 16 * db + 4 * g4) = o[db];
        if (g4 == 0) {
            mlx[wave][0][l15] = m_run;
            mlx[wave][1][l15] = l_run;
            mlx[wave][2][l15] = m_s;
            mlx[wave][3][l15] = l_s;
        }
        __syncthreads();
        if (wave != 0) return;
#pragma unroll
        for (int w = 1; w < WPU; ++w) {
            const float* oth = reinterpret_cast<const float*>(tiles[w][0][1]);
            const float m2 = mlx[w][0][l15], l2 = mlx[w][1][l15];
            const float mf = fmaxf(m_run, m2);
            const float ms = (mf == -INFINITY) ? 0.f : mf;
            const float a1 = fast_exp2(m_run - ms), a2 = fast_exp2(m2 - ms);
            l_run = l_run * a1 + l2 * a2;
            m_run = mf;
            const float m3 = mlx[w][2][l15], l3 = mlx[w][3][l15];
            const float mg = fmaxf(ms_run, m3);
            const float mgs = (mg == -INFINITY) ? 0.f : mg;
            ls_run = ls_run * fast_exp2(ms_run - mgs) + l3 * fast_exp2(m3 - mgs);
            ms_run = mg;
#pragma unroll
            for (int db = 0; db < NDB; ++db) o[db] = o[db] * a1 + *reinterpret_cast<const f32x4*>(oth + l15 * D + 16 * db + 4 * g4) * a2;
            asm volatile("" ::: "memory");  // one wave's tile at a time: hoisting all three costs 96 registers
        }
    }

    // ---- epilogue: normalise and store (the prefix partials are already in) -----------------------------------------
    if (!rvalid) return;
    if (a.lse && g4 == 0) a.lse[ridx] = ls_run > 0.f ? ms_run * kLn2 + __logf(ls_run) : -INFINITY;
    const float inv = l_run > 0.f ? 1.0f / l_run : 0.f;
#pragma unroll
    for (int db = 0; db < NDB; ++db) {
        const f32x4 x = o[db] * inv;
        const u32x2 pk = {TR::pack2(x[0], x[1]), TR::pack2(x[2], x[3])};
        *reinterpret_cast<u32x2*>(static_cast<uint16_t*>(a.out) + ridx * D + 16 * db + 4 * g4) = pk;
    }
}

// shapes only: fewer than 4 one-wave units per CU (measured: B=32, 8/1 heads, 1152 keys 48 -> 31 us; at 1024 and
// 2048 units -- C3, C5 -- one wave per unit is as fast or faster), and enough keys to deal out
static bool gqa_few_units(const SuffixArgs& a, int chunks) {
    return (int64_t)a.units * chunks < 256 * 4 && a.kv_len + (a.pk ? a.p_len : 0) >= 128;
}

// Shapes-only eligibility (capture-safe): enough query rows per unit for the matrix cores to pay, enough units to
// fill the chip with one wave each, and byte offsets inside a unit's cache that fit the 32-bit buffer addressing.
bool suffix_gqa_eligible(const SuffixArgs& a, int D, bool any_shape) {
    if (D != 64 && D != 128 && D != 256) return false;
    const int64_t chunks = (a.rows + 15) / 16;
    // measured on MI355X (tools/kbench.py, HYD_SUFFIX_IMPL=valu|gqa; round 5's kernel): from 3 rows per unit the matrix cores win at
    // every suffix length (12/4 heads, B = 1024: 10.2 / 23.4 / 82.5 us at S = 8 / 64 / 256 against 12.0 / 30.6 / 104); with 2 rows they win
    // from S = 64 on and lose below (16/8 heads: 20 vs 14 us at S = 8), with 1 row only from S = 128: those stay on the dot-product kernel
    if (!any_shape && a.rows < 3) return false;
    // head dim 256 (32 KB of tiles and one SIMD per wave): measured against the dot-product kernel, 7 x faster with one wave per
    // unit (B = 2048, 8/1 heads, S = 256: 667 -> 90 us) but 15 % slower in the four-waves-per-unit form that few units take
    if (D == 256 && gqa_few_units(a, (int)chunks)) return false;
    const int64_t span = (int64_t)a.kv_len * (a.k_ts > a.v_ts ? a.k_ts : a.v_ts) * 2;
    return span < (int64_t)1 << 31 && a.Hkv <= 65535 && chunks <= 65535;
}

template <typename T, int D, int WPU, bool NT, int HPW>
static int launch_gqa_k(const SuffixArgs& a, dim3 grid, size_t pad, hipStream_t s) {
    constexpr int NWV = WPU * HPW;
    // the K/V landing tiles, plus the (m, l) exchange area that only the cross-wave merge of WPU > 1 touches: one-wave units ask
    // for the tiles alone, which is at most 64 KB for every shape picked from shapes (HPW = 4 at D = 128, HPW = 2 at D = 256)
    constexpr size_t lds = (size_t)NWV * (D == 64 ? 2 : 1) * 2 * 32 * D * 2 + (WPU > 1 ? (size_t)NWV * 4 * 16 * sizeof(float) : 0);
    auto kern = suffix_attn_gqa_kernel<T, D, WPU, NT, HPW>;
    if (lds + pad > 64 * 1024) {
        // more than the default dynamic-LDS limit (four-wave units; development pads): raise it, once per DEVICE and instantiation --
        // the attribute belongs to the function as loaded on the current device, and a process may launch on several
        static hipError_t attr_rc[16];
        static bool attr_set[16];
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return (int)hipErrorInvalidDevice;
        if (!__atomic_load_n(&attr_set[dev], __ATOMIC_ACQUIRE)) {  // idempotent: two threads racing both set the same value
            attr_rc[dev] = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
            __atomic_store_n(&attr_set[dev], true, __ATOMIC_RELEASE);
        }
        if (attr_rc[dev] != hipSuccess) return (int)attr_rc[dev];
    }
    hipLaunchKernelGGL(kern, grid, dim3(64 * NWV), lds + pad, s, a);
    return (int)hipGetLastError();
}

template <typename T, int D>
static int launch_gqa_t(const SuffixArgs& a_in, hipStream_t s) {
    SuffixArgs a = a_in;
    const int chunks = (a.rows + 15) / 16;
    bool few_units = gqa_few_units(a, chunks);
#ifdef HYD_ABLATION_BUILD
    if (const char* e = getenv("HYD_GQA_WPU")) few_units = atoi(e) == 4;
    if (const char* e = getenv("HYD_GQA_BLIND")) a.dbg_blind = atoi(e);
#endif
    // kv heads of a sequence per workgroup (one-wave units of the unique phase).  Measured with this kernel (profiles/r05_gqa_hpw.txt,
    // us at S = 32 / 128 / 256): 8 kv heads: 1 head per workgroup 61 / 197 / 364, 2: 60 / 185 / 355, 4: 64 / 184 / 352, 8: 71 / 187 / 341 -- a
    // workgroup that fills the CU's LDS alone (8 x 16 KB) leaves it idle between workgroups, which short suffixes pay for; 16 kv heads:
    // 1: 55 / 181 / 348, 4: 63 / 211 / 403, 8: 70 / 223 / 446 -- a part of a token's row per workgroup is worse than one head.  So: all
    // heads of the token in one workgroup when there are at most 4 (<= 64 KB of tiles: two workgroups per CU), 4 of 8, else one.
    int hpw = 1;
    if (!few_units && !a.shared_kv && !a.pk && a.Hkv <= 8) {
        hpw = a.Hkv % 4 == 0 ? 4 : a.Hkv % 2 == 0 ? 2 : 1;
        if (D == 256 && hpw > 2) hpw = 2;  // 32 KB of tiles per wave
    }
#ifdef HYD_ABLATION_BUILD
    if (const char* e = getenv("HYD_GQA_HPW")) hpw = few_units ? 1 : atoi(e);
#endif
    dim3 grid((unsigned)a.B * (unsigned)(a.Hkv / hpw), chunks, 1);
    size_t pad = 0;  // development: dynamic LDS that only lowers the occupancy (waves per CU = 160 KiB / (16 KiB + pad))
#ifdef HYD_ABLATION_BUILD
    if (const char* e = getenv("HYD_GQA_LDS_PAD")) pad = (size_t)atoi(e);
#endif
    if constexpr (D == 256) {
        if (few_units || a.shared_kv) return (int)hipErrorInvalidValue;  // (not eligible: suffix_gqa_eligible, level_is_small)
    } else {
        if (a.shared_kv) {  // a small shared level: its keys are read by several workgroups, default cache policy
            if (few_units) return launch_gqa_k<T, D, 4, false, 1>(a, grid, 0, s);
            return launch_gqa_k<T, D, 1, false, 1>(a, grid, pad, s);
        }
        if (few_units) return launch_gqa_k<T, D, 4, true, 1>(a, grid, 0, s);
    }
    switch (hpw) {
#ifdef HYD_ABLATION_BUILD
        case 8:  // (A/B only: not chosen from shapes any more)
            if constexpr (D < 256) return launch_gqa_k<T, D, 1, true, 8>(a, grid, 0, s);
            else return (int)hipErrorInvalidValue;
#endif
        case 4:
            if constexpr (D < 256) return launch_gqa_k<T, D, 1, true, 4>(a, grid, 0, s);
            else return (int)hipErrorInvalidValue;
        case 2: return launch_gqa_k<T, D, 1, true, 2>(a, grid, 0, s);
        default: return launch_gqa_k<T, D, 1, true, 1>(a, grid, pad, s);
    }
}

int launch_suffix_gqa(const SuffixArgs& a, int dtype, int D, hipStream_t s) {
    if (dtype == HYD_F16) {
        if (D == 128) return launch_gqa_t<F16, 128>(a, s);
        if (D == 64) return launch_gqa_t<F16, 64>(a, s);
        if (D == 256) return launch_gqa_t<F16, 256>(a, s);
    } else {
        if (D == 128) return launch_gqa_t<BF16, 128>(a, s);
        if (D == 64) return launch_gqa_t<BF16, 64>(a, s);
        if (D == 256) return launch_gqa_t<BF16, 256>(a, s);
    }
    return (int)hipErrorInvalidValue;
}

}  // namespace hyd
