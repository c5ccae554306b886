// Prefix pass (K1 / K1v / K2c of SURVEY.md): batched-query attention of all queries of a group
// against the group's single shared K/V, on the gfx950 matrix cores.
//
// Replaces flash-attn's _flash_attn_forward / _flash_attn_varlen_forward as called from
// /root/reference/hydragen/flash.py:284-351 and hydragen/attention.py:270,313,344.
//
// Work decomposition (wave64, 512-thread workgroups = 8 waves = 2 per SIMD):
//   workgroup  = 128 folded query rows (b_local, iq, gqa-head) of one (group, kv-head, kv-split)
//   wave w     = row sub-block rw = w & 3 (32 rows)  x  key half kg = w >> 2 of every 128-key tile
//   per 64-key half tile and wave:
//       S^T[key][row]  = K . Q^T       16 x v_mfma_f32_32x32x16 (A = K fragment from LDS (ds_read_b128,
//                                      XOR-swizzled rows), B = Q fragment held in registers)
//       online softmax in registers: a lane owns one query row (lane & 31) and 32 of its 64 scores,
//                                    the partner lane (lane ^ 32) owns the other 32
//       O^T[d][row]   += V^T . P^T     16 x MFMA (A = V^T fragment via ds_read_b64_tr_b16 from the
//                                      row-major V tile, B = P^T converted in registers; the
//                                      contraction index is permuted so no cross-lane moves are needed)
//   the two key halves keep independent (m, l, O) and are merged through LDS at the end.
// K/V tiles go global -> LDS by DMA (global_load_lds) into a double buffer: tile t+1 lands while tile t
// is computed, one barrier per tile.  HBM/L2: blocks that share a (group, kv-head) are remapped onto one XCD.
#include <type_traits>

#include "hyd_kernels.h"

namespace hyd {

template <int D>
__device__ __forceinline__ int kswz(int row, int ch) {
    return D == 128 ? (ch ^ (row & 15)) : (ch ^ ((row >> 1) & 7));
}
template <int D>
__device__ __forceinline__ int vswz(int row, int ch) {
    const int s = D == 128 ? (row & 3) : ((row >> 1) & 1);
    return (((ch >> 2) ^ s) << 2) | (ch & 3);
}

typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;

__device__ __forceinline__ u32x2 lds_tr16(unsigned lds_byte_addr) {
    s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(uintptr_t)lds_byte_addr);
    return __builtin_bit_cast(u32x2, t);
}

// One LDS-DMA instruction: 64 lanes x 16 B from per-lane global addresses to LDS [dst, dst + 1 KiB)
// (dst wave-uniform).  Issued from inline asm on purpose: hipcc then does not know an LDS write is in
// flight, so it neither drains it (vmcnt(0)) in front of the transposing reads of the current tile nor
// counts it; the tile loop waits for it explicitly (dma_wait_all) right before its barrier.
__device__ __forceinline__ void dma16(const char* gsrc, const char* lds_dst) {
    const unsigned l = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)lds_dst);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(l)
                 : "memory");
}
__device__ __forceinline__ void dma_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// ABL: development-only ablation mask (bit0 no in-loop DMA, bit1 no QK, bit2 no softmax, bit3 no PV);
// only ABL = 0 is ever used for results.
template <typename T, int D, bool CAUSAL, int ABL = 0>
__global__ __launch_bounds__(512) void prefix_attn_kernel(const PrefixArgs a) {
    using TR = Traits<T>;
    constexpr int RB = D * 2;            // bytes per K/V row
    constexpr int CPR = D / 8;           // 16-byte chunks per row
    constexpr int NC = D / 16;           // k-chunks of the QK^T contraction
    constexpr int NDB = D / 32;          // 32-wide d blocks of O^T
    constexpr int NLD = (128 * CPR) / 512;  // 16-byte chunks per thread per tensor per tile
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* mlbuf = reinterpret_cast<float*>(smem + 512 * RB);  // [4][2][64], after the K/V buffers

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // compute roles: rw = 32-row sub-block, kg = key half.  DMA roles (kgd) are fixed by wave index.
    const int kg = wave >> 2;
    const int rw = wave & 3;
    const int kgd = wave >> 2;
    const int l31 = lane & 31, hi = lane >> 5;

    // ---- which (group, kv head, split, row block) ------------------------------------------
    const int lin = xcd_remap(blockIdx.x, gridDim.x);
    const int rb = lin % a.row_blocks;
    int t = lin / a.row_blocks;
    const int sp = t % a.nsplit;
    t /= a.nsplit;
    const int hk = t % a.Hkv;
    const int gi = t / a.Hkv;

    int q_tok0, nqtok, nq_eff;
    if (a.cu_q) {
        q_tok0 = a.cu_q[gi];
        nqtok = a.cu_q[gi + 1] - q_tok0;
        nq_eff = nqtok;
    } else {
        q_tok0 = gi * a.per * a.nq;
        nqtok = a.per * a.nq;
        nq_eff = a.nq;
    }
    const int Mrows = nqtok * a.g;
    if (rb * 128 >= Mrows) return;  // block-uniform

    const uint16_t* k16 = static_cast<const uint16_t*>(a.k);
    const uint16_t* v16 = static_cast<const uint16_t*>(a.v);
    int L;
    if (a.cu_k) {
        const int t0 = a.cu_k[gi];
        L = a.cu_k[gi + 1] - t0;
        k16 += (int64_t)t0 * a.k_ts;
        v16 += (int64_t)t0 * a.v_ts;
    } else {
        L = a.kv_len;
        k16 += (int64_t)gi * a.k_gs;
        v16 += (int64_t)gi * a.v_gs;
    }
    k16 += (int64_t)hk * a.k_hs;
    v16 += (int64_t)hk * a.v_hs;

    const int kbeg = sp * a.split_len;
    int kend = min(L, kbeg + a.split_len);
    if (CAUSAL && a.per == 1) {
        // rows of this block only see keys <= iq_max + L - nq
        const int rmax = min(Mrows, rb * 128 + 128) - 1;
        kend = min(kend, rmax / a.g + L - nq_eff + 1);
    }
    const int nkt = kend > kbeg ? (kend - kbeg + 127) >> 7 : 0;

    // ---- this lane's query row ------------------------------------------------------------
    const int r = rb * 128 + rw * 32 + l31;
    const bool rvalid = r < Mrows;
    const int rtok = rvalid ? r / a.g : 0;  // query token inside the group
    const int hq = hk * a.g + (rvalid ? r % a.g : 0);
    const int64_t row_off = ((int64_t)(q_tok0 + rtok) * a.Hq + hq) * D;
    int row_lim = 0x3fffffff;  // last visible key (causal)
    if (CAUSAL) row_lim = (rtok % nq_eff) + L - nq_eff;

    u32x4 qf[NC];
    {
        const uint16_t* qrow = static_cast<const uint16_t*>(a.q) + row_off;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            u32x4 z = {0u, 0u, 0u, 0u};
            qf[c] = rvalid ? *reinterpret_cast<const u32x4*>(qrow + 16 * c + 8 * hi) : z;
        }
    }

    // ---- LDS map (bytes): KA[2] | KB[2] | V[2] | mlbuf ----------------------------------------------
    // KA[i]: rows 0..63 of K tile (kg = 0 waves), KB[i]: rows 64..127 (kg = 1 waves), V[i]: 128 rows.
    constexpr int KH_BYTES = 64 * RB;
    constexpr int V_BYTES = 128 * RB;
    constexpr int KA_OFF = 0, KB_OFF = 2 * KH_BYTES, V_OFF = 4 * KH_BYTES;
    typedef const __attribute__((address_space(3))) char* lptr_c;

    // ---- per-lane LDS byte addresses; tile/buffer/row-block offsets are compile-time immediates -------
    const int ksw = D == 128 ? (l31 & 15) : ((l31 >> 1) & 7);
    const int kx = hi ^ ksw;
    unsigned kaddr[NC];  // K fragment c of row l31, relative to the start of a 64-row half tile
#pragma unroll
    for (int c = 0; c < NC; ++c) kaddr[c] = (unsigned)(uintptr_t)(lptr_c)(smem + l31 * RB + (((2 * c) ^ kx) << 4));
    const int i16 = lane & 15, g16 = lane >> 4;
    const int vsw = D == 128 ? (i16 >> 2) : ((i16 >> 3) & 1);
    unsigned vaddr[NDB];  // V^T fragment address inside V buffer 0 for key slot 0
#pragma unroll
    for (int db = 0; db < NDB; ++db)
        vaddr[db] = (unsigned)(uintptr_t)(lptr_c)(smem + V_OFF + (kg * 64 + 4 * hi + (i16 >> 2)) * RB +
                                                  ((db ^ vsw) << 6) + 32 * (g16 & 1) + 8 * (i16 & 3));

    // ---- staging: global -> LDS DMA (global_load_lds, 16 B per lane, 1 KiB per wave instruction) -------
    // The LDS image of a wave instruction is lane-linear (base + lane*16), so the XOR swizzles are
    // applied to the per-lane SOURCE chunk (they are involutions inside a row).  Wave w fills rows
    // [16w, 16w+16) of the 128-row K image (waves 0-3 -> KA half, waves 4-7 -> KB half) and of the V tile.
    // Per-lane byte offsets inside a tile are fixed; the tile base is wave-uniform.
    constexpr int RPI = 1024 / RB;  // tile rows per wave instruction
    const int drow = (lane * 16) / RB;                // row inside the instruction
    const int dcp = ((lane * 16) % RB) >> 4;          // 16-byte slot inside the row (LDS side)
    unsigned koff[NLD], voff[NLD];  // < 128 rows * token stride * 2 B: always fits 32 bits
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int row = (wave * NLD + i) * RPI + drow;  // row of the 128-row tile
        const int kch = D == 128 ? (dcp ^ (row & 15)) : (dcp ^ ((row >> 1) & 7));
        const int vs_ = D == 128 ? (row & 3) : ((row >> 1) & 1);
        const int vch = (((dcp >> 2) ^ vs_) << 2) | (dcp & 3);
        koff[i] = (unsigned)(((int64_t)row * a.k_ts + kch * 8) * 2);
        voff[i] = (unsigned)(((int64_t)row * a.v_ts + vch * 8) * 2);
    }
    const char* kbase = reinterpret_cast<const char*>(k16) + (int64_t)kbeg * a.k_ts * 2;
    const char* vbase = reinterpret_cast<const char*>(v16) + (int64_t)kbeg * a.v_ts * 2;
    const int nkeys = kend - kbeg;
    // One DMA instruction ("piece") at a time, so the tile loop can spread them between its MFMAs: issued
    // back to back they fill the VMEM queue and the wave sits at issue until the address path has taken
    // them all (the fetch then runs in series with the compute instead of underneath it).
    // pieces [0, NLD): K half tile of tile tk (or none if tk < 0); pieces [NLD, 2*NLD): V tile tv.
    int dma_tk = -1, dma_tv = -1;
    auto dma_piece_c = [&](auto J_C) {
        constexpr int j = decltype(J_C)::value;
        if constexpr (j < NLD) {
            constexpr int i = j;
            const int tk = dma_tk;
            if (tk < 0) return;
            char* Kd = smem + (kgd ? KB_OFF : KA_OFF) + (tk & 1) * KH_BYTES + (wave & 3) * NLD * 1024;
            const char* tb = kbase + (int64_t)tk * 128 * a.k_ts * 2;  // wave-uniform
            // last, partial tile: rows past the end re-read the last valid key (scores masked)
            const int over = (tk * 128 + 128 <= nkeys) ? 0 : max(0, tk * 128 + (wave * NLD + i) * RPI + drow - (nkeys - 1));
            dma16(tb + koff[i] - (int64_t)over * a.k_ts * 2, Kd + i * 1024);
        } else if constexpr (j < 2 * NLD) {
            constexpr int i = j - NLD;
            const int tv = dma_tv;
            if (tv < 0) return;
            char* Vd = smem + V_OFF + (tv & 1) * V_BYTES + wave * NLD * 1024;
            const char* tb = vbase + (int64_t)tv * 128 * a.v_ts * 2;
            const int over = (tv * 128 + 128 <= nkeys) ? 0 : max(0, tv * 128 + (wave * NLD + i) * RPI + drow - (nkeys - 1));
            dma16(tb + voff[i] - (int64_t)over * a.v_ts * 2, Vd + i * 1024);
        }
    };
    auto dma_piece = [&](int j) {  // j is a constant after unrolling; the switch keeps every index static
        using std::integral_constant;
        switch (j) {
            case 0: dma_piece_c(integral_constant<int, 0>{}); break;
            case 1: dma_piece_c(integral_constant<int, 1>{}); break;
            case 2: dma_piece_c(integral_constant<int, 2>{}); break;
            case 3: dma_piece_c(integral_constant<int, 3>{}); break;
            case 4: dma_piece_c(integral_constant<int, 4>{}); break;
            case 5: dma_piece_c(integral_constant<int, 5>{}); break;
            case 6: dma_piece_c(integral_constant<int, 6>{}); break;
            case 7: dma_piece_c(integral_constant<int, 7>{}); break;
            default: break;
        }
    };
    auto dma = [&](int tk, int tv) {  // whole tile at once (prologue)
        dma_tk = tk;
        dma_tv = tv;
#pragma unroll
        for (int j = 0; j < 2 * NLD; ++j) dma_piece(j);
    };

    f32x16 o[NDB];
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[db][i] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const float sc = a.scale_log2e;
    f32x16 s[2];
    u32x4 pf[4];
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    // ---- the three phases of a 64-key half tile -------------------------------------------------------
    // S^T = K Q^T.  K fragments are prefetched PD steps ahead of the MFMA that consumes them (hipcc
    // otherwise serialises ds_read -> s_waitcnt -> mfma, exposing the LDS latency 16 times per tile).
    // KOFF = compile-time byte offset of the half tile inside LDS.
    auto qk_phase = [&](auto KOFF_C, auto P0_C) {
        constexpr int KOFF = decltype(KOFF_C)::value;
        constexpr int P0 = decltype(P0_C)::value;  // first DMA piece to issue here, or -1 for none
        constexpr int PD = 4;
        u32x4 kfr[PD];
        // step i -> (c = i / 2, kb = i % 2): consecutive MFMAs alternate between the two accumulators (an
        // extra issue slot between two MFMAs on the SAME accumulator costs ~40 cycles, MI355X_MICROARCH.md)
        auto ldk = [&](int i) -> u32x4 {
            return *reinterpret_cast<const __attribute__((address_space(3))) u32x4*>(
                (uintptr_t)(kaddr[i / 2] + KOFF + (i % 2) * 32 * RB));
        };
#pragma unroll
        for (int i = 0; i < PD; ++i) kfr[i] = ldk(i);
        __builtin_amdgcn_sched_group_barrier(0x100, PD, 0);
#pragma unroll
        for (int i = 0; i < 2 * NC; ++i) {
            const int kb = i % 2, c = i / 2;
            s[kb] = TR::mfma32(kfr[i % PD], qf[c], c == 0 ? zero16 : s[kb]);
            if (i + PD < 2 * NC) kfr[i % PD] = ldk(i + PD);
            __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            constexpr int EVERY = (2 * NC) / NLD;
            if (P0 >= 0 && !(ABL & 1) && (i % EVERY) == EVERY - 1) dma_piece(P0 + i / EVERY);
        }
    };
    // masking + online softmax (base 2) + P^T fragments; whole-vector arithmetic so that hipcc emits the
    // packed fp32 forms (v_pk_fma_f32 / v_pk_add_f32) and v_max3_f32: this phase is VALU-issue bound.
    auto sm_phase = [&](int kw0) {
        int lim = kend - 1;
        if (CAUSAL) lim = min(lim, row_lim);
        const bool need_mask =
            (kw0 + 64 > kend) || (CAUSAL && __builtin_amdgcn_ballot_w64(row_lim < kw0 + 63) != 0ull);
        if (need_mask) {
            const int lr = lim - kw0 - 4 * hi;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    if (kb * 32 + 8 * (i >> 2) + (i & 3) > lr) s[kb][i] = -INFINITY;
        }
        float tmax = fmaxf(s[0][0], s[1][0]);
#pragma unroll
        for (int i = 1; i < 16; ++i) tmax = fmaxf(fmaxf(tmax, s[0][i]), s[1][i]);
        tmax = pair_max(tmax);
        const float m_new = fmaxf(m_run, tmax * sc);
        const float msafe = (m_new == -INFINITY) ? 0.f : m_new;
        const float alpha = fast_exp2(m_run - msafe);
        s[0] = s[0] * sc - msafe;
        s[1] = s[1] * sc - msafe;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            s[0][i] = fast_exp2(s[0][i]);
            s[1][i] = fast_exp2(s[1][i]);
        }
        const f32x16 t16 = s[0] + s[1];
        const f32x8 t8 = t16.lo + t16.hi;
        const f32x4 t4 = t8.lo + t8.hi;
        const f32x2 t2 = t4.lo + t4.hi;
        l_run = l_run * alpha + (t2[0] + t2[1]);
        if (__builtin_amdgcn_ballot_w64(m_new > m_run) != 0ull) {
#pragma unroll
            for (int db = 0; db < NDB; ++db) o[db] *= alpha;
        }
        m_run = m_new;
        // slot ks (16 keys) <- regs [8*(ks&1), +8) of block ks>>1
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int kb = ks >> 1, b0 = 8 * (ks & 1);
            pf[ks][0] = TR::pack2(s[kb][b0 + 0], s[kb][b0 + 1]);
            pf[ks][1] = TR::pack2(s[kb][b0 + 2], s[kb][b0 + 3]);
            pf[ks][2] = TR::pack2(s[kb][b0 + 4], s[kb][b0 + 5]);
            pf[ks][3] = TR::pack2(s[kb][b0 + 6], s[kb][b0 + 7]);
        }
    };
    // O^T += V^T P^T (V^T fragments by the transposing LDS read; hipcc pipelines them with counted waits).
    // VOFF = compile-time byte offset of the V buffer relative to V buffer 0.
    auto pv_phase = [&](auto VOFF_C, auto P0_C) {
        constexpr int VOFF = decltype(VOFF_C)::value;
        constexpr int P0 = decltype(P0_C)::value;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
            for (int db = 0; db < NDB; ++db) {
                const u32x2 t0 = lds_tr16(vaddr[db] + VOFF + (16 * ks) * RB);
                const u32x2 t1 = lds_tr16(vaddr[db] + VOFF + (16 * ks + 8) * RB);
                const u32x4 vf = {t0[0], t0[1], t1[0], t1[1]};
                o[db] = TR::mfma32(vf, pf[ks], o[db]);
            }
            if (P0 >= 0 && !(ABL & 1)) {  // spread NLD DMA pieces over the 4 key slots
                if (NLD == 4) dma_piece(P0 + ks);
                else if (ks % 2 == 1) dma_piece(P0 + ks / 2);
            }
        }
    };

    // ---- pipeline ---------------------------------------------------------------------------------
    // The two waves of a SIMD (same rows, kg = 0 / 1) run the phases in different orders so that one is
    // on the matrix pipe while the other is in the softmax VALU phase:
    //     kg = 0 :  QK(t)   SM(t)  PV(t)     | barrier
    //     kg = 1 :  SM(t)   PV(t)  QK(t+1)   | barrier     (its half of the K stream runs one tile ahead)
    // The tile loop is unrolled by two so every LDS buffer offset is an instruction immediate.
    using std::integral_constant;
    if (nkt > 0) {
        dma(0, 0);
        if (kgd == 1 && nkt > 1) dma(1, -1);
    }
    // Make the compiler wait for the Q fragments here, not (conservatively, with vmcnt(0)) inside the
    // loop where it would also drain the DMA of the next K/V tiles.
#pragma unroll
    for (int c = 0; c < NC; ++c) asm volatile("" ::"v"(qf[c]));
    dma_wait_all();
    __syncthreads();
    if (kg == 1 && nkt > 0 && kbeg + 64 < kend) qk_phase(integral_constant<int, KB_OFF>{}, integral_constant<int, -1>{});
    __syncthreads();  // KB[0] is re-filled (tile 2) by the first loop iteration

    if (kg == 0) {
        auto step = [&](auto PAR_C, int kt) {
            constexpr int par = decltype(PAR_C)::value;
            dma_tv = kt + 1 < nkt ? kt + 1 : -1;
            dma_tk = kgd == 0 ? dma_tv : (kt + 2 < nkt ? kt + 2 : -1);
            const int kw0 = kbeg + kt * 128;  // always < kend for kg = 0
            if constexpr (!(ABL & 2)) qk_phase(integral_constant<int, KA_OFF + par * KH_BYTES>{}, integral_constant<int, 0>{});
            if constexpr (!(ABL & 4)) sm_phase(kw0);
            if constexpr (!(ABL & 8)) pv_phase(integral_constant<int, par * V_BYTES>{}, integral_constant<int, NLD>{});
            dma_wait_all();
            __syncthreads();
        };
        for (int kt = 0; kt < nkt; kt += 2) {
            step(integral_constant<int, 0>{}, kt);
            if (kt + 1 < nkt) step(integral_constant<int, 1>{}, kt + 1);
        }
    } else {
        auto step = [&](auto PAR_C, int kt) {
            constexpr int par = decltype(PAR_C)::value;
            dma_tv = kt + 1 < nkt ? kt + 1 : -1;
            dma_tk = kgd == 0 ? dma_tv : (kt + 2 < nkt ? kt + 2 : -1);
            const int kw0 = kbeg + kt * 128 + 64;
            if (kw0 < kend) {
                if constexpr (!(ABL & 4)) sm_phase(kw0);
                if constexpr (!(ABL & 8)) pv_phase(integral_constant<int, par * V_BYTES>{}, integral_constant<int, NLD>{});
            } else if constexpr (!(ABL & 1)) {
#pragma unroll
                for (int j = NLD; j < 2 * NLD; ++j) dma_piece(j);
            }
            if (kt + 1 < nkt && kw0 + 128 < kend && !(ABL & 2)) {
                qk_phase(integral_constant<int, KB_OFF + (par ^ 1) * KH_BYTES>{}, integral_constant<int, 0>{});
            } else if constexpr (!(ABL & 1)) {
#pragma unroll
                for (int j = 0; j < NLD; ++j) dma_piece(j);
            }
            dma_wait_all();
            __syncthreads();
        };
        for (int kt = 0; kt < nkt; kt += 2) {
            step(integral_constant<int, 0>{}, kt);
            if (kt + 1 < nkt) step(integral_constant<int, 1>{}, kt + 1);
        }
    }

    // ---- merge the two key halves through LDS, normalise, store --------------------------------
    float l_tot = pair_sum(l_run);
    f32x4* obuf = reinterpret_cast<f32x4*>(smem);  // [4 rw][NDB*4][64 lanes] of f32x4
    if (kg == 1) {
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                f32x4 x = {o[db][4 * q4], o[db][4 * q4 + 1], o[db][4 * q4 + 2], o[db][4 * q4 + 3]};
                obuf[(rw * NDB * 4 + db * 4 + q4) * 64 + lane] = x;
            }
        mlbuf[rw * 128 + lane] = m_run;
        mlbuf[rw * 128 + 64 + lane] = l_tot;
    }
    __syncthreads();
    if (kg != 0) return;

    const float m1 = mlbuf[rw * 128 + lane];
    const float l1 = mlbuf[rw * 128 + 64 + lane];
    const float mf = fmaxf(m_run, m1);
    const float mfs = (mf == -INFINITY) ? 0.f : mf;
    const float a0 = fast_exp2(m_run - mfs), a1 = fast_exp2(m1 - mfs);
    const float lf = l_tot * a0 + l1 * a1;
    const float inv = lf > 0.f ? 1.0f / lf : 0.f;
    const float w0 = a0 * inv, w1 = a1 * inv;

    if (!rvalid) return;
    const int64_t obase = (int64_t)sp * a.out_split_stride + row_off;
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            const f32x4 y = obuf[(rw * NDB * 4 + db * 4 + q4) * 64 + lane];
            f32x4 x;
#pragma unroll
            for (int j = 0; j < 4; ++j) x[j] = o[db][4 * q4 + j] * w0 + y[j] * w1;
            const int d0 = 32 * db + 8 * q4 + 4 * hi;
            if (a.out_f32) {
                *reinterpret_cast<f32x4*>(static_cast<float*>(a.out) + obase + d0) = x;
            } else {
                u32x2 pk = {TR::pack2(x[0], x[1]), TR::pack2(x[2], x[3])};
                *reinterpret_cast<u32x2*>(static_cast<uint16_t*>(a.out) + obase + d0) = pk;
            }
        }
    if (a.lse && hi == 0) {
        const float lse = lf > 0.f ? mf * kLn2 + __logf(lf) : -INFINITY;
        int64_t idx;
        if (a.lse_layout == HYD_LSE_BQH)
            idx = (int64_t)(q_tok0 + rtok) * a.Hq + hq;
        else
            idx = ((int64_t)gi * a.Hq + hq) * a.lse_q_stride + rtok;
        a.lse[(int64_t)sp * a.lse_split_stride + idx] = lse;
    }
}

template <typename T, int D, bool CAUSAL, int ABL = 0>
static int launch_prefix_t(const PrefixArgs& a, int grid, hipStream_t s) {
    constexpr size_t lds = 2 * 256 * (D * 2) + 4 * 128 * sizeof(float);
    auto kern = prefix_attn_kernel<T, D, CAUSAL, ABL>;
    static bool attr_set = false;  // idempotent; value never changes
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds);
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, s, a);
    return (int)hipGetLastError();
}

int launch_prefix(const PrefixArgs& a, int dtype, int D, bool causal, int grid, hipStream_t s) {
#ifdef HYD_ABLATION_BUILD
    if (a.dbg && dtype == HYD_BF16 && D == 128 && !causal) {
        switch (a.dbg) {
#define HYD_ABL(N) case N: return launch_prefix_t<BF16, 128, false, N>(a, grid, s);
            HYD_ABL(1) HYD_ABL(3) HYD_ABL(5) HYD_ABL(7) HYD_ABL(9) HYD_ABL(11) HYD_ABL(13) HYD_ABL(15) HYD_ABL(14)
#undef HYD_ABL
            default: break;
        }
    }
#endif
#define HYD_DISPATCH(TT, DD)                                                    \
    return causal ? launch_prefix_t<TT, DD, true>(a, grid, s) : launch_prefix_t<TT, DD, false>(a, grid, s)
    if (dtype == HYD_F16) {
        if (D == 128) { HYD_DISPATCH(F16, 128); }
        if (D == 64) { HYD_DISPATCH(F16, 64); }
    } else {
        if (D == 128) { HYD_DISPATCH(BF16, 128); }
        if (D == 64) { HYD_DISPATCH(BF16, 64); }
    }
#undef HYD_DISPATCH
    return (int)hipErrorInvalidValue;
}

}  // namespace hyd
