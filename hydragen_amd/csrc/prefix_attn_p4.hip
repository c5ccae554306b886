// Prefix pass, software-pipelined variant "p4": 4 waves per workgroup = ONE wave per SIMD with the
// whole 512-entry register file of its SIMD lane slot (ArchVGPR + AccVGPR).
//
// Why (measured on MI355X, tests/probes/coexec_probe.hip): on one SIMD the MFMA work of one wave and the
// VALU work of ANOTHER wave do not overlap at all -- both go through the single VALU issue port -- while
// independent VALU instructions of the SAME wave do run in the shadow of its MFMAs.  The online-softmax
// VALU work per 32x64 score tile (32 x v_exp_f32 at ~16 cycles + ~70 plain ops) is ~80 % of the tile's
// MFMA time, so it has to hide inside one wave's instruction stream.  Each wave therefore software-
// pipelines across 64-key tiles:
//     stage X:  S^T(t+1) = K(t+1) Q^T   [16 MFMA]   with   exp2 / row-sum / pack of S^T(t)   [VALU]
//     stage Y:  O^T += V^T(t) P^T(t)    [16 MFMA]   with   mask / row-max of S^T(t+1)        [VALU]
// which needs two S^T tiles live (+32 VGPRs) -- more than a 256-register wave can hold next to the
// O^T accumulators (64), the Q fragments (32) and the operand staging registers, hence one wave per SIMD.
//
// Same maths, layouts, swizzles and LDS-DMA staging as prefix_attn.hip (see there); differences:
//   workgroup = 256 threads, 128 folded query rows; wave w owns rows [32w, 32w+32) for ALL keys (no key
//   split, so no cross-wave merge at the end); tile = 64 keys; K runs one tile ahead of V in LDS.
#include <type_traits>

#include "hyd_kernels.h"

namespace hyd {

typedef __attribute__((address_space(3))) s16x4* p4_lds_s16x4_ptr;

__device__ __forceinline__ u32x2 p4_lds_tr16(unsigned lds_byte_addr) {
    s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((p4_lds_s16x4_ptr)(uintptr_t)lds_byte_addr);
    return __builtin_bit_cast(u32x2, t);
}

// One LDS-DMA instruction (see prefix_attn.hip: dma16): hidden from hipcc's waitcnt model on purpose.
__device__ __forceinline__ void p4_dma16(const char* gsrc, const char* lds_dst) {
    const unsigned l = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)lds_dst);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(l)
                 : "memory");
}
__device__ __forceinline__ void p4_dma_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

template <typename T, int D, bool CAUSAL>
__global__ __launch_bounds__(256) void prefix_attn_p4_kernel(const PrefixArgs a) {
    using TR = Traits<T>;
    using std::integral_constant;
    typedef integral_constant<bool, true> true_c;
    typedef integral_constant<bool, false> false_c;
    constexpr int RB = D * 2;            // bytes per K/V row
    constexpr int CPR = D / 8;           // 16-byte chunks per row
    constexpr int NC = D / 16;           // k-chunks of the QK^T contraction
    constexpr int NDB = D / 32;          // 32-wide d blocks of O^T
    constexpr int TK = 64;               // keys per tile
    constexpr int NLD = (TK * CPR) / 256;  // DMA instructions per wave per tensor per tile
    constexpr int RPI = 1024 / RB;       // tile rows per DMA instruction
    constexpr int T_BYTES = TK * RB;     // one K (or V) tile
    constexpr int K_OFF = 0, V_OFF = 2 * T_BYTES;
    extern __shared__ __attribute__((aligned(16))) char smem[];  // K[2] | V[2]
    typedef const __attribute__((address_space(3))) char* lptr_c;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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
        const int rmax = min(Mrows, rb * 128 + 128) - 1;
        kend = min(kend, rmax / a.g + L - nq_eff + 1);
    }
    const int nkeys = kend - kbeg;
    const int nkt = nkeys > 0 ? (nkeys + TK - 1) / TK : 0;

    // ---- this lane's query row ------------------------------------------------------------
    const int r = rb * 128 + wave * 32 + l31;
    const bool rvalid = r < Mrows;
    const int rtok = rvalid ? r / a.g : 0;
    const int hq = hk * a.g + (rvalid ? r % a.g : 0);
    const int64_t row_off = ((int64_t)(q_tok0 + rtok) * a.Hq + hq) * D;
    int row_lim = 0x3fffffff;
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

    // ---- per-lane LDS byte addresses; buffer / row-block offsets are compile-time immediates ----------
    const int ksw = D == 128 ? (l31 & 15) : ((l31 >> 1) & 7);
    const int kx = hi ^ ksw;
    unsigned kaddr[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) kaddr[c] = (unsigned)(uintptr_t)(lptr_c)(smem + K_OFF + l31 * RB + (((2 * c) ^ kx) << 4));
    const int i16 = lane & 15, g16 = lane >> 4;
    const int vsw = D == 128 ? (i16 >> 2) : ((i16 >> 3) & 1);
    unsigned vaddr[NDB];
#pragma unroll
    for (int db = 0; db < NDB; ++db)
        vaddr[db] = (unsigned)(uintptr_t)(lptr_c)(smem + V_OFF + (4 * hi + (i16 >> 2)) * RB + ((db ^ vsw) << 6) +
                                                  32 * (g16 & 1) + 8 * (i16 & 3));

    // ---- staging: global -> LDS DMA, one 1 KiB instruction ("piece") at a time, branch-free -----------
    const int drow = (lane * 16) / RB;
    const int dcp = ((lane * 16) % RB) >> 4;
    unsigned koff[NLD], voff[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int row = (wave * NLD + i) * RPI + drow;  // row of the 64-row tile
        const int kch = D == 128 ? (dcp ^ (row & 15)) : (dcp ^ ((row >> 1) & 7));
        const int vs_ = D == 128 ? (row & 3) : ((row >> 1) & 1);
        const int vch = (((dcp >> 2) ^ vs_) << 2) | (dcp & 3);
        koff[i] = (unsigned)(((int64_t)row * a.k_ts + kch * 8) * 2);
        voff[i] = (unsigned)(((int64_t)row * a.v_ts + vch * 8) * 2);
    }
    const char* kbase = reinterpret_cast<const char*>(k16) + (int64_t)kbeg * a.k_ts * 2;
    const char* vbase = reinterpret_cast<const char*>(v16) + (int64_t)kbeg * a.v_ts * 2;
    const int krs = (int)(a.k_ts * 2), vrs = (int)(a.v_ts * 2);
    int dma_tk = 0, dma_tv = 0;  // always valid tile indices (clamped to the last tile by the callers)
    auto dma_piece_c = [&](auto J_C) {
        constexpr int j = decltype(J_C)::value;
        if constexpr (j < NLD) {
            constexpr int i = j;
            const int tk = dma_tk;
            char* Kd = smem + K_OFF + (tk & 1) * T_BYTES + wave * NLD * 1024;
            const char* tb = kbase + (int64_t)tk * TK * a.k_ts * 2;
            const int over = max(0, tk * TK + (wave * NLD + i) * RPI + drow - (nkeys - 1));
            p4_dma16(tb + (int)(koff[i] - (unsigned)(over * krs)), Kd + i * 1024);
        } else if constexpr (j < 2 * NLD) {
            constexpr int i = j - NLD;
            const int tv = dma_tv;
            char* Vd = smem + V_OFF + (tv & 1) * T_BYTES + wave * NLD * 1024;
            const char* tb = vbase + (int64_t)tv * TK * a.v_ts * 2;
            const int over = max(0, tv * TK + (wave * NLD + i) * RPI + drow - (nkeys - 1));
            p4_dma16(tb + (int)(voff[i] - (unsigned)(over * vrs)), Vd + i * 1024);
        }
    };
    auto dma_piece = [&](int j) {
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

    f32x16 o[NDB];
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[db][i] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const float sc = a.scale_log2e;
    f32x16 sA[2], sB[2];
    u32x4 pf[4];
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float msafe_cur = 0.f, alpha_cur = 1.f;

    // ---- stage X: S^T(next) = K Q^T  with  exp2 / row-sum / pack of S^T(cur) in the MFMA shadows ---------
    auto stage_x = [&](auto KOFF_C, auto P0_C, auto DOSM_C, f32x16(&sn)[2], f32x16(&scur)[2]) {
        constexpr int KOFF = decltype(KOFF_C)::value;
        constexpr int P0 = decltype(P0_C)::value;
        constexpr bool DO_SM = decltype(DOSM_C)::value;
        constexpr int PD = 4;
        u32x4 kfr[PD];
        auto ldk = [&](int i) -> u32x4 {  // step i -> (c = i / 2, kb = i % 2): alternate the two accumulators
            return *reinterpret_cast<const __attribute__((address_space(3))) u32x4*>(
                (uintptr_t)(kaddr[i / 2] + KOFF + (i % 2) * 32 * RB));
        };
#pragma unroll
        for (int i = 0; i < PD; ++i) kfr[i] = ldk(i);
        f32x2 rsum = {0.f, 0.f};
        constexpr int NSTEP = 2 * NC;
        constexpr int EPS = 32 / NSTEP;
#pragma unroll
        for (int i = 0; i < NSTEP; ++i) {
            const int kb = i % 2, c = i / 2;
            sn[kb] = TR::mfma32(kfr[i % PD], qf[c], c == 0 ? zero16 : sn[kb]);
            if (i + PD < NSTEP) kfr[i % PD] = ldk(i + PD);
            if (DO_SM) {
#pragma unroll
                for (int e = 0; e < EPS; e += 2) {
                    const int idx = i * EPS + e;
                    const int b = idx / 16, rr = idx % 16;
                    f32x2 v = {scur[b][rr], scur[b][rr + 1]};
                    v = v * sc - msafe_cur;
                    v[0] = fast_exp2(v[0]);
                    v[1] = fast_exp2(v[1]);
                    rsum += v;
                    pf[2 * b + (rr >= 8)][(rr % 8) / 2] = TR::pack2(v[0], v[1]);
                }
            }
            constexpr int EVERY = NSTEP / NLD;
            if (P0 >= 0 && (i % EVERY) == EVERY - 1) dma_piece(P0 + i / EVERY);
        }
        if (DO_SM) l_run = l_run * alpha_cur + (rsum[0] + rsum[1]);
    };
    // ---- stage Y: O^T += V^T P^T  with  mask / row-max of S^T(next) in the MFMA shadows --------------------
    auto stage_y = [&](auto VOFF_C, auto P0_C, auto DOMAX_C, auto MASK_C, f32x16(&sn)[2], int kw0n) -> float {
        constexpr int VOFF = decltype(VOFF_C)::value;
        constexpr int P0 = decltype(P0_C)::value;
        constexpr bool DO_MAX = decltype(DOMAX_C)::value;
        constexpr bool MASK = decltype(MASK_C)::value;
        int lr = 0;
        if (DO_MAX && MASK) {
            int lim = kend - 1;
            if (CAUSAL) lim = min(lim, row_lim);
            lr = lim - kw0n - 4 * hi;
        }
        float tmax = -INFINITY;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
            for (int db = 0; db < NDB; ++db) {
                const u32x2 t0 = p4_lds_tr16(vaddr[db] + VOFF + (16 * ks) * RB);
                const u32x2 t1 = p4_lds_tr16(vaddr[db] + VOFF + (16 * ks + 8) * RB);
                const u32x4 vf = {t0[0], t0[1], t1[0], t1[1]};
                o[db] = TR::mfma32(vf, pf[ks], o[db]);
                if (DO_MAX) {
                    constexpr int EPS = 32 / (4 * NDB);
#pragma unroll
                    for (int e = 0; e < EPS; ++e) {
                        const int idx = (ks * NDB + db) * EPS + e;
                        const int b = idx / 16, rr = idx % 16;
                        if (MASK && (b * 32 + 8 * (rr >> 2) + (rr & 3) > lr)) sn[b][rr] = -INFINITY;
                        tmax = fmaxf(tmax, sn[b][rr]);
                    }
                }
            }
            if (P0 >= 0) {
                if (NLD == 4) dma_piece(P0 + ks);
                else if (NLD == 2 && ks % 2 == 1) dma_piece(P0 + ks / 2);
                else if (NLD == 1 && ks == 3) dma_piece(P0);
            }
        }
        return tmax;
    };
    auto finish_max = [&](float tmax) {
        tmax = pair_max(tmax);
        const float m_new = fmaxf(m_run, tmax * sc);
        msafe_cur = (m_new == -INFINITY) ? 0.f : m_new;
        alpha_cur = fast_exp2(m_run - msafe_cur);
        if (__builtin_amdgcn_ballot_w64(m_new > m_run) != 0ull) {
#pragma unroll
            for (int db = 0; db < NDB; ++db) o[db] *= alpha_cur;
        }
        m_run = m_new;
    };
    auto mask_max = [&](f32x16(&sn)[2], int kw0n) -> float {
        int lim = kend - 1;
        if (CAUSAL) lim = min(lim, row_lim);
        const int lr = lim - kw0n - 4 * hi;
        float tmax = -INFINITY;
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) {
                if (b * 32 + 8 * (rr >> 2) + (rr & 3) > lr) sn[b][rr] = -INFINITY;
                tmax = fmaxf(tmax, sn[b][rr]);
            }
        return tmax;
    };

    // ---- pipeline ---------------------------------------------------------------------------------
    if (nkt > 0) {
        dma_tk = 0;
        dma_tv = 0;
#pragma unroll
        for (int j = 0; j < 2 * NLD; ++j) dma_piece(j);  // K(0), V(0)
        dma_tk = min(1, nkt - 1);
#pragma unroll
        for (int j = 0; j < NLD; ++j) dma_piece(j);      // K(1): K runs one tile ahead of V
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) asm volatile("" ::"v"(qf[c]));  // wait for Q here, not inside the loop
    p4_dma_wait_all();
    __syncthreads();

    if (nkt > 0) {
        stage_x(integral_constant<int, 0>{}, integral_constant<int, -1>{}, false_c{}, sA, sB);  // S^T(0)
        finish_max(mask_max(sA, kbeg));
        __syncthreads();  // every wave is done reading K[0] (tile 0) before step 0 re-fills it with tile 2
        // step t: X = QK(t+1) | softmax(t);  Y = PV(t) | max(t+1).  Unrolled by two: buffer offsets are
        // immediates and the two S^T buffers swap roles statically.
        auto step = [&](auto PAR_C, auto MASK_C, int kt) {
            constexpr int par = decltype(PAR_C)::value;
            f32x16(&scur)[2] = par ? sB : sA;
            f32x16(&snxt)[2] = par ? sA : sB;
            // register-class pins: accumulators and Q fragments are MFMA-only operands -> AccVGPRs; the score
            // tiles are VALU operands -> ArchVGPRs (otherwise hipcc parks them in AccVGPRs and pays a
            // v_accvgpr_read/write per element, eating the MFMA shadows the softmax is supposed to use)
#pragma unroll
            for (int db = 0; db < NDB; ++db) asm volatile("" : "+a"(o[db]));
#pragma unroll
            for (int c = 0; c < NC; ++c) asm volatile("" : "+a"(qf[c]));
            asm volatile("" : "+v"(scur[0]), "+v"(scur[1]));
            dma_tk = min(kt + 2, nkt - 1);  // -> K[(kt+2)&1] = K[par]: last read by QK(kt), before the previous barrier
            dma_tv = min(kt + 1, nkt - 1);  // -> V[par^1]: last read by PV(kt-1)
            stage_x(integral_constant<int, (par ^ 1) * T_BYTES>{}, integral_constant<int, 0>{}, true_c{}, snxt, scur);
            const float tmax = stage_y(integral_constant<int, par * T_BYTES>{}, integral_constant<int, NLD>{}, true_c{},
                                       MASK_C, snxt, kbeg + (kt + 1) * TK);
            finish_max(tmax);
            p4_dma_wait_all();
            __syncthreads();
        };
        typedef integral_constant<bool, CAUSAL> mask_dflt;
        int kt = 0;
        for (; kt + 2 < nkt; ++kt) {
            if (kt & 1) step(integral_constant<int, 1>{}, mask_dflt{}, kt);
            else step(integral_constant<int, 0>{}, mask_dflt{}, kt);
        }
        if (kt + 1 < nkt) {  // successor is the last tile: may be partial -> masked
            if (kt & 1) step(integral_constant<int, 1>{}, true_c{}, kt);
            else step(integral_constant<int, 0>{}, true_c{}, kt);
            ++kt;
        }
        auto tail = [&](auto PAR_C) {  // last tile: softmax + PV only
            constexpr int par = decltype(PAR_C)::value;
            f32x16(&scur)[2] = par ? sB : sA;
            float rs = 0.f;
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int rr = 0; rr < 16; rr += 2) {
                    f32x2 v = {scur[b][rr], scur[b][rr + 1]};
                    v = v * sc - msafe_cur;
                    v[0] = fast_exp2(v[0]);
                    v[1] = fast_exp2(v[1]);
                    rs += v[0] + v[1];
                    pf[2 * b + (rr >= 8)][(rr % 8) / 2] = TR::pack2(v[0], v[1]);
                }
            l_run = l_run * alpha_cur + rs;
            (void)stage_y(integral_constant<int, par * T_BYTES>{}, integral_constant<int, -1>{}, false_c{}, false_c{}, scur,
                          0);
        };
        if (kt & 1) tail(integral_constant<int, 1>{});
        else tail(integral_constant<int, 0>{});
    }

    // ---- normalise, store ---------------------------------------------------------------------------
    const float lf = pair_sum(l_run);
    const float inv = lf > 0.f ? 1.0f / lf : 0.f;
    if (!rvalid) return;
    const int64_t obase = (int64_t)sp * a.out_split_stride + row_off;
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            f32x4 x;
#pragma unroll
            for (int j = 0; j < 4; ++j) x[j] = o[db][4 * q4 + j] * inv;
            const int d0 = 32 * db + 8 * q4 + 4 * hi;
            if (a.out_f32) {
                *reinterpret_cast<f32x4*>(static_cast<float*>(a.out) + obase + d0) = x;
            } else {
                u32x2 pk = {TR::pack2(x[0], x[1]), TR::pack2(x[2], x[3])};
                *reinterpret_cast<u32x2*>(static_cast<uint16_t*>(a.out) + obase + d0) = pk;
            }
        }
    if (a.lse && hi == 0) {
        const float lse = lf > 0.f ? m_run * kLn2 + __logf(lf) : -INFINITY;
        int64_t idx;
        if (a.lse_layout == HYD_LSE_BQH)
            idx = (int64_t)(q_tok0 + rtok) * a.Hq + hq;
        else
            idx = ((int64_t)gi * a.Hq + hq) * a.lse_q_stride + rtok;
        a.lse[(int64_t)sp * a.lse_split_stride + idx] = lse;
    }
}

template <typename T, int D, bool CAUSAL>
static int launch_p4_t(const PrefixArgs& a, int grid, hipStream_t s) {
    constexpr size_t lds = 4 * 64 * (D * 2);
    hipLaunchKernelGGL((prefix_attn_p4_kernel<T, D, CAUSAL>), dim3(grid), dim3(256), lds, s, a);
    return (int)hipGetLastError();
}

int launch_prefix_p4(const PrefixArgs& a, int dtype, int D, bool causal, int grid, hipStream_t s) {
#define HYD_DISPATCH(TT, DD) return causal ? launch_p4_t<TT, DD, true>(a, grid, s) : launch_p4_t<TT, DD, false>(a, grid, s)
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
