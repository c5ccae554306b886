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
// K/V tiles are staged global -> registers -> LDS with the loads for tile t+1 issued before the
// compute on tile t.  HBM/L2: blocks that share a (group, kv-head) are remapped onto one XCD.
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

__device__ __forceinline__ u32x2 lds_tr16(const char* p) {
    s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p));
    return __builtin_bit_cast(u32x2, t);
}

template <typename T, int D, bool CAUSAL>
__global__ __launch_bounds__(512) void prefix_attn_kernel(const PrefixArgs a) {
    using TR = Traits<T>;
    constexpr int RB = D * 2;            // bytes per K/V row
    constexpr int CPR = D / 8;           // 16-byte chunks per row
    constexpr int NC = D / 16;           // k-chunks of the QK^T contraction
    constexpr int NDB = D / 32;          // 32-wide d blocks of O^T
    constexpr int NLD = (128 * CPR) / 512;  // 16-byte chunks per thread per tensor per tile
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ks = smem;                     // [128 keys][RB], swizzled
    char* Vs = smem + 128 * RB;          // [128 keys][RB], swizzled
    float* mlbuf = reinterpret_cast<float*>(smem + 256 * RB);  // [4][2][64]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rw = wave & 3, kg = wave >> 2;
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

    // ---- per-lane LDS addresses (bytes) ---------------------------------------------------
    const int ksw = D == 128 ? (l31 & 15) : ((l31 >> 1) & 7);
    const int kx = hi ^ ksw;
    int kaddr[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) kaddr[c] = (kg * 64 + l31) * RB + (((2 * c) ^ kx) << 4);
    const int i16 = lane & 15, g16 = lane >> 4;
    const int vsw = D == 128 ? (i16 >> 2) : ((i16 >> 3) & 1);
    int vaddr[NDB];
#pragma unroll
    for (int db = 0; db < NDB; ++db)
        vaddr[db] = (kg * 64 + 4 * hi + (i16 >> 2)) * RB + ((db ^ vsw) << 6) + 32 * (g16 & 1) + 8 * (i16 & 3);

    // ---- staging: thread -> (tile row, 16B chunk) -----------------------------------------
    u32x4 kreg[NLD], vreg[NLD];
    auto gload = [&](int kt) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int c = tid + 512 * i;
            const int row = c / CPR, ch = c % CPR;
            // unpredicated loads: rows past the end re-read the last valid key; their scores are masked to
            // -inf so they contribute exactly 0
            const int key = min(kbeg + kt * 128 + row, kend - 1);
            kreg[i] = *reinterpret_cast<const u32x4*>(k16 + (int64_t)key * a.k_ts + ch * 8);
            vreg[i] = *reinterpret_cast<const u32x4*>(v16 + (int64_t)key * a.v_ts + ch * 8);
        }
    };
    auto sstore = [&]() {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int c = tid + 512 * i;
            const int row = c / CPR, ch = c % CPR;
            *reinterpret_cast<u32x4*>(Ks + row * RB + (kswz<D>(row, ch) << 4)) = kreg[i];
            *reinterpret_cast<u32x4*>(Vs + row * RB + (vswz<D>(row, ch) << 4)) = vreg[i];
        }
    };

    f32x16 o[NDB];
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[db][i] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const float sc = a.scale_log2e;

    if (nkt > 0) {
        gload(0);
        sstore();
    }
    __syncthreads();

    for (int kt = 0; kt < nkt; ++kt) {
        const bool more = kt + 1 < nkt;
        if (more) gload(kt + 1);

        const int kw0 = kbeg + kt * 128 + kg * 64;  // first key of this wave's half tile
        if (kw0 < kend) {
            // ---- S^T = K Q^T -----------------------------------------------------------------
            f32x16 s[2];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
                for (int i = 0; i < 16; ++i) s[kb][i] = 0.f;
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    const u32x4 kf = *reinterpret_cast<const u32x4*>(Ks + kaddr[c] + kb * 32 * RB);
                    s[kb] = TR::mfma32(kf, qf[c], s[kb]);
                }
            }
            // ---- masking (tail of the key range, causal diagonal) ------------------------------
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
            // ---- online softmax (base 2) -------------------------------------------------------
            float tmax = s[0][0];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int i = 0; i < 16; ++i) tmax = fmaxf(tmax, s[kb][i]);
            tmax = pair_max(tmax);
            const float m_new = fmaxf(m_run, tmax * sc);
            const float msafe = (m_new == -INFINITY) ? 0.f : m_new;
            const float alpha = fast_exp2(m_run - msafe);
            float rs = 0.f;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float p = fast_exp2(__builtin_fmaf(s[kb][i], sc, -msafe));
                    s[kb][i] = p;
                    rs += p;
                }
            l_run = l_run * alpha + rs;
            if (__builtin_amdgcn_ballot_w64(m_new > m_run) != 0ull) {
#pragma unroll
                for (int db = 0; db < NDB; ++db) o[db] *= alpha;
            }
            m_run = m_new;
            // ---- P^T fragments: slot ks (16 keys) <- regs [8*(ks&1), +8) of block ks>>1 -----------
            u32x4 pf[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int kb = ks >> 1, b0 = 8 * (ks & 1);
                pf[ks][0] = TR::pack2(s[kb][b0 + 0], s[kb][b0 + 1]);
                pf[ks][1] = TR::pack2(s[kb][b0 + 2], s[kb][b0 + 3]);
                pf[ks][2] = TR::pack2(s[kb][b0 + 4], s[kb][b0 + 5]);
                pf[ks][3] = TR::pack2(s[kb][b0 + 6], s[kb][b0 + 7]);
            }
            // ---- O^T += V^T P^T ----------------------------------------------------------------
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
                for (int db = 0; db < NDB; ++db) {
                    const u32x2 t0 = lds_tr16(Vs + vaddr[db] + (16 * ks) * RB);
                    const u32x2 t1 = lds_tr16(Vs + vaddr[db] + (16 * ks + 8) * RB);
                    const u32x4 vf = {t0[0], t0[1], t1[0], t1[1]};
                    o[db] = TR::mfma32(vf, pf[ks], o[db]);
                }
            }
        }
        __syncthreads();
        if (more) sstore();
        __syncthreads();
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

template <typename T, int D, bool CAUSAL>
static int launch_prefix_t(const PrefixArgs& a, int grid, hipStream_t s) {
    constexpr size_t lds = 256 * (D * 2) + 4 * 128 * sizeof(float);
    auto kern = prefix_attn_kernel<T, D, CAUSAL>;
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
