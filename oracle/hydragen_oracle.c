/*
 * CPU ORACLE (test infrastructure, NOT product code) -- plain-C float32 restatement of the
 * reference's decode hot path at KERNEL level: it follows the reference kernels' arithmetic
 * order (base-2 online softmax in 64-key tiles, probabilities rounded to the q dtype before
 * P.V, split-K + reduce, natural-log LSE), so it lands much closer to the reference's own
 * Triton outputs than a float64 formula does.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this file's
 * shared object.  The product library (hydragen_amd/csrc) never links or calls it.
 *
 * Parity status: PINNED -- tests/test_oracle.py checks it against tests/golden/*.npz, which were
 * produced by the reference's Python run in the authoring container (oracle/make_golden.py).
 *
 * Reference lines followed (relative to /root/reference):
 *   orc_suffix_splitk   hydragen/xformers_stuff.py:267-428 (_fwd_kernel_splitK),
 *                       hydragen/flash.py:76-160 (_splitK_reduce), flash.py:163-281 (driver)
 *   orc_pick_split_k    hydragen/flash.py:37-73
 *   orc_attn_fwd        flash-attn semantics used at hydragen/flash.py:284-351
 *   orc_combine_lse     hydragen/attention.py:21-43
 *   orc_hydragen_decode hydragen/attention.py:246-354 (uniform levels, seq_lens given)
 *
 * Build:  gcc -O2 -fopenmp -shared -fPIC oracle/hydragen_oracle.c -o oracle/libhydragen_oracle.so -lm
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define LOG2E 1.44269504f

/* ---- 16-bit float <-> float32 ------------------------------------------------------------ */
static inline float bf16_to_f32(uint16_t h) {
    uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
static inline uint16_t f32_to_bf16(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40); /* NaN */
    uint32_t lsb = (u >> 16) & 1u;
    u += 0x7fffu + lsb;
    return (uint16_t)(u >> 16);
}
static inline float f16_to_f32(uint16_t h) {
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1fu;
    uint32_t man = h & 0x3ffu;
    uint32_t u;
    if (exp == 0) {
        if (man == 0) {
            u = sign;
        } else { /* subnormal */
            int e = -1;
            do { e++; man <<= 1; } while ((man & 0x400u) == 0);
            man &= 0x3ffu;
            u = sign | ((uint32_t)(127 - 15 - e) << 23) | (man << 13);
        }
    } else if (exp == 31) {
        u = sign | 0x7f800000u | (man << 13);
    } else {
        u = sign | ((exp + 127 - 15) << 23) | (man << 13);
    }
    float f;
    memcpy(&f, &u, 4);
    return f;
}
static inline uint16_t f32_to_f16(float f) {
    uint32_t x;
    memcpy(&x, &f, 4);
    uint32_t sign = (x >> 16) & 0x8000u;
    x &= 0x7fffffffu;
    if (x >= 0x7f800000u) return (uint16_t)(sign | 0x7c00u | (x > 0x7f800000u ? 0x200u : 0));
    {
        float a;
        memcpy(&a, &x, 4);
        if (a >= 65520.0f) return (uint16_t)(sign | 0x7c00u); /* rounds to inf */
    }
    if (x < 0x38800000u) { /* subnormal or zero in f16 */
        float a;
        memcpy(&a, &x, 4);
        /* scale so that the integer part is the f16 subnormal mantissa, round-nearest-even */
        float r = a * 16777216.0f; /* 2^24 */
        float rr = nearbyintf(r);
        return (uint16_t)(sign | (uint16_t)rr);
    }
    uint32_t mant = x & 0x7fffffu;
    uint32_t e = (x >> 23) - 127 + 15;
    uint32_t h = (e << 10) | (mant >> 13);
    uint32_t rem = mant & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) h++;
    return (uint16_t)(sign | h);
}
static inline float ld(const uint16_t* p, int dtype) { return dtype ? bf16_to_f32(*p) : f16_to_f32(*p); }
static inline float rnd(float f, int dtype) {
    return dtype ? bf16_to_f32(f32_to_bf16(f)) : f16_to_f32(f32_to_f16(f));
}
void orc_round_array(const float* in, uint16_t* out, long n, int dtype) {
    for (long i = 0; i < n; i++) out[i] = dtype ? f32_to_bf16(in[i]) : f32_to_f16(in[i]);
}
void orc_widen_array(const uint16_t* in, float* out, long n, int dtype) {
    for (long i = 0; i < n; i++) out[i] = ld(in + i, dtype);
}

/* ---- flash.py:37-73 ----------------------------------------------------------------------- */
int orc_pick_split_k(int B, int H, int M, int BLOCK_M, int Mk, int BLOCK_N, int sm_count) {
    int num_m_blocks = (M + BLOCK_M - 1) / BLOCK_M;
    int num_n_blocks = (Mk + BLOCK_N - 1) / BLOCK_N;
    int bnm = B * H * num_m_blocks;
    if ((double)bnm >= 0.8 * sm_count) return 1;
    int max_splits = sm_count < num_n_blocks ? sm_count : num_n_blocks;
    if (max_splits > 128) max_splits = 128;
    double eff[129];
    double best = 0.0;
    for (int s = 1; s <= max_splits; s++) {
        int eligible = (s == 1) || ((num_n_blocks + s - 1) / s != (num_n_blocks + s - 2) / (s - 1));
        if (!eligible) { eff[s] = 0.0; continue; }
        double n_waves = (double)bnm * s / sm_count;
        eff[s] = n_waves / ceil(n_waves);
        if (eff[s] > best) best = eff[s];
    }
    for (int s = 1; s <= max_splits; s++)
        if (eff[s] >= 0.85 * best) return s;
    return 1;
}

/* ---- suffix pass, kernel-level: xformers_stuff.py:267-428 + flash.py:76-160 ---------------
 * q [B, nq, Hq, D]; k, v [B, Mk, Hkv, D] (row stride Hkv*D); seq_len int32 [B].
 * out [B, nq, Hq, D] fp32 holding values already rounded to the q dtype (flash.py:254,148);
 * lse [B, nq, Hq] natural log (flash.py:159-160).  split_k as chosen by the caller
 * (use orc_pick_split_k with the reference's arguments to mimic flash.py:193-196).
 */
void orc_suffix_splitk(const uint16_t* q, const uint16_t* k, const uint16_t* v, int dtype, int B, int nq,
                       int Hq, int Hkv, int D, int Mk, const int32_t* seq_len, int split_k, float* out,
                       float* lse) {
    const int g = Hq / Hkv;
    const int BLOCK_N = 64;
    const int split_size = (Mk + split_k - 1) / split_k;      /* flash.py:207 */
    const float qk_scale = (1.0f / sqrtf((float)D)) * LOG2E;  /* xformers_stuff.py:349 */
#pragma omp parallel for collapse(2) schedule(dynamic)
    for (int b = 0; b < B; b++) {
        for (int hk = 0; hk < Hkv; hk++) {
            float* acc = (float*)malloc(sizeof(float) * (size_t)split_k * D);
            float* mm = (float*)malloc(sizeof(float) * split_k);
            float* ll = (float*)malloc(sizeof(float) * split_k);
            float p[64];
            for (int iq = 0; iq < nq; iq++)
                for (int gq = 0; gq < g; gq++) {
                    const int h = hk * g + gq;
                    const uint16_t* qr = q + (((size_t)b * nq + iq) * Hq + h) * D;
                    const int kv_len = seq_len ? seq_len[b] : Mk;
                    for (int s = 0; s < split_k; s++) {
                        float m_i = -INFINITY, l_i = 0.f;
                        float* a = acc + (size_t)s * D;
                        for (int d = 0; d < D; d++) a[d] = 0.f;
                        int lo = s * split_size;
                        int hi = (s + 1) * split_size < kv_len ? (s + 1) * split_size : kv_len;
                        for (int n0 = lo; n0 < hi; n0 += BLOCK_N) {
                            float tmax = -INFINITY;
                            int cnt = hi - n0 < BLOCK_N ? hi - n0 : BLOCK_N;
                            for (int j = 0; j < cnt; j++) {
                                const uint16_t* kr = k + (((size_t)b * Mk + n0 + j) * Hkv + hk) * D;
                                float s_ = 0.f;
                                for (int d = 0; d < D; d++) s_ += ld(qr + d, dtype) * ld(kr + d, dtype);
                                p[j] = s_ * qk_scale;
                                if (p[j] > tmax) tmax = p[j];
                            }
                            float m_new = m_i > tmax ? m_i : tmax;
                            float alpha = exp2f(m_i - m_new);
                            float rs = 0.f;
                            for (int j = 0; j < cnt; j++) {
                                p[j] = exp2f(p[j] - m_new);
                                rs += p[j];                     /* l uses the un-rounded p (:388) */
                                p[j] = rnd(p[j], dtype);        /* p.to(Q.dtype)  (:391)          */
                            }
                            l_i = l_i * alpha + rs;
                            m_i = m_new;
                            for (int d = 0; d < D; d++) a[d] *= alpha;
                            for (int j = 0; j < cnt; j++) {
                                const uint16_t* vr = v + (((size_t)b * Mk + n0 + j) * Hkv + hk) * D;
                                for (int d = 0; d < D; d++) a[d] += p[j] * ld(vr + d, dtype);
                            }
                        }
                        mm[s] = m_i;
                        ll[s] = l_i;
                    }
                    /* _splitK_reduce  flash.py:118-160 */
                    float m = mm[0], l_sum = ll[0];
                    float* a0 = acc;
                    for (int s = 1; s < split_k; s++) {
                        float m_k = mm[s], l_k = ll[s];
                        float* ak = acc + (size_t)s * D;
                        float m_new = m > m_k ? m : m_k;
                        if (m_k < m) {
                            float al = exp2f(m_k - m_new);
                            for (int d = 0; d < D; d++) a0[d] += ak[d] * al;
                            l_sum += l_k * al;
                        } else {
                            float al = exp2f(m - m_new);
                            for (int d = 0; d < D; d++) a0[d] = a0[d] * al + ak[d];
                            l_sum = l_sum * al + l_k;
                        }
                        m = m_new;
                    }
                    float* o = out + (((size_t)b * nq + iq) * Hq + h) * D;
                    for (int d = 0; d < D; d++) o[d] = rnd(a0[d] / l_sum, dtype);
                    lse[((size_t)b * nq + iq) * Hq + h] = (m + log2f(l_sum)) / LOG2E;
                }
            free(acc); free(mm); free(ll);
        }
    }
}

/* ---- generic exact attention with query folding (prefix pass / causal suffix) --------------
 * Sequence b uses the KV of group b / (B/sb).  Group gidx's keys are tokens
 * [cu_k[gidx], cu_k[gidx+1]) of the packed k/v when cu_k != NULL, else tokens
 * [gidx*P, gidx*P + (kv_lens ? kv_lens[gidx] : P)).  Token row stride = Hkv*D.
 * causal: query iq sees keys j <= iq + (len - nq)  (bottom-right, flash-attn >= 2.1).
 * out [B,nq,Hq,D] fp32 (NOT rounded), lse [B,nq,Hq] natural log.
 */
void orc_attn_fwd(const uint16_t* q, const uint16_t* k, const uint16_t* v, int dtype, int B, int nq, int Hq,
                  int Hkv, int D, int sb, const int32_t* cu_k, int P, const int32_t* kv_lens, int causal,
                  float* out, float* lse) {
    const int g = Hq / Hkv;
    const float scale = 1.0f / sqrtf((float)D);
    const int per = B / sb;
#pragma omp parallel for collapse(2) schedule(dynamic)
    for (int b = 0; b < B; b++) {
        for (int h = 0; h < Hq; h++) {
            const int gi = b / per, hk = h / g;
            long t0 = cu_k ? cu_k[gi] : (long)gi * P;
            int len = cu_k ? cu_k[gi + 1] - cu_k[gi] : (kv_lens ? kv_lens[gi] : P);
            float* s = (float*)malloc(sizeof(float) * (len > 0 ? len : 1));
            for (int iq = 0; iq < nq; iq++) {
                const uint16_t* qr = q + (((size_t)b * nq + iq) * Hq + h) * D;
                float* o = out + (((size_t)b * nq + iq) * Hq + h) * D;
                int vis = causal ? iq + (len - nq) + 1 : len;
                if (vis > len) vis = len;
                for (int d = 0; d < D; d++) o[d] = 0.f;
                if (vis <= 0) { lse[((size_t)b * nq + iq) * Hq + h] = -INFINITY; continue; }
                float m = -INFINITY;
                for (int j = 0; j < vis; j++) {
                    const uint16_t* kr = k + ((size_t)(t0 + j) * Hkv + hk) * D;
                    float a = 0.f;
                    for (int d = 0; d < D; d++) a += ld(qr + d, dtype) * ld(kr + d, dtype);
                    s[j] = a * scale;
                    if (s[j] > m) m = s[j];
                }
                float l = 0.f;
                for (int j = 0; j < vis; j++) {
                    float pj = expf(s[j] - m);
                    l += pj;
                    const uint16_t* vr = v + ((size_t)(t0 + j) * Hkv + hk) * D;
                    for (int d = 0; d < D; d++) o[d] += pj * ld(vr + d, dtype);
                }
                for (int d = 0; d < D; d++) o[d] /= l;
                lse[((size_t)b * nq + iq) * Hq + h] = m + logf(l);
            }
            free(s);
        }
    }
}

/* ---- attention.py:21-43 ------------------------------------------------------------------- */
void orc_combine_lse(const float* const* outs, const float* const* lses, int n, long rows, int D, float* out) {
#pragma omp parallel for schedule(static)
    for (long r = 0; r < rows; r++) {
        float m = -INFINITY;
        for (int i = 0; i < n; i++) if (lses[i][r] > m) m = lses[i][r];
        float den = 0.f;
        float* o = out + r * D;
        for (int d = 0; d < D; d++) o[d] = 0.f;
        for (int i = 0; i < n; i++) {
            float w = expf(lses[i][r] - m);
            den += w;
            const float* oi = outs[i] + r * D;
            for (int d = 0; d < D; d++) o[d] += oi[d] * w;
        }
        for (int d = 0; d < D; d++) o[d] /= den;
    }
}

/* ---- attention.py:246-354 for one uniform shared level + ragged suffix (the decode case) ----
 * Partials are rounded to the q dtype before the merge, as the reference does (README.md:488-490).
 * scratch: caller-provided float buffer of 2*B*nq*Hq*D + 2*B*nq*Hq elements.
 */
void orc_hydragen_decode(const uint16_t* q, const uint16_t* k, const uint16_t* v, const uint16_t* sk,
                         const uint16_t* sv, int dtype, int B, int nq, int Hq, int Hkv, int D, int sb, int P,
                         int Mk, const int32_t* seq_len, float* scratch, float* out) {
    size_t n = (size_t)B * nq * Hq;
    float* po = scratch;
    float* so = po + n * D;
    float* pl = so + n * D;
    float* sl = pl + n;
    orc_attn_fwd(q, sk, sv, dtype, B, nq, Hq, Hkv, D, sb, NULL, P, NULL, 0, po, pl);
    for (size_t i = 0; i < n * D; i++) po[i] = rnd(po[i], dtype);
    int split_k = 1;
    orc_suffix_splitk(q, k, v, dtype, B, nq, Hq, Hkv, D, Mk, seq_len, split_k, so, sl);
    const float* outs[2] = {po, so};
    const float* lses[2] = {pl, sl};
    orc_combine_lse(outs, lses, 2, (long)n, D, out);
}
