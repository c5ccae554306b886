// C ABI of libhydragen_hip.so (see include/hydragen_hip.h): argument validation, shapes-only launch
// planning, workspace carving.  No allocation, no synchronisation, no device reads on the host.
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cstdlib>

#include "hyd_kernels.h"

using namespace hyd;

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

constexpr int kNumCU = 256;  // MI355X
constexpr int kMaxSplits = 32;

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Development switches exist only in HYD_ABLATION_BUILD libraries (A/B measurements on hardware); the product library
// reads no environment variable and keeps no mutable state.
inline int dev_switch(const char* name) {
#ifdef HYD_ABLATION_BUILD
    const char* e = getenv(name);
    return e ? atoi(e) : 0;
#else
    (void)name;
    return 0;
#endif
}

struct PrefixPlan {
    int g, per, row_blocks, nsplit, split_len, grid, qpg, wg_rows;
};

// softmax scale in base-2 exponent units: the caller's scale when it gives one, D^-0.5 otherwise
float scale_log2e_of(float softmax_scale, int D) {
    return (softmax_scale > 0.f ? softmax_scale : 1.0f / sqrtf((float)D)) * kLog2e;
}

int check_scale(float s) {
    if (!(s >= 0.f) || s > 1.0e4f) return fail(HYD_ERR_BAD_ARG, "softmax_scale %g (0 = head_dim^-0.5, or a positive finite scale)", (double)s);
    return HYD_OK;
}

int check_common(int dtype, int B, int nq, int Hq, int Hkv, int D) {
    if (dtype != HYD_F16 && dtype != HYD_BF16) return fail(HYD_ERR_UNSUPPORTED, "dtype %d: only f16/bf16", dtype);
    if (D != 64 && D != 128 && D != 256) return fail(HYD_ERR_UNSUPPORTED, "head_dim %d: only 64, 128 and 256 are implemented", D);
    if (B <= 0 || nq <= 0 || Hq <= 0 || Hkv <= 0) return fail(HYD_ERR_BAD_ARG, "non-positive size B=%d nq=%d Hq=%d Hkv=%d", B, nq, Hq, Hkv);
    if (Hq % Hkv != 0) return fail(HYD_ERR_BAD_ARG, "qheads %d not divisible by kvheads %d", Hq, Hkv);
    return HYD_OK;
}

int plan_prefix(const hyd_prefix_params* p, PrefixPlan* pl, int max_splits = kMaxSplits) {
    if (!p) return fail(HYD_ERR_BAD_ARG, "null params");
    int rc = check_common(p->dtype, p->B, p->nq, p->Hq, p->Hkv, p->D);
    if (!rc) rc = check_scale(p->softmax_scale);
    if (rc) return rc;
    if (p->sb <= 0) return fail(HYD_ERR_BAD_ARG, "sb %d", p->sb);
    if (p->kv_len < 0) return fail(HYD_ERR_BAD_ARG, "kv_len %d", p->kv_len);
    pl->g = p->Hq / p->Hkv;
    int qtok_per_group;
    if (p->cu_seqlens_q) {
        if (p->nq != 1) return fail(HYD_ERR_BAD_ARG, "cu_seqlens_q requires nq == 1 (packed query tokens)");
        if (p->max_q_len <= 0) return fail(HYD_ERR_BAD_ARG, "cu_seqlens_q requires max_q_len > 0");
        qtok_per_group = p->max_q_len;
        pl->per = 1;
    } else {
        if (p->B % p->sb != 0) return fail(HYD_ERR_BAD_ARG, "batch %d not divisible by shared batch %d", p->B, p->sb);
        pl->per = p->B / p->sb;
        qtok_per_group = pl->per * p->nq;
    }
    pl->qpg = qtok_per_group;
    const int64_t mrows = (int64_t)qtok_per_group * pl->g;
    pl->row_blocks = (int)((mrows + 127) / 128);
    pl->wg_rows = 128;
    {
        // 256-row workgroups (D = 128) when 128-row ones would need more than one round of the chip anyway: half the
        // K/V staging per flop and no cross-half merge.
        const int force_rows = dev_switch("HYD_PREFIX_ROWS");  // 0 in product builds
        const int64_t units128 = (int64_t)p->sb * p->Hkv * pl->row_blocks;
        const bool can = p->D != 256;  // D = 256: one query block per wave, 128-row workgroups only
        if (can && (force_rows == 256 || (force_rows == 0 && units128 > kNumCU))) {
            pl->wg_rows = 256;
            pl->row_blocks = (int)((mrows + 255) / 256);
        }
    }
    const int64_t units = (int64_t)p->sb * p->Hkv * pl->row_blocks;
    int ns = p->num_splits;
    if (ns <= 0) {
        // shapes-only heuristic: the prefix kernel runs one workgroup per CU (128 KB of LDS), so aim for ONE round of
        // at most kNumCU workgroups (a second, partly filled round costs a whole extra pass plus its fixed ~12 us);
        // keep >= 256 keys per split
        ns = 1;
        if (units <= kNumCU / 2) {
            const int max_by_len = p->kv_len / 256 > 0 ? p->kv_len / 256 : 1;
            const int want = (int)(kNumCU / units);
            ns = want < max_by_len ? want : max_by_len;
            // ... but every split is one more fp32 slice of ALL rows written here and read back by the consumer.  With the
            // workgroups in one round the pass takes about fixed + (kv_len / ns) c + ns s1: c = 13.6 ns per key of a 128-row unit
            // (C2's key loop), s1 = what a slice costs, calibrated at twice its write + read time at 5 TB/s on the sweeps of
            // profiles/r05_split_plan_sweep.txt (the consumer's epilogue pays for it in latency as well).  The minimum is at
            // ns = sqrt(kv_len c / s1): C3 16, a TP = 8 shard of C2 (4/4 heads) 4 instead of 8 (whole call 48.4 -> 43.4 us),
            // B = 64, 32/8 heads, P = 4096 8 instead of 16 (37.1 -> 30.9), B = 256, 32/32, P = 512 unsplit (67.2 -> 62.8).
            // c is per key of a 128-row unit at D = 128; the loop's MFMAs per key (D / 16 for QK^T + D / 16 for PV) and the staged bytes
            // scale with D, and so does a slice (s1): the optimum is the same split count at every head dim.
            const double c_us = 0.0136 * (double)p->D / 128.0;
            const double s1_us = (double)p->B * p->nq * p->Hq * p->D * 16.0 / 5.0e6;
            const double best = sqrt((double)p->kv_len * c_us / s1_us);
            const int by_cost = best < 1.5 ? 1 : (int)(best + 0.5);
            if (by_cost < ns) ns = by_cost;
        }
    }
    if (p->cu_seqlens_q) ns = 1;  // merged LSE re-layout needs uniform query counts
    if (max_splits > kMaxSplits) max_splits = kMaxSplits;
    if (ns > max_splits) ns = max_splits;
    if (ns < 1) ns = 1;
    int split_len = (int)align_up((size_t)((p->kv_len + ns - 1) / ns), 128);
    if (split_len == 0) split_len = 128;
    ns = p->kv_len > 0 ? (p->kv_len + split_len - 1) / split_len : 1;
    // The prefix kernel addresses one split's keys through a 32-bit buffer resource + scalar offset (it issues blocks up
    // to 4 x 32 rows past the end, which must not wrap): cut longer spans into more splits, or refuse.
    {
        const int64_t ts = p->k_tok_stride > p->v_tok_stride ? p->k_tok_stride : p->v_tok_stride;
        const int64_t max_rows = ts > 0 ? (((int64_t)1 << 31) / (ts * 2)) - 512 : (int64_t)1 << 30;
        if (max_rows < 128) return fail(HYD_ERR_UNSUPPORTED, "token stride %lld elements is too large for 32-bit key offsets", (long long)ts);
        if (split_len > max_rows) {
            split_len = (int)(max_rows / 128 * 128);
            ns = (p->kv_len + split_len - 1) / split_len;
            if (ns > kMaxSplits || p->cu_seqlens_q)
                return fail(HYD_ERR_UNSUPPORTED, "%d keys at a token stride of %lld elements exceed the 2 GiB per split the prefix pass addresses",
                            p->kv_len, (long long)ts);
        }
    }
    pl->nsplit = ns;
    pl->split_len = split_len;
    const int64_t grid = units * ns;
    if (grid <= 0 || grid > 0x7fffffff) return fail(HYD_ERR_UNSUPPORTED, "grid too large");
    pl->grid = (int)grid;
    return HYD_OK;
}

// Split-KV slices are fp32 everywhere.  (16-bit slices on the fused decode path were measured: C3 80.5 -> 79.3 us,
// C5 145.7 -> 140.9 us flushed, but every slice then carries its own rounding and the bf16 mean relative difference
// of C3 / C5 / deep hierarchies rose from 0.8 % to 1.1-1.2 %, above the bound the parity tests state.)
size_t prefix_ws_bytes(const hyd_prefix_params* p, const PrefixPlan& pl, size_t esz = sizeof(float)) {
    if (pl.nsplit <= 1) return 0;
    const size_t rows = (size_t)p->B * p->nq * p->Hq;
    return (size_t)pl.nsplit * (align_up(rows * p->D * esz, 256) + align_up(rows * sizeof(float), 256));
}

void fill_prefix_args(const hyd_prefix_params* p, const PrefixPlan& pl, PrefixArgs* a) {
    memset(a, 0, sizeof(*a));
    a->q = p->q;
    a->k = p->k;
    a->v = p->v;
    a->cu_k = p->cu_seqlens_k;
    a->cu_q = p->cu_seqlens_q;
    a->k_gs = p->k_group_stride;
    a->k_ts = p->k_tok_stride;
    a->k_hs = p->k_head_stride;
    a->v_gs = p->v_group_stride;
    a->v_ts = p->v_tok_stride;
    a->v_hs = p->v_head_stride;
    a->B = p->B;
    a->nq = p->nq;
    a->Hq = p->Hq;
    a->Hkv = p->Hkv;
    a->g = pl.g;
    a->sb = p->sb;
    a->per = pl.per;
    a->kv_len = p->kv_len;
    a->row_blocks = pl.row_blocks;
    a->wg_rows = pl.wg_rows;
    // Two waves per SIMD (8-wave workgroups of 32-row waves) wherever that unit is built: D = 128, one workgroup per unit
    // (persistent launches keep the 4-wave unit: launch_prefix_w64_t).  Same rows per workgroup, rings and results; fewer
    // cycles (the SIMD issues one wave's VALU / LDS / DMA instructions beside the other's MFMAs), the same energy: 10 %
    // faster where the pass is short (P = 128: 9.8 against 11.0 us), equal where the chip runs at its power limit.
    {
        const int force_waves = dev_switch("HYD_PREFIX_WAVES");  // 0 in product builds
        a->waves = (p->D == 128 && force_waves != 4) ? 8 : 4;
    }
    a->nsplit = pl.nsplit;
    a->split_len = pl.split_len;
    a->vgrid = pl.grid;
    a->lse_q_stride = pl.qpg;
    a->scale_log2e = scale_log2e_of(p->softmax_scale, p->D);
    a->div_row_blocks = make_fastdiv((uint32_t)pl.row_blocks);
    a->div_nsplit = make_fastdiv((uint32_t)pl.nsplit);
    a->div_hkv = make_fastdiv((uint32_t)p->Hkv);
    a->div_g = make_fastdiv((uint32_t)pl.g);
    a->dbg = dev_switch("HYD_DBG");  // timing-ablation kernel variants; always 0 in product builds
}

// Run the prefix pass.  With nsplit > 1 the kernel writes fp32 slices + BQH LSEs into `ws`; if
// `merge` they are then combined into p->out / p->lse, otherwise the caller consumes the slices.
int launch_prefix_any(const PrefixArgs& a, int dtype, int D, bool causal, int grid, int max_wgs, hipStream_t s) {
    // max_wgs > 0: at most that many PERSISTENT workgroups walk the units (hyd_decode_params.shared_max_workgroups: the
    // rest of the chip stays free for work on another stream); 0 = one workgroup per unit
#ifdef HYD_ABLATION_BUILD
    if (const int np = dev_switch("HYD_PREFIX_PERSIST")) max_wgs = np;
#endif
    if (max_wgs > 0 && grid > max_wgs) grid = max_wgs;
    return launch_prefix_w64(a, dtype, D, causal, grid, s);
}

int run_prefix(const hyd_prefix_params* p, const PrefixPlan& pl, bool merge, hipStream_t s, int max_wgs = 0, bool out_f32 = false) {
    PrefixArgs a;
    fill_prefix_args(p, pl, &a);
    const size_t rows = (size_t)p->B * p->nq * p->Hq;
    if (pl.nsplit == 1) {
        a.out = p->out;
        a.lse = p->lse;
        a.out_f32 = out_f32 ? 1 : 0;  // only the decode entry asks for an fp32 partial (hyd_decode_params.f32_partials)
        a.lse_layout = p->lse_layout;
        int rc = launch_prefix_any(a, p->dtype, p->D, p->causal != 0, pl.grid, max_wgs, s);
        return rc ? fail(HYD_ERR_LAUNCH, "prefix kernel launch failed: hip error %d", rc) : HYD_OK;
    }
    const size_t esz = sizeof(float);
    const size_t need = prefix_ws_bytes(p, pl, esz);
    if (!p->workspace || p->workspace_bytes < need)
        return fail(HYD_ERR_WORKSPACE, "prefix pass needs %zu workspace bytes, got %zu", need, p->workspace_bytes);
    const size_t o_bytes = align_up(rows * p->D * esz, 256);
    const size_t l_bytes = align_up(rows * sizeof(float), 256);
    char* ws = static_cast<char*>(p->workspace);
    float* wo = reinterpret_cast<float*>(ws);
    float* wl = reinterpret_cast<float*>(ws + (size_t)pl.nsplit * o_bytes);
    a.out = wo;
    a.lse = wl;
    a.out_f32 = 1;
    a.lse_layout = HYD_LSE_BQH;
    a.out_split_stride = (int64_t)(o_bytes / esz);
    a.lse_split_stride = (int64_t)(l_bytes / sizeof(float));
    int rc = launch_prefix_any(a, p->dtype, p->D, p->causal != 0, pl.grid, max_wgs, s);
    if (rc) return fail(HYD_ERR_LAUNCH, "prefix kernel launch failed: hip error %d", rc);
    if (!merge) return HYD_OK;
    CombineArgs c;
    memset(&c, 0, sizeof(c));
    for (int i = 0; i < pl.nsplit; ++i) {
        c.outs[i] = wo + (size_t)i * a.out_split_stride;
        c.lses[i] = wl + (size_t)i * a.lse_split_stride;
    }
    c.n = pl.nsplit;
    c.rows = (int64_t)rows;
    c.D = p->D;
    c.dtype_in = HYD_F32;
    c.dtype_out = p->dtype;
    c.out = p->out;
    c.out_lse = p->lse;
    c.lse_layout = p->lse_layout;
    c.Hq = p->Hq;
    c.qpg = pl.qpg;
    rc = launch_combine(c, s);
    return rc ? fail(HYD_ERR_LAUNCH, "combine kernel launch failed: hip error %d", rc) : HYD_OK;
}

int check_ptr_align(const void* p, const char* name) {
    if (!p) return fail(HYD_ERR_BAD_ARG, "%s is null", name);
    if ((reinterpret_cast<uintptr_t>(p) & 15u) != 0) return fail(HYD_ERR_BAD_ARG, "%s must be 16-byte aligned", name);
    return HYD_OK;
}
int check_stride8(int64_t s, const char* name) {
    if (s % 8 != 0) return fail(HYD_ERR_BAD_ARG, "%s (%lld) must be a multiple of 8 elements", name, (long long)s);
    return HYD_OK;
}

int check_suffix(const hyd_suffix_params* p, bool need_kv) {
    if (!p) return fail(HYD_ERR_BAD_ARG, "null params");
    int rc = check_common(p->dtype, p->B, p->nq, p->Hq, p->Hkv, p->D);
    if (!rc) rc = check_scale(p->softmax_scale);
    if (rc) return rc;
    if (p->kv_len < 0) return fail(HYD_ERR_BAD_ARG, "kv_len %d", p->kv_len);
    if ((rc = check_ptr_align(p->q, "q"))) return rc;
    if ((rc = check_ptr_align(p->out, "out"))) return rc;
    if (need_kv && p->kv_len > 0) {
        if ((rc = check_ptr_align(p->k, "k"))) return rc;
        if ((rc = check_ptr_align(p->v, "v"))) return rc;
        if ((rc = check_stride8(p->k_batch_stride, "k_batch_stride")) || (rc = check_stride8(p->k_tok_stride, "k_tok_stride")) ||
            (rc = check_stride8(p->k_head_stride, "k_head_stride")) || (rc = check_stride8(p->v_batch_stride, "v_batch_stride")) ||
            (rc = check_stride8(p->v_tok_stride, "v_tok_stride")) || (rc = check_stride8(p->v_head_stride, "v_head_stride")))
            return rc;
    }
    return HYD_OK;
}

void fill_suffix_args(const hyd_suffix_params* p, SuffixArgs* ap) {
    SuffixArgs& a = *ap;
    memset(&a, 0, sizeof(a));
    a.q = p->q;
    a.k = p->k;
    a.v = p->v;
    a.out = p->out;
    a.lse = p->lse;
    a.sl32 = p->seq_lens_i32;
    a.sl64 = p->seq_lens_i64;
    a.order = p->seq_order;
    a.k_bs = p->k_batch_stride;
    a.k_ts = p->k_tok_stride;
    a.k_hs = p->k_head_stride;
    a.v_bs = p->v_batch_stride;
    a.v_ts = p->v_tok_stride;
    a.v_hs = p->v_head_stride;
    a.B = p->B;
    a.nq = p->nq;
    a.Hq = p->Hq;
    a.Hkv = p->Hkv;
    a.g = p->Hq / p->Hkv;
    a.kv_len = p->kv_len;
    a.rows = p->nq * a.g;
    a.units = p->B * p->Hkv;
    a.scale_log2e = scale_log2e_of(p->softmax_scale, p->D);
}

int run_suffix(const hyd_suffix_params* p, const hyd_partial* parts, int n_parts, hipStream_t s) {
    SuffixArgs a;
    fill_suffix_args(p, &a);
    const size_t rows = (size_t)p->B * p->nq * p->Hq;
    int n = 0;
    for (int i = 0; i < n_parts; ++i) {
        if (!parts[i].out || !parts[i].lse || parts[i].count <= 0)
            return fail(HYD_ERR_BAD_ARG, "partial %d: null pointer or non-positive count", i);
        for (int j = 0; j < parts[i].count; ++j) {
            if (n >= kMaxCombine) return fail(HYD_ERR_UNSUPPORTED, "more than %d partials", kMaxCombine);
            const size_t esz = parts[i].is_f32 ? 4 : 2;
            // stacked slices are padded to 256 bytes exactly as hyd_prefix_attn_fwd lays them out
            const size_t ostride = parts[i].count > 1 ? align_up(rows * p->D * esz, 256) : rows * p->D * esz;
            const size_t lstride = parts[i].count > 1 ? align_up(rows * 4, 256) : rows * 4;
            a.partials[n].out = static_cast<const char*>(parts[i].out) + (size_t)j * ostride;
            a.partials[n].lse = reinterpret_cast<const float*>(reinterpret_cast<const char*>(parts[i].lse) + (size_t)j * lstride);
            a.partials[n].is_f32 = parts[i].is_f32;
            ++n;
        }
    }
    a.n_partials = n;
    a.n_pre = 0;
    while (a.n_pre < 2 && a.n_pre < n && !a.partials[a.n_pre].is_f32) ++a.n_pre;
    if ((int64_t)p->kv_len * p->k_tok_stride * 2 >= (1ll << 31) || (int64_t)p->kv_len * p->v_tok_stride * 2 >= (1ll << 31))
        return fail(HYD_ERR_UNSUPPORTED, "unique K/V of one sequence spans >= 2 GiB (32-bit in-sequence offsets)");
    if (p->Hkv > 4 * 65535 || a.rows > 8 * 65535) return fail(HYD_ERR_UNSUPPORTED, "too many kv heads / query rows for the suffix grid");
    int rc = launch_suffix(a, p->dtype, p->D, s);
    return rc ? fail(HYD_ERR_LAUNCH, "suffix kernel launch failed: hip error %d", rc) : HYD_OK;
}

void level_to_prefix(const hyd_decode_params* p, int i, hyd_prefix_params* pp) {
    const hyd_suffix_params& s = p->suffix;
    const hyd_level& lv = p->levels[i];
    memset(pp, 0, sizeof(*pp));
    pp->q = s.q;
    pp->k = lv.k;
    pp->v = lv.v;
    pp->cu_seqlens_k = lv.cu_seqlens_k;
    pp->k_group_stride = lv.k_group_stride;
    pp->k_tok_stride = lv.k_tok_stride;
    pp->k_head_stride = lv.k_head_stride;
    pp->v_group_stride = lv.v_group_stride;
    pp->v_tok_stride = lv.v_tok_stride;
    pp->v_head_stride = lv.v_head_stride;
    pp->dtype = s.dtype;
    pp->B = s.B;
    pp->nq = s.nq;
    pp->Hq = s.Hq;
    pp->Hkv = s.Hkv;
    pp->D = s.D;
    pp->softmax_scale = s.softmax_scale;
    pp->sb = lv.sb;
    pp->kv_len = lv.kv_len;
    pp->causal = 0;
    pp->lse_layout = HYD_LSE_BQH;
    pp->num_splits = 0;
}

// A shared level whose groups are small (few query rows per (group, kv head), short prefix) wastes the prefix
// kernel: one 512-thread workgroup with ~12 us of fixed cost per (group, head) for a fraction of a 128 x 128 tile
// (C4's second level, 32 groups x 32 queries x 64 keys: 1024 workgroups = 4 rounds = 52 us for 1 GFLOP).  Such a
// level is the matrix-core suffix kernel's problem with "sequence" = group: nq' = per * nq query tokens per group
// over the group's P keys, one wave per (group, kv head, 16-row chunk).  Shapes-only decision (capture-safe).
bool level_is_small(const hyd_prefix_params& pp, const PrefixPlan& pl) {
    if (dev_switch("HYD_LEVEL_PREFIX") || pp.cu_seqlens_k || pp.cu_seqlens_q || pp.causal) return false;
    if (pp.D != 64 && pp.D != 128) return false;
    const int64_t rows = (int64_t)pl.qpg * pl.g;  // query rows per (group, kv head)
    if (rows > 64 || pp.kv_len > 1024) return false;
    const int64_t ts = pp.k_tok_stride > pp.v_tok_stride ? pp.k_tok_stride : pp.v_tok_stride;
    return (int64_t)pp.kv_len * ts * 2 < ((int64_t)1 << 31) && pp.Hkv <= 65535;
}

int run_level_small(const hyd_prefix_params& pp, const PrefixPlan& pl, void* out, float* lse, hipStream_t s) {
    SuffixArgs a;
    memset(&a, 0, sizeof(a));
    a.q = pp.q; a.k = pp.k; a.v = pp.v; a.out = out; a.lse = lse;
    a.k_bs = pp.k_group_stride; a.k_ts = pp.k_tok_stride; a.k_hs = pp.k_head_stride;
    a.v_bs = pp.v_group_stride; a.v_ts = pp.v_tok_stride; a.v_hs = pp.v_head_stride;
    a.B = pp.sb; a.nq = pl.qpg; a.Hq = pp.Hq; a.Hkv = pp.Hkv; a.g = pl.g; a.kv_len = pp.kv_len;
    a.rows = pl.qpg * pl.g;
    a.units = pp.sb * pp.Hkv;
    a.scale_log2e = scale_log2e_of(pp.softmax_scale, pp.D);
    a.shared_kv = 1;
    const int rc = launch_suffix_gqa(a, pp.dtype, pp.D, s);
    return rc ? fail(HYD_ERR_LAUNCH, "small-level kernel launch failed: hip error %d", rc) : HYD_OK;
}

// The suffix epilogue merges at most kMaxCombine partials, so the levels of one decode call share that budget:
// each level may cut its keys into at most kMaxCombine / n_levels slices (>= 8 with HYD_MAX_LEVELS = 8).
int level_split_cap(int n_levels) { return n_levels > 0 ? kMaxCombine / n_levels : kMaxSplits; }

// Scratch of the two-stream form: the unique pass's partial, a 16-bit [B, nq, Hq, D] + its fp32 LSE [B, nq, Hq].
size_t unique_partial_bytes(const hyd_suffix_params& sp) {
    const size_t rows = (size_t)sp.B * sp.nq * sp.Hq;
    return align_up(rows * sp.D * 2, 256) + align_up(rows * 4, 256);
}

// per-level workspace: nsplit == 1 -> one dtype slice + lse; nsplit > 1 -> fp32 slices (prefix_ws_bytes)
size_t level_ws_bytes(const hyd_prefix_params& pp, const PrefixPlan& pl, bool f32_partials) {
    const size_t rows = (size_t)pp.B * pp.nq * pp.Hq;
    const bool small = level_is_small(pp, pl);
    if (pl.nsplit > 1 && !small) return prefix_ws_bytes(&pp, pl);
    return align_up(rows * pp.D * ((f32_partials && !small) ? 4 : 2), 256) + align_up(rows * 4, 256);
}

int check_prefix_ptrs(const hyd_prefix_params* p) {
    int rc;
    if ((rc = check_ptr_align(p->q, "q"))) return rc;
    if ((rc = check_ptr_align(p->k, "shared k"))) return rc;
    if ((rc = check_ptr_align(p->v, "shared v"))) return rc;
    if ((rc = check_stride8(p->k_group_stride, "k_group_stride")) || (rc = check_stride8(p->k_tok_stride, "k_tok_stride")) ||
        (rc = check_stride8(p->k_head_stride, "k_head_stride")) || (rc = check_stride8(p->v_group_stride, "v_group_stride")) ||
        (rc = check_stride8(p->v_tok_stride, "v_tok_stride")) || (rc = check_stride8(p->v_head_stride, "v_head_stride")))
        return rc;
    return HYD_OK;
}

}  // namespace

extern "C" {

int hyd_version(void) { return HYD_VERSION; }

const char* hyd_last_error_string(void) { return g_err; }

int hyd_prefix_plan(const hyd_prefix_params* p, int32_t* num_splits, int32_t* grid, int32_t* split_len) {
    PrefixPlan pl;
    int rc = plan_prefix(p, &pl);
    if (rc) return rc;
    if (num_splits) *num_splits = pl.nsplit;
    if (grid) *grid = pl.grid;
    if (split_len) *split_len = pl.split_len;
    return HYD_OK;
}

size_t hyd_prefix_workspace_bytes(const hyd_prefix_params* p) {
    PrefixPlan pl;
    if (plan_prefix(p, &pl)) return 0;
    return prefix_ws_bytes(p, pl);
}

int hyd_prefix_attn_fwd(const hyd_prefix_params* p, void* stream) {
    PrefixPlan pl;
    int rc = plan_prefix(p, &pl);
    if (rc) return rc;
    if ((rc = check_prefix_ptrs(p))) return rc;
    if ((rc = check_ptr_align(p->out, "out"))) return rc;
    if (p->kv_len == 0) return fail(HYD_ERR_BAD_ARG, "kv_len == 0: attention over no keys is undefined");
    return run_prefix(p, pl, /*merge=*/true, static_cast<hipStream_t>(stream));
}

int hyd_suffix_attn_fwd(const hyd_suffix_params* p, void* stream) {
    int rc = check_suffix(p, true);
    if (rc) return rc;
    if (p->n_partials < 0 || p->n_partials > HYD_MAX_LEVELS) return fail(HYD_ERR_BAD_ARG, "n_partials %d", p->n_partials);
    if (p->kv_len == 0 && p->n_partials == 0) return fail(HYD_ERR_BAD_ARG, "kv_len == 0 and no partials");
    return run_suffix(p, p->partials, p->n_partials, static_cast<hipStream_t>(stream));
}

int hyd_combine_lse(const void* const* outs, const float* const* lses, int32_t n, int64_t rows, int32_t D,
                    int32_t dtype, void* out, float* out_lse, void* stream) {
    if (!outs || !lses || !out) return fail(HYD_ERR_BAD_ARG, "null pointer");
    if (n <= 0 || n > kMaxCombine) return fail(HYD_ERR_UNSUPPORTED, "n = %d partials (1..%d supported)", n, kMaxCombine);
    if (rows < 0 || D <= 0) return fail(HYD_ERR_BAD_ARG, "rows %lld D %d", (long long)rows, D);
    if (dtype != HYD_F16 && dtype != HYD_BF16 && dtype != HYD_F32) return fail(HYD_ERR_UNSUPPORTED, "dtype %d", dtype);
    CombineArgs c;
    memset(&c, 0, sizeof(c));
    bool aligned = (reinterpret_cast<uintptr_t>(out) & 15u) == 0;
    for (int i = 0; i < n; ++i) {
        if (!outs[i] || !lses[i]) return fail(HYD_ERR_BAD_ARG, "partial %d is null", i);
        c.outs[i] = outs[i];
        c.lses[i] = lses[i];
        aligned = aligned && (reinterpret_cast<uintptr_t>(outs[i]) & 15u) == 0;
    }
    c.n = n;
    c.rows = rows;
    c.D = D;
    c.dtype_in = dtype;
    c.dtype_out = dtype;
    c.out = out;
    c.out_lse = out_lse;
    c.lse_layout = HYD_LSE_BQH;
    c.scalar_only = aligned ? 0 : 1;  // unaligned views take the element-wise kernel
    int rc = launch_combine(c, static_cast<hipStream_t>(stream));
    return rc ? fail(HYD_ERR_LAUNCH, "combine kernel launch failed: hip error %d", rc) : HYD_OK;
}

int hyd_rope_append_decode(const hyd_rope_params* p, void* stream) {
    if (!p) return fail(HYD_ERR_BAD_ARG, "null params");
    int rc = check_common(p->dtype, p->B, 1, p->Hq, p->Hkv, p->D);
    if (rc) return rc;
    if ((rc = check_ptr_align(p->q, "q")) || (rc = check_ptr_align(p->k, "k")) || (rc = check_ptr_align(p->v, "v")) ||
        (rc = check_ptr_align(p->q_out, "q_out")) || (rc = check_ptr_align(p->k_cache, "k_cache")) ||
        (rc = check_ptr_align(p->v_cache, "v_cache")) || (rc = check_ptr_align(p->cos, "cos")) ||
        (rc = check_ptr_align(p->sin, "sin")))
        return rc;
    if (!p->position_ids || !p->seq_lens) return fail(HYD_ERR_BAD_ARG, "position_ids / seq_lens is null");
    if ((rc = check_stride8(p->q_batch_stride, "q_batch_stride")) || (rc = check_stride8(p->k_batch_stride, "k_batch_stride")) ||
        (rc = check_stride8(p->v_batch_stride, "v_batch_stride")) || (rc = check_stride8(p->kc_batch_stride, "kc_batch_stride")) ||
        (rc = check_stride8(p->kc_tok_stride, "kc_tok_stride")) || (rc = check_stride8(p->kc_head_stride, "kc_head_stride")) ||
        (rc = check_stride8(p->vc_batch_stride, "vc_batch_stride")) || (rc = check_stride8(p->vc_tok_stride, "vc_tok_stride")) ||
        (rc = check_stride8(p->vc_head_stride, "vc_head_stride")))
        return rc;
    if (p->cs_stride % 4 != 0) return fail(HYD_ERR_BAD_ARG, "cos/sin row stride must be a multiple of 4 floats");
    if (p->cache_len <= 0) return fail(HYD_ERR_BAD_ARG, "cache_len %d", p->cache_len);
    if (p->max_pos <= 0) return fail(HYD_ERR_BAD_ARG, "max_pos %d: the cos/sin tables need at least one row", p->max_pos);
    RopeArgs a;
    memset(&a, 0, sizeof(a));
    a.q = p->q; a.k = p->k; a.v = p->v; a.q_out = p->q_out; a.k_cache = p->k_cache; a.v_cache = p->v_cache;
    a.cos = p->cos; a.sin = p->sin; a.pos = p->position_ids; a.shared_len = p->shared_len; a.seq_lens = p->seq_lens;
    a.q_bs = p->q_batch_stride; a.k_bs = p->k_batch_stride; a.v_bs = p->v_batch_stride;
    a.kc_bs = p->kc_batch_stride; a.kc_ts = p->kc_tok_stride; a.kc_hs = p->kc_head_stride;
    a.vc_bs = p->vc_batch_stride; a.vc_ts = p->vc_tok_stride; a.vc_hs = p->vc_head_stride;
    a.pos_stride = p->pos_stride; a.cs_stride = p->cs_stride;
    a.B = p->B; a.Hq = p->Hq; a.Hkv = p->Hkv; a.cache_len = p->cache_len; a.max_pos = p->max_pos;
    rc = launch_rope_append(a, p->dtype, p->D, static_cast<hipStream_t>(stream));
    return rc ? fail(HYD_ERR_LAUNCH, "rope_append kernel launch failed: hip error %d", rc) : HYD_OK;
}

int hyd_add_rmsnorm(const hyd_add_rmsnorm_params* p, void* stream) {
    if (!p) return fail(HYD_ERR_BAD_ARG, "null params");
    if (p->dtype != HYD_F16 && p->dtype != HYD_BF16) return fail(HYD_ERR_UNSUPPORTED, "dtype %d (fp16 / bf16)", p->dtype);
    if (p->rows < 0 || p->rows > 0x7fffffff) return fail(HYD_ERR_BAD_ARG, "rows %lld", (long long)p->rows);
    if (p->n <= 0 || p->n % 8 != 0 || p->n > 16384) return fail(HYD_ERR_UNSUPPORTED, "n %d: a multiple of 8 up to 16384", p->n);
    if (!p->x || !p->weight || !p->norm_out) return fail(HYD_ERR_BAD_ARG, "x / weight / norm_out is null");
    int rc;
    if ((rc = check_ptr_align(p->x, "x")) || (rc = check_ptr_align(p->weight, "weight")) || (rc = check_ptr_align(p->norm_out, "norm_out")) ||
        (rc = check_stride8(p->x_row_stride, "x_row_stride")) || (rc = check_stride8(p->norm_row_stride, "norm_row_stride")))
        return rc;
    if (p->residual && ((rc = check_ptr_align(p->residual, "residual")) || (rc = check_stride8(p->residual_row_stride, "residual_row_stride"))))
        return rc;
    if (p->residual && p->sum_out && ((rc = check_ptr_align(p->sum_out, "sum_out")) || (rc = check_stride8(p->sum_row_stride, "sum_row_stride"))))
        return rc;
    NormArgs a;
    memset(&a, 0, sizeof(a));
    a.x = p->x; a.residual = p->residual; a.weight = p->weight; a.sum_out = p->residual ? p->sum_out : nullptr; a.norm_out = p->norm_out;
    a.x_rs = p->x_row_stride; a.r_rs = p->residual_row_stride; a.s_rs = p->sum_row_stride; a.o_rs = p->norm_row_stride;
    a.rows = p->rows; a.n = p->n; a.eps = p->eps;
    rc = launch_add_rmsnorm(a, p->dtype, static_cast<hipStream_t>(stream));
    return rc ? fail(HYD_ERR_LAUNCH, "add_rmsnorm kernel launch failed: hip error %d", rc) : HYD_OK;
}

int hyd_swiglu(const hyd_swiglu_params* p, void* stream) {
    if (!p) return fail(HYD_ERR_BAD_ARG, "null params");
    if (p->dtype != HYD_F16 && p->dtype != HYD_BF16) return fail(HYD_ERR_UNSUPPORTED, "dtype %d (fp16 / bf16)", p->dtype);
    if (p->rows < 0 || p->n <= 0 || p->n % 8 != 0) return fail(HYD_ERR_UNSUPPORTED, "rows %lld, n %d: n must be a positive multiple of 8", (long long)p->rows, p->n);
    if (p->rows * (p->n / 8) > 0x7fffffffLL * 256) return fail(HYD_ERR_UNSUPPORTED, "rows x n too large for one launch");
    if (!p->gate || !p->up || !p->out) return fail(HYD_ERR_BAD_ARG, "gate / up / out is null");
    int rc;
    if ((rc = check_ptr_align(p->gate, "gate")) || (rc = check_ptr_align(p->up, "up")) || (rc = check_ptr_align(p->out, "out")) ||
        (rc = check_stride8(p->gate_row_stride, "gate_row_stride")) || (rc = check_stride8(p->up_row_stride, "up_row_stride")) ||
        (rc = check_stride8(p->out_row_stride, "out_row_stride")))
        return rc;
    SwigluArgs a;
    memset(&a, 0, sizeof(a));
    a.gate = p->gate; a.up = p->up; a.out = p->out;
    a.g_rs = p->gate_row_stride; a.u_rs = p->up_row_stride; a.o_rs = p->out_row_stride;
    a.rows = p->rows; a.n = p->n;
    rc = launch_swiglu(a, p->dtype, static_cast<hipStream_t>(stream));
    return rc ? fail(HYD_ERR_LAUNCH, "swiglu kernel launch failed: hip error %d", rc) : HYD_OK;
}

int hyd_sample_tokens(const hyd_sample_params* p, void* stream) {
    if (!p) return fail(HYD_ERR_BAD_ARG, "null params");
    if (p->dtype != HYD_F16 && p->dtype != HYD_BF16 && p->dtype != HYD_F32) return fail(HYD_ERR_UNSUPPORTED, "dtype %d", p->dtype);
    if (p->rows < 0 || p->n <= 0) return fail(HYD_ERR_BAD_ARG, "rows %d, n %d", p->rows, p->n);
    if (!p->logits || !p->out) return fail(HYD_ERR_BAD_ARG, "logits / out is null");
    if (!(p->temperature >= 0.f)) return fail(HYD_ERR_BAD_ARG, "temperature %g must be >= 0", (double)p->temperature);
    if (p->row_stride < p->n) return fail(HYD_ERR_BAD_ARG, "row_stride %lld < n %d", (long long)p->row_stride, p->n);
    const int esz = p->dtype == HYD_F32 ? 4 : 2;
    if ((reinterpret_cast<uintptr_t>(p->logits) & (uintptr_t)(esz - 1)) != 0 || (reinterpret_cast<uintptr_t>(p->out) & 7u) != 0)
        return fail(HYD_ERR_BAD_ARG, "logits / out is not aligned to its element size");
    SampleArgs a;
    memset(&a, 0, sizeof(a));
    a.logits = p->logits; a.out = p->out; a.row_stride = p->row_stride; a.seed = p->seed; a.offset = p->offset;
    a.rows = p->rows; a.n = p->n;
    a.inv_temperature = p->temperature > 0.f ? 1.0f / p->temperature : 0.f;
    a.vec_ok = ((reinterpret_cast<uintptr_t>(p->logits) & 15u) == 0 && p->row_stride % 8 == 0) ? 1 : 0;
    const int rc = launch_sample(a, p->dtype, static_cast<hipStream_t>(stream));
    return rc ? fail(HYD_ERR_LAUNCH, "sample kernel launch failed: hip error %d", rc) : HYD_OK;
}

// hyd_decode_params.single_launch_small: one uniform shared level that already counts as small (few query rows per
// (group, kv head), short prefix), unique keys present, the same token strides in the shared and the unique tensors, and
// so few keys in all that the call is launch latency.  Measured per graph-replayed call (tests/probes/single_launch_probe.py,
// units x keys = B * Hkv * (P + S)): 1152 keys (BASELINE config 1) 6.8 -> 4.1 us, 2176 keys 11.4 -> 8.1, 8072 keys (two
// sequences on a 1000-key prefix) 19.5 -> 16.5, 8704 keys (32 sequences) 14.0 -> 14.1: even.  Shapes only: capture-safe.
constexpr int64_t kSingleLaunchMaxKeys = 8192;
static bool decode_runs_as_one_launch(const hyd_decode_params* p, const hyd_prefix_params* pps, const PrefixPlan* pls, const bool* small) {
    const hyd_suffix_params& sp = p->suffix;
    if (p->phase != HYD_PHASE_ALL || !p->single_launch_small || p->n_levels != 1 || !small[0] || sp.kv_len <= 0) return false;
    if (sp.lse) return false;  // suffix.lse is the LSE of the unique keys alone in every form; one walk over both segments cannot give it
    const hyd_prefix_params& pp = pps[0];
    if (pp.cu_seqlens_k || pp.sb <= 0 || sp.B % pp.sb != 0) return false;  // (a small level runs unsplit whatever the plan says)
    if (pp.k_tok_stride != sp.k_tok_stride || pp.v_tok_stride != sp.v_tok_stride) return false;
    if ((int64_t)sp.B * sp.Hkv * ((int64_t)pp.kv_len + sp.kv_len) > kSingleLaunchMaxKeys) return false;
    SuffixArgs a;
    fill_suffix_args(&sp, &a);
    return suffix_gqa_eligible(a, sp.D, /*any_shape=*/true);
}

// attention.py:273-274: a single shared level and no unique keys -> the prefix result IS the answer
static bool decode_is_prefix_only(const hyd_decode_params* p) { return p->n_levels == 1 && p->suffix.kv_len == 0; }

int hyd_ipc_get_handle(const void* dev_ptr, void* handle_out) {
    static_assert(sizeof(hipIpcMemHandle_t) == HYD_IPC_HANDLE_BYTES, "IPC handle size");
    if (!dev_ptr || !handle_out) return fail(HYD_ERR_BAD_ARG, "null pointer");
    hipIpcMemHandle_t h;
    const hipError_t e = hipIpcGetMemHandle(&h, const_cast<void*>(dev_ptr));
    if (e != hipSuccess) return fail(HYD_ERR_LAUNCH, "hipIpcGetMemHandle: %s", hipGetErrorString(e));
    memcpy(handle_out, &h, sizeof(h));
    return HYD_OK;
}

int hyd_ipc_open_handle(const void* handle, void** dev_ptr_out) {
    if (!handle || !dev_ptr_out) return fail(HYD_ERR_BAD_ARG, "null pointer");
    hipIpcMemHandle_t h;
    memcpy(&h, handle, sizeof(h));
    const hipError_t e = hipIpcOpenMemHandle(dev_ptr_out, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) return fail(HYD_ERR_LAUNCH, "hipIpcOpenMemHandle: %s", hipGetErrorString(e));
    return HYD_OK;
}

int hyd_ipc_close_handle(void* dev_ptr) {
    if (!dev_ptr) return fail(HYD_ERR_BAD_ARG, "null pointer");
    const hipError_t e = hipIpcCloseMemHandle(dev_ptr);
    return e == hipSuccess ? HYD_OK : fail(HYD_ERR_LAUNCH, "hipIpcCloseMemHandle: %s", hipGetErrorString(e));
}

size_t hyd_allreduce_block_bytes(int32_t world, size_t max_bytes) {
    if (world < 1 || world > HYD_ALLREDUCE_MAX_WORLD || max_bytes == 0) return 0;
    return allreduce_block_bytes(world, max_bytes);
}

const uint32_t* hyd_allreduce_status(const void* own_block) {
    // layout of allreduce.hip: 2 x 8 flags of 32 words, then the local words {epoch, arrivals, status}
    return own_block ? static_cast<const uint32_t*>(own_block) + 2 * HYD_ALLREDUCE_MAX_WORLD * 32 + 2 : nullptr;
}

int hyd_allreduce_sum(const hyd_allreduce_params* p, void* stream) {
    if (!p || !p->blocks) return fail(HYD_ERR_BAD_ARG, "null params");
    if (p->world < 1 || p->world > HYD_ALLREDUCE_MAX_WORLD) return fail(HYD_ERR_UNSUPPORTED, "world %d (1..%d)", p->world, HYD_ALLREDUCE_MAX_WORLD);
    if (p->rank < 0 || p->rank >= p->world) return fail(HYD_ERR_BAD_ARG, "rank %d of %d", p->rank, p->world);
    if (p->dtype != HYD_F16 && p->dtype != HYD_BF16 && p->dtype != HYD_F32) return fail(HYD_ERR_UNSUPPORTED, "dtype %d", p->dtype);
    if (p->count < 0) return fail(HYD_ERR_BAD_ARG, "count %lld", (long long)p->count);
    if (p->timeout_log2_polls != 0 && (p->timeout_log2_polls < 10 || p->timeout_log2_polls > 31))
        return fail(HYD_ERR_BAD_ARG, "timeout_log2_polls %d (0 = default, or 10..31)", p->timeout_log2_polls);
    if (p->count == 0) return HYD_OK;
    const size_t bytes = (size_t)p->count * (p->dtype == HYD_F32 ? 4 : 2);
    if (bytes > p->max_bytes) return fail(HYD_ERR_WORKSPACE, "%zu bytes exceed the blocks' max_bytes %zu", bytes, p->max_bytes);
    int rc;
    if ((rc = check_ptr_align(p->in, "in")) || (rc = check_ptr_align(p->out, "out"))) return rc;
    char* blocks[HYD_ALLREDUCE_MAX_WORLD];
    for (int i = 0; i < p->world; ++i) {
        if ((rc = check_ptr_align(p->blocks[i], "block"))) return rc;
        blocks[i] = static_cast<char*>(p->blocks[i]);
    }
    if (p->world == 1) {
        if (p->in != p->out) (void)hipMemcpyAsync(p->out, p->in, bytes, hipMemcpyDeviceToDevice, static_cast<hipStream_t>(stream));
        return HYD_OK;
    }
    rc = launch_allreduce(blocks, allreduce_block_bytes(p->world, p->max_bytes), p->in, p->out, p->count, p->dtype, p->rank,
                          p->world, p->max_bytes, p->timeout_log2_polls, static_cast<hipStream_t>(stream));
    return rc ? fail(HYD_ERR_LAUNCH, "all-reduce kernel launch failed: hip error %d", rc) : HYD_OK;
}

size_t hyd_decode_workspace_bytes(const hyd_decode_params* p) {
    if (!p || p->n_levels < 0 || p->n_levels > HYD_MAX_LEVELS) return 0;
    if (decode_is_prefix_only(p)) {
        // written straight to `out` by the prefix pass; only its split-KV slices (if any) need scratch
        hyd_prefix_params pp;
        level_to_prefix(p, 0, &pp);
        PrefixPlan pl;
        if (plan_prefix(&pp, &pl)) return 0;
        return prefix_ws_bytes(&pp, pl);
    }
    size_t total = 0;
    for (int i = 0; i < p->n_levels; ++i) {
        hyd_prefix_params pp;
        level_to_prefix(p, i, &pp);
        PrefixPlan pl;
        if (plan_prefix(&pp, &pl, level_split_cap(p->n_levels))) return 0;
        total += level_ws_bytes(pp, pl, p->f32_partials != 0);
    }
    // the unique pass's own partial (HYD_PHASE_UNIQUE_PARTIAL / HYD_PHASE_MERGE: the two-stream form)
    if (p->n_levels > 0 && p->suffix.kv_len > 0) total += unique_partial_bytes(p->suffix);
    return total;
}

size_t hyd_workspace_bytes(int32_t B, int32_t nq, int32_t Hq, int32_t Hkv, int32_t D, int32_t n_levels,
                           const int32_t* level_sb, const int32_t* level_kv_len) {
    if (n_levels < 0 || n_levels > HYD_MAX_LEVELS) return 0;
    hyd_decode_params p;
    memset(&p, 0, sizeof(p));
    p.suffix.dtype = HYD_BF16;
    p.suffix.B = B;
    p.suffix.nq = nq;
    p.suffix.Hq = Hq;
    p.suffix.Hkv = Hkv;
    p.suffix.D = D;
    p.suffix.kv_len = 1;  // a decode step has unique keys: size the unique partial of the two-stream form as well
    p.f32_partials = 1;   // an upper bound for every form of the call: an unsplit level's partial may be kept in fp32
    p.n_levels = n_levels;
    for (int i = 0; i < n_levels; ++i) {
        p.levels[i].sb = level_sb[i];
        p.levels[i].kv_len = level_kv_len[i];
    }
    return hyd_decode_workspace_bytes(&p);
}

int hyd_decode_attn_fused(const hyd_decode_params* p, void* stream) {
    if (!p) return fail(HYD_ERR_BAD_ARG, "null params");
    if (p->n_levels < 0 || p->n_levels > HYD_MAX_LEVELS) return fail(HYD_ERR_BAD_ARG, "n_levels %d", p->n_levels);
    if (p->phase < HYD_PHASE_ALL || p->phase > HYD_PHASE_MERGE) return fail(HYD_ERR_BAD_ARG, "phase %d", p->phase);
    if (p->shared_max_workgroups < 0) return fail(HYD_ERR_BAD_ARG, "shared_max_workgroups %d", p->shared_max_workgroups);
    if (p->f32_partials != 0 && p->f32_partials != 1) return fail(HYD_ERR_BAD_ARG, "f32_partials %d", p->f32_partials);
    const hyd_suffix_params& sp = p->suffix;
    int rc = check_suffix(&sp, true);
    if (rc) return rc;
    if (p->n_levels == 0 && sp.kv_len == 0) return fail(HYD_ERR_BAD_ARG, "no shared levels and no unique keys");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t rows = (size_t)sp.B * sp.nq * sp.Hq;
    const bool do_shared = p->phase == HYD_PHASE_ALL || p->phase == HYD_PHASE_SHARED;
    const bool do_unique = p->phase == HYD_PHASE_ALL || p->phase == HYD_PHASE_UNIQUE;
    const bool two_stream = p->phase == HYD_PHASE_UNIQUE_PARTIAL || p->phase == HYD_PHASE_MERGE;
    // The two-stream form needs both a shared and a unique part; without one of them the in-order phases already are
    // the whole operator (the caller asks hyd_decode_two_stream_ok first).
    if (two_stream && (p->n_levels == 0 || sp.kv_len == 0))
        return fail(HYD_ERR_UNSUPPORTED, "two-stream phases need at least one shared level and unique keys");

    if (decode_is_prefix_only(p)) {
        if (!do_shared) return HYD_OK;  // nothing left for the unique phase
        hyd_prefix_params pp;
        level_to_prefix(p, 0, &pp);
        PrefixPlan pl;
        if ((rc = plan_prefix(&pp, &pl))) return rc;
        if ((rc = check_prefix_ptrs(&pp))) return rc;
        if (pp.kv_len == 0) return fail(HYD_ERR_BAD_ARG, "level 0 has kv_len == 0");
        pp.out = sp.out;
        pp.lse = nullptr;
        pp.workspace = p->workspace;
        pp.workspace_bytes = p->workspace_bytes;
        return run_prefix(&pp, pl, true, s, p->shared_max_workgroups);
    }

    // ---- plan and validate everything before the first launch ----------------------------------------------
    hyd_prefix_params pps[HYD_MAX_LEVELS];
    PrefixPlan pls[HYD_MAX_LEVELS];
    bool small[HYD_MAX_LEVELS];
    size_t bytes[HYD_MAX_LEVELS];
    size_t need = 0;
    int n_parts = 0;
    for (int i = 0; i < p->n_levels; ++i) {
        level_to_prefix(p, i, &pps[i]);
        if ((rc = plan_prefix(&pps[i], &pls[i], level_split_cap(p->n_levels)))) return rc;
        if ((rc = check_prefix_ptrs(&pps[i]))) return rc;
        if (pps[i].kv_len == 0) return fail(HYD_ERR_BAD_ARG, "level %d has kv_len == 0", i);
        small[i] = level_is_small(pps[i], pls[i]);
        bytes[i] = level_ws_bytes(pps[i], pls[i], p->f32_partials != 0);
        need += bytes[i];
        n_parts += (pls[i].nsplit == 1 || small[i]) ? 1 : pls[i].nsplit;
    }
    if (n_parts + (two_stream ? 1 : 0) > kMaxCombine) return fail(HYD_ERR_UNSUPPORTED, "%d partials (more than %d)", n_parts, kMaxCombine);
    if (decode_runs_as_one_launch(p, pps, pls, small)) {
        // launch latency, not work: the grouped-query kernel walks the group's shared keys, then the sequence's own
        SuffixArgs a;
        fill_suffix_args(&sp, &a);
        a.pk = pps[0].k; a.pv = pps[0].v;
        a.pk_gs = pps[0].k_group_stride; a.pk_hs = pps[0].k_head_stride;
        a.pv_gs = pps[0].v_group_stride; a.pv_hs = pps[0].v_head_stride;
        a.p_len = pps[0].kv_len;
        a.p_per = sp.B / pps[0].sb;
        a.shared_kv = 1;  // p_per sequences read the same prefix keys: default cache policy, not the read-once hint
        rc = launch_suffix_gqa(a, sp.dtype, sp.D, s);
        return rc ? fail(HYD_ERR_LAUNCH, "single-launch decode kernel failed: hip error %d", rc) : HYD_OK;
    }
    const size_t levels_bytes = need;
    if (two_stream) need += unique_partial_bytes(sp);
    if (need > 0 && (!p->workspace || p->workspace_bytes < need))
        return fail(HYD_ERR_WORKSPACE, "decode needs %zu workspace bytes, got %zu", need, p->workspace_bytes);

    hyd_partial parts[HYD_MAX_LEVELS];
    char* ws = static_cast<char*>(p->workspace);
    for (int i = 0; i < p->n_levels; ++i) {
        hyd_prefix_params& pp = pps[i];
        const PrefixPlan& pl = pls[i];
        const bool part_f32 = p->f32_partials != 0 && !small[i];
        if (pl.nsplit == 1 || small[i]) {
            pp.out = ws;
            pp.lse = reinterpret_cast<float*>(ws + align_up(rows * sp.D * (part_f32 ? 4 : 2), 256));
            parts[i].out = pp.out;
            parts[i].lse = pp.lse;
            parts[i].count = 1;
            parts[i].is_f32 = part_f32 ? 1 : 0;
        } else {
            pp.workspace = ws;
            pp.workspace_bytes = bytes[i];
            parts[i].out = ws;
            parts[i].lse = reinterpret_cast<const float*>(ws + (size_t)pl.nsplit * align_up(rows * sp.D * 4, 256));
            parts[i].count = pl.nsplit;
            parts[i].is_f32 = 1;
        }
        if (do_shared) {
            if (small[i]) rc = run_level_small(pp, pl, const_cast<void*>(parts[i].out), const_cast<float*>(parts[i].lse), s);
            else rc = run_prefix(&pp, pl, /*merge=*/false, s, p->shared_max_workgroups, part_f32);
            if (rc) return rc;
        }
        ws += bytes[i];
    }
    if (two_stream) {
        // the unique pass's partial lives behind the levels' regions
        char* up = static_cast<char*>(p->workspace) + levels_bytes;
        void* u_out = up;
        float* u_lse = reinterpret_cast<float*>(up + align_up(rows * sp.D * 2, 256));
        if (p->phase == HYD_PHASE_UNIQUE_PARTIAL) {
            hyd_suffix_params su = sp;
            su.out = u_out;
            su.lse = u_lse;
            return run_suffix(&su, nullptr, 0, s);
        }
        CombineArgs c;  // HYD_PHASE_MERGE: every level's partial(s) + the unique partial -> out
        memset(&c, 0, sizeof(c));
        int n = 0;
        for (int i = 0; i < p->n_levels; ++i) {
            const size_t esz = parts[i].is_f32 ? 4 : 2;
            const size_t ostride = parts[i].count > 1 ? align_up(rows * sp.D * esz, 256) : rows * sp.D * esz;
            const size_t lstride = parts[i].count > 1 ? align_up(rows * 4, 256) : rows * 4;
            for (int j = 0; j < parts[i].count; ++j, ++n) {
                c.outs[n] = static_cast<const char*>(parts[i].out) + (size_t)j * ostride;
                c.lses[n] = reinterpret_cast<const float*>(reinterpret_cast<const char*>(parts[i].lse) + (size_t)j * lstride);
                if (parts[i].is_f32) c.f32_mask |= 1ull << n;
            }
        }
        c.outs[n] = u_out;
        c.lses[n] = u_lse;
        c.n = n + 1;
        c.rows = (int64_t)rows;
        c.D = sp.D;
        c.dtype_in = c.f32_mask ? HYD_MIXED : sp.dtype;
        c.dtype_out = sp.dtype;
        c.out = sp.out;
        c.lse_layout = HYD_LSE_BQH;
        rc = launch_combine(c, s);
        return rc ? fail(HYD_ERR_LAUNCH, "combine kernel launch failed: hip error %d", rc) : HYD_OK;
    }
    if (!do_unique) return HYD_OK;
    if (sp.kv_len == 0) {
        // several levels, no unique keys: merge the level partials only (suffix contributes lse = -inf)
        hyd_suffix_params s0 = sp;
        s0.seq_lens_i32 = nullptr;
        s0.seq_lens_i64 = nullptr;
        return run_suffix(&s0, parts, p->n_levels, s);
    }
    return run_suffix(&sp, parts, p->n_levels, s);
}

int hyd_decode_two_stream_ok(const hyd_decode_params* p) {
    return p && p->n_levels > 0 && p->n_levels <= HYD_MAX_LEVELS && p->suffix.kv_len > 0 && !decode_is_prefix_only(p) ? 1 : 0;
}

}  // extern "C"
