// Kernel argument blocks + launcher prototypes shared by the C-ABI layer (api.hip) and the kernels.
#pragma once
#include "hyd_common.h"

namespace hyd {

constexpr int kMaxCombine = 64;  // (f32_mask of CombineArgs is 64 bits wide)
constexpr int HYD_MIXED = 3;     // CombineArgs.dtype_in only: the partials are a mix of fp32 and the 16-bit output dtype  // partials one combine launch / one suffix epilogue can merge

// Division by a launch constant without a divide on the device (Granlund & Montgomery, round-up method, exact for every
// 32-bit n and d >= 1): q = (t + ((n - t) >> sh1)) >> sh2 with t = mulhi(mul, n).  The host fills it (make_fastdiv).
struct FastDiv {
    uint32_t mul;
    uint32_t sh;  // sh1 | sh2 << 8
};
inline FastDiv make_fastdiv(uint32_t d) {
    uint32_t l = 0;
    while (l < 32 && (uint64_t(1) << l) < d) ++l;  // ceil(log2 d)
    FastDiv f;
    f.mul = (uint32_t)(((uint64_t(1) << 32) * ((uint64_t(1) << l) - d)) / d + 1);
    f.sh = (l < 1 ? l : 1) | ((l > 1 ? l - 1 : 0) << 8);
    return f;
}

struct PrefixArgs {
    const void* q;
    const void* k;
    const void* v;
    void* out;   // dtype [..] or fp32 when out_f32; split sp writes at out + sp*out_split_stride
    float* lse;  // may be null
    const int32_t* cu_k;
    const int32_t* cu_q;
    int64_t k_gs, k_ts, k_hs, v_gs, v_ts, v_hs;
    int64_t out_split_stride;  // elements
    int64_t lse_split_stride;  // elements
    int32_t B, nq, Hq, Hkv, g, sb, per;
    int32_t kv_len;
    int32_t lse_q_stride;  // BHQ layout: query tokens per group
    int32_t row_blocks, nsplit, split_len;
    int32_t vgrid;    // units = sb * Hkv * nsplit * row_blocks; the launch grid may be smaller (persistent workgroups)
    int32_t wg_rows;  // query rows per workgroup: 128, or 256 (pipelined kernel, D = 128, large row counts)
    int32_t waves;    // waves per workgroup: 4 (one per SIMD, 64-row waves) or 8 (two per SIMD, 32-row waves; D = 128)
    int32_t lse_layout, out_f32;
    float scale_log2e;
    FastDiv div_row_blocks, div_nsplit, div_hkv, div_g;  // the unit / row decode divides by these four
    int32_t dbg;  // 0 in product builds; HYD_ABLATION_BUILD reads HYD_DBG to pick a timing-ablation variant of a prefix kernel
};

struct PartialDev {
    const void* out;
    const float* lse;
    int32_t is_f32;
    int32_t pad;
};

struct SuffixArgs {
    const void* q;
    const void* k;
    const void* v;
    void* out;
    float* lse;
    const int32_t* sl32;
    const int64_t* sl64;
    const int32_t* order;  // optional permutation of the sequences: the unit in dispatch slot i works on sequence order[i] (hyd_suffix_params.seq_order)
    int64_t k_bs, k_ts, k_hs, v_bs, v_ts, v_hs;
    int32_t B, nq, Hq, Hkv, g, kv_len;
    int32_t rows;  // nq * g
    int32_t units; // B * Hkv
    int32_t n_partials;
    float scale_log2e;
    int32_t packed;  // shapes allow the lane-group path for short sequences (set by launch_suffix)
    int32_t n_pre;   // leading 16-bit partials the kernels fetch under the K/V stream (0..2, set by run_suffix)
    int32_t rows_wps_log2;  // token-row kernel (suffix_attn.hip): log2 of the waves of a workgroup that share one sequence (set by its launcher)
    int32_t dbg_blind;  // 0 in product builds; HYD_ABLATION_BUILD: HYD_GQA_BLIND=1 restores the blind first key step of the grouped-query kernel (A/B)
    int32_t shared_kv;  // the keys are read by several workgroups (a small shared level on the grouped-query kernel): no non-temporal hint
    // grouped-query kernel only: a shared-prefix segment walked before the unit's own keys (tiny problems: the whole
    // operator in one launch).  Sequence b reads rows [0, p_len) of group b / p_per; token strides equal k_ts / v_ts.
    const void* pk;
    const void* pv;
    int64_t pk_gs, pk_hs, pv_gs, pv_hs;
    int32_t p_len, p_per;
    PartialDev partials[kMaxCombine];
};

struct CombineArgs {
    const void* outs[kMaxCombine];
    const float* lses[kMaxCombine];
    void* out;
    float* out_lse;
    int64_t rows;
    int32_t n, D, dtype_in, dtype_out;  // dtype_in may be HYD_F32 with a 16-bit dtype_out; HYD_MIXED: per-partial, f32_mask
    uint64_t f32_mask;                  // dtype_in == HYD_MIXED: bit i set = partial i is fp32, else it has dtype_out
    // optional BHQ re-layout of out_lse: row = tok*Hq + h -> ((tok / qpg)*Hq + h)*qpg + tok % qpg
    int32_t lse_layout, Hq, qpg;
    int32_t scalar_only;  // force the element-wise kernel (unaligned tensors)
};

struct RopeArgs {
    const void* q;
    const void* k;
    const void* v;
    void* q_out;
    void* k_cache;
    void* v_cache;
    const float* cos;
    const float* sin;
    const int64_t* pos;
    const int64_t* shared_len;  // may be null
    int32_t* seq_lens;
    int64_t q_bs, k_bs, v_bs;          // batch strides of q/k/v (elements); heads contiguous
    int64_t kc_bs, kc_ts, kc_hs, vc_bs, vc_ts, vc_hs;
    int64_t pos_stride, cs_stride;
    int32_t B, Hq, Hkv, cache_len, max_pos;
};

struct NormArgs {  // layer_ops.hip: h = residual + x; normed = RMSNorm(h) * weight
    const void* x;
    const void* residual;  // may be null (plain RMSNorm of x)
    const void* weight;
    void* sum_out;         // may be null
    void* norm_out;
    int64_t x_rs, r_rs, s_rs, o_rs;  // row strides (elements)
    int64_t rows;
    int32_t n;
    float eps;
};

struct SwigluArgs {  // layer_ops.hip: out = silu(gate) * up
    const void* gate;
    const void* up;
    void* out;
    int64_t g_rs, u_rs, o_rs;  // row strides (elements)
    int64_t rows;
    int32_t n;
};

struct SampleArgs {  // layer_ops.hip: out[row] ~ softmax(logits[row] / T) (Gumbel-max), or argmax when inv_temperature == 0
    const void* logits;
    int64_t* out;
    int64_t row_stride;  // elements
    uint64_t seed, offset;
    int32_t rows, n;
    float inv_temperature;
    int32_t vec_ok;      // rows are 16-byte aligned
};

// launchers (defined next to the kernels); return hipError_t as int
int launch_prefix_w64(const PrefixArgs& a, int dtype, int D, bool causal, int grid, hipStream_t s);
int launch_prefix_w64_f16(const PrefixArgs& a, int D, bool causal, int grid, hipStream_t s);  // prefix_attn_w64_f16.hip
int launch_rope_append(const RopeArgs& a, int dtype, int D, hipStream_t s);
int launch_add_rmsnorm(const NormArgs& a, int dtype, hipStream_t s);
int launch_swiglu(const SwigluArgs& a, int dtype, hipStream_t s);
int launch_sample(const SampleArgs& a, int dtype, hipStream_t s);
int launch_suffix(const SuffixArgs& a, int dtype, int D, hipStream_t s);
bool suffix_gqa_eligible(const SuffixArgs& a, int D, bool any_shape);
int launch_suffix_gqa(const SuffixArgs& a, int dtype, int D, hipStream_t s);
int launch_combine(const CombineArgs& a, hipStream_t s);
size_t allreduce_block_bytes(int world, size_t max_bytes);
int launch_allreduce(char* const* blocks, size_t block_bytes, const void* in, void* out, int64_t count, int dtype,
                     int rank, int world, size_t max_bytes, int timeout_log2_polls, hipStream_t s);

}  // namespace hyd
