/*
 * hydragen_hip.h -- C ABI of libhydragen_hip.so: Hydragen's decomposed shared-prefix attention
 * (prefix pass + suffix pass + log-sum-exp combine) as hand-written HIP kernels for gfx950
 * (MI355X / CDNA4).
 *
 * This is the drop-in boundary for the reference's hot path.  Each entry point names the
 * reference interface it replaces (file:line relative to ScalingIntelligence/hydragen):
 *
 *   hyd_prefix_attn_fwd     hydragen/flash.py:284-306  flash_attention        (K1, K2c)
 *                           hydragen/flash.py:309-351  flash_attention_varlen (K1v)
 *                           i.e. flash-attn 2.3.6 _flash_attn_forward/_flash_attn_varlen_forward
 *                           as called from hydragen/attention.py:270,313,344
 *   hyd_suffix_attn_fwd     hydragen/flash.py:163-281  flash_attention_seqlen
 *                           (= xformers_stuff.py:189-428 _fwd_kernel_splitK + flash.py:76-160
 *                           _splitK_reduce), optionally with attention.py:352 combine fused in
 *   hyd_combine_lse         hydragen/attention.py:154-174 combine_lse (N partials; replaces both
 *                           combine_lse_triton :105-151 and combine_lse_torch :21-43)
 *   hyd_decode_attn_fused   hydragen/attention.py:177-354 hydragen_attention for the decode
 *                           case (seq_lens given): all shared levels + suffix + combine
 *   hyd_workspace_bytes     replaces the per-call torch.empty scratch of flash.py:199-204
 *
 * Conventions:
 *   - plain pointers and sizes only; all pointers are DEVICE pointers unless stated otherwise;
 *   - the library never allocates device memory, never synchronises, reads no environment
 *     variable, holds no mutable global state, and launches only on the hipStream_t passed in
 *     (as void*): every call is HIP-graph-capture safe;
 *   - launch geometry is derived from shapes only, never from device data;
 *   - every call returns HYD_OK (0) or a negative error code; hyd_last_error_string() gives the
 *     message for the calling thread's last failure;
 *   - 16-bit dtypes: HYD_F16 (IEEE half) and HYD_BF16; accumulation, softmax and LSE are fp32.
 *   - q/out are [B, nq, Hq, D] contiguous; K/V tensors are token-major with explicit element
 *     strides and a contiguous head_dim; supported head_dim: 64, 128, 256.
 */
#ifndef HYDRAGEN_HIP_H
#define HYDRAGEN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* the library is built with -fvisibility=hidden: these entry points are its whole dynamic symbol table */
#define HYD_API __attribute__((visibility("default")))

#define HYD_VERSION 500 /* 0.5.0: hyd_suffix_params.seq_order (schedule hint for ragged lengths); 0.4.0: hyd_add_rmsnorm, hyd_swiglu, hyd_sample_tokens (model-shell glue); 0.3.0: two-stream phases + hyd_decode_params.shared_max_workgroups, hyd_decode_two_stream_ok; 0.2.2: hyd_allreduce_params.timeout_log2_polls; 0.2.1: softmax_scale; 0.2.0: hyd_decode_params.phase, hyd_rope_params.max_pos, hyd_allreduce_* */
#define HYD_MAX_LEVELS 8

enum {
    HYD_OK = 0,
    HYD_ERR_BAD_ARG = -1,     /* null pointer, non-positive size, indivisible batch ...      */
    HYD_ERR_UNSUPPORTED = -2, /* dtype / head_dim / row count the kernels do not implement    */
    HYD_ERR_WORKSPACE = -3,   /* workspace missing or smaller than hyd_*_workspace_bytes says */
    HYD_ERR_LAUNCH = -4       /* hipLaunchKernel reported an error                            */
};

enum { HYD_F16 = 0, HYD_BF16 = 1, HYD_F32 = 2 /* hyd_combine_lse only */ };

/* LSE layouts: BQH = [B, nq, Hq] (what attention.py:276-280 re-lays flash's output into),
 *              BHQ = [sb, Hq, (B/sb)*nq] (what flash-attn returns, flash.py:295-306). */
enum { HYD_LSE_BQH = 0, HYD_LSE_BHQ = 1 };

/* ------------------------------------------------------------------------------------------
 * Prefix pass: batched-query attention of every query of a group against that group's single
 * shared K/V, on the MFMA matrix cores.  Sequence b belongs to group b / (B/sb)
 * (attention.py:264-268).  GQA: q-head h reads kv-head h / (Hq/Hkv).
 * ------------------------------------------------------------------------------------------ */
typedef struct hyd_prefix_params {
    const void* q;               /* [B, nq, Hq, D]                                             */
    const void* k;               /* group gi, token t, head h at k + gi*k_group_stride +       */
    const void* v;               /*   t*k_tok_stride + h*k_head_stride (elements); D contiguous */
    void* out;                   /* [B, nq, Hq, D] in dtype                                    */
    float* lse;                  /* may be NULL; natural log, softmax scale included           */
    const int32_t* cu_seqlens_k; /* NULL, or [sb+1]: group gi owns packed tokens               */
                                 /*   [cu[gi], cu[gi+1]) of k/v (k_group_stride ignored)        */
    const int32_t* cu_seqlens_q; /* NULL, or [sb+1] packed query tokens per group (then B is   */
                                 /*   the total number of query tokens and nq must be 1)       */
    void* workspace;             /* >= hyd_prefix_workspace_bytes(); may be NULL if that is 0  */
    size_t workspace_bytes;
    int64_t k_group_stride, k_tok_stride, k_head_stride;
    int64_t v_group_stride, v_tok_stride, v_head_stride;
    int32_t dtype;      /* HYD_F16 | HYD_BF16                                                  */
    int32_t B, nq, Hq, Hkv, D;
    int32_t sb;         /* number of groups (shared sequences); B % sb == 0                    */
    int32_t kv_len;     /* keys per group; with cu_seqlens_k: the maximum over groups          */
    int32_t max_q_len;  /* only with cu_seqlens_q: max query tokens per group                  */
    int32_t causal;     /* 0 | 1 (bottom-right aligned: query i sees keys j <= i + kv - nq)    */
    int32_t lse_layout; /* HYD_LSE_BQH | HYD_LSE_BHQ                                           */
    int32_t num_splits; /* split-KV factor; 0 = choose from shapes                             */
    float softmax_scale; /* 0 = D^-0.5 (what the reference always uses, flash.py:295-304); > 0: that
                          * scale -- lets a caller run a zero-padded head dim with the true one's scale */
    int32_t reserved_;
} hyd_prefix_params;

HYD_API size_t hyd_prefix_workspace_bytes(const hyd_prefix_params* p);
HYD_API int hyd_prefix_attn_fwd(const hyd_prefix_params* p, void* stream);

/* ------------------------------------------------------------------------------------------
 * Suffix pass: every query row of sequence b against the first seq_len[b] keys of b's own
 * K/V (non-causal), one wavefront per (sequence, kv-head), HBM-bandwidth bound.  When
 * n_partials > 0 the log-sum-exp merge with already-computed partial results (the prefix
 * passes) is done in the epilogue and `out` is the final attention output.
 * ------------------------------------------------------------------------------------------ */
typedef struct hyd_partial {
    const void* out;  /* [count][B, nq, Hq, D]; dtype, or fp32 when is_f32                      */
    const float* lse; /* [count][B, nq, Hq]                                                     */
    int32_t count;    /* number of stacked partials behind these pointers (split-KV slices)    */
    int32_t is_f32;
} hyd_partial;

typedef struct hyd_suffix_params {
    const void* q;              /* [B, nq, Hq, D]                                              */
    const void* k;              /* sequence b, token t, head h at k + b*k_batch_stride +       */
    const void* v;              /*   t*k_tok_stride + h*k_head_stride (elements)               */
    void* out;                  /* [B, nq, Hq, D] in dtype                                     */
    float* lse;                 /* [B, nq, Hq] or NULL; LSE of the suffix pass alone           */
    const int32_t* seq_lens_i32; /* [B] or NULL                                                */
    const int64_t* seq_lens_i64; /* [B] or NULL (the reference's callers hold int64:           */
                                 /*   llama.py:569); both NULL = every sequence uses kv_len    */
    const int32_t* seq_order;    /* [B] or NULL: a PERMUTATION of 0..B-1, the order in which   */
                                 /*   the sequences are handed to the chip (longest first keeps */
                                 /*   the last workgroups short when lengths are ragged: C2     */
                                 /*   heads, lengths 1..128 at random: 184 -> 169 us).  Only   */
                                 /*   the schedule depends on it, never a result; the caller    */
                                 /*   vouches that it is a permutation.  During decode every    */
                                 /*   length grows by one per step, so one argsort at the start */
                                 /*   of a generation serves all of its steps.                  */
    int64_t k_batch_stride, k_tok_stride, k_head_stride;
    int64_t v_batch_stride, v_tok_stride, v_head_stride;
    int32_t dtype;
    int32_t B, nq, Hq, Hkv, D;
    int32_t kv_len;             /* allocated keys per sequence (Mk); lengths are clamped to it */
    int32_t n_partials;         /* entries used in partials[]                                  */
    float softmax_scale;        /* 0 = D^-0.5; > 0: that scale (hyd_decode_attn_fused applies it to    */
    int32_t reserved_;          /*   every level as well)                                              */
    hyd_partial partials[HYD_MAX_LEVELS];
} hyd_suffix_params;

HYD_API int hyd_suffix_attn_fwd(const hyd_suffix_params* p, void* stream);

/* ------------------------------------------------------------------------------------------
 * combine_lse for N partials (attention.py:21-43): out = sum_i out_i*exp(lse_i-m) / sum_i exp(lse_i-m).
 * rows = B*nq*Hq.  `outs`/`lses` are HOST arrays of n device pointers.  dtype may be HYD_F32.
 * out_lse (may be NULL) receives the merged LSE  m + log(sum_i exp(lse_i - m)).
 * ------------------------------------------------------------------------------------------ */
HYD_API int hyd_combine_lse(const void* const* outs, const float* const* lses, int32_t n, int64_t rows, int32_t D,
                    int32_t dtype, void* out, float* out_lse, void* stream);

/* ------------------------------------------------------------------------------------------
 * Whole decode-step operator: hydragen_attention with seq_lens given (attention.py:177-354):
 * one prefix pass per shared level into the workspace, then the suffix pass with the merge
 * fused into its epilogue.  With kv_len == 0 and one level the prefix result is written to
 * `out` directly (attention.py:273-274).
 * ------------------------------------------------------------------------------------------ */
typedef struct hyd_level {
    const void* k;
    const void* v;
    const int32_t* cu_seqlens_k; /* NULL for uniform levels                                   */
    int64_t k_group_stride, k_tok_stride, k_head_stride;
    int64_t v_group_stride, v_tok_stride, v_head_stride;
    int32_t sb;                  /* shared sequences in this level                            */
    int32_t kv_len;              /* uniform length, or max length when cu_seqlens_k != NULL   */
} hyd_level;

/* Which part of the operator a call enqueues.  ALL is the normal one-call form.  SHARED runs only the
 * per-level prefix passes (they read q and the shared caches and fill the workspace); UNIQUE runs only
 * the suffix pass + merge and expects the workspace as a SHARED call with the same parameters left it.
 * The two halves touch disjoint inputs (the unique K/V and seq_lens are read by UNIQUE only), so a
 * caller may run SHARED on one stream while this step's k/v are still being appended on another.
 *
 * Two-stream form (the reference issues the two passes one after the other, attention.py:250-352; they do not
 * depend on each other, the prefix pass is matrix-core bound and the suffix pass HBM bound):
 *   stream A: SHARED            (with shared_max_workgroups > 0 the prefix pass keeps to that many CUs)
 *   stream B: UNIQUE_PARTIAL    suffix pass alone; its normalised partial + LSE go to the workspace
 *   join, then MERGE            log-sum-exp combine of every partial into `out` (attention.py:21-43)
 * with the same parameters (and workspace) in all three calls.  The caller owns the streams and the
 * fork / join (events, or the edges of a captured graph); hyd_decode_two_stream_ok() says whether the
 * shapes have both parts.  Results equal the one-call form up to one extra rounding of the unique partial
 * to the 16-bit dtype. */
enum { HYD_PHASE_ALL = 0, HYD_PHASE_SHARED = 1, HYD_PHASE_UNIQUE = 2, HYD_PHASE_UNIQUE_PARTIAL = 3, HYD_PHASE_MERGE = 4 };

typedef struct hyd_decode_params {
    hyd_suffix_params suffix;    /* q, unique k/v, seq_lens, out; n_partials/partials ignored */
    hyd_level levels[HYD_MAX_LEVELS];
    int32_t n_levels;
    int32_t phase;               /* HYD_PHASE_*                                               */
    void* workspace;             /* >= hyd_decode_workspace_bytes()                           */
    size_t workspace_bytes;
    int32_t shared_max_workgroups; /* 0 = one workgroup per unit of a prefix pass (the whole chip); > 0: at most
                                    * that many persistent workgroups, one per CU, walk the units           */
    int32_t f32_partials;          /* 0: an unsplit level's partial is stored in the 16-bit dtype (what the reference
                                    * does: its flash-attn output is 16-bit, README.md:488-490); 1: kept fp32 (one
                                    * rounding less, + 2 bytes per output element written and read back)           */
    int32_t single_launch_small;   /* 1 (HYD_PHASE_ALL only): a problem so small that it is launch latency, not work --
                                    * one uniform shared level with few query rows per (group, kv head), short prefix,
                                    * few keys in all -- runs as ONE kernel that walks the group's shared keys and then
                                    * the sequence's own (what the no-sharing baseline does, without the replicated
                                    * prefix).  The result then differs from the phase-split forms by their roundings of
                                    * the partial.  Not taken when suffix.lse is set (that is the LSE of the unique
                                    * keys alone in every form).  0: always the prefix pass + suffix pass pair.       */
    int32_t reserved;
} hyd_decode_params;

HYD_API size_t hyd_decode_workspace_bytes(const hyd_decode_params* p);
HYD_API int hyd_decode_attn_fused(const hyd_decode_params* p, void* stream);
/* 1 when the shapes have both a shared and a unique part (the two-stream phases apply), else 0. */
HYD_API int hyd_decode_two_stream_ok(const hyd_decode_params* p);

/* Upper bound helper mirroring SURVEY 8b's `hyd_workspace_bytes(shape...)`: bytes that
 * hyd_decode_attn_fused needs for n_levels uniform levels of the given shapes, in ANY form of the
 * call (every phase, f32_partials set or not: unsplit levels are sized with fp32 partials). */
HYD_API size_t hyd_workspace_bytes(int32_t B, int32_t nq, int32_t Hq, int32_t Hkv, int32_t D, int32_t n_levels,
                           const int32_t* level_sb, const int32_t* level_kv_len);

/* ------------------------------------------------------------------------------------------
 * Decode-step preamble of the attention block: RoPE of this step's q and k at
 * absolute positions (llama.py:485-501), append of k/v into the unique caches at index
 * position - shared_len (llama.py:236-262, 487-492) and seq_lens = index + 1 (llama.py:569),
 * in one kernel.  q/k/v are [B, 1, H, D] with heads contiguous; cos/sin are fp32 [max_pos, D]
 * tables in the rotate-half convention (first D/2 columns are read).
 * ------------------------------------------------------------------------------------------ */
typedef struct hyd_rope_params {
    const void* q;               /* [B, 1, Hq, D], batch stride q_batch_stride                   */
    const void* k;               /* [B, 1, Hkv, D]                                               */
    const void* v;               /* [B, 1, Hkv, D]                                               */
    void* q_out;                 /* [B, 1, Hq, D] contiguous: rotated queries                    */
    void* k_cache;               /* [maxB, cache_len, Hkv, D] with the strides below            */
    void* v_cache;
    const float* cos;            /* [max_pos, D] fp32, row stride cs_stride                      */
    const float* sin;
    const int64_t* position_ids; /* [B] absolute positions, element stride pos_stride            */
    const int64_t* shared_len;   /* [B] or NULL (= 0)                                            */
    int32_t* seq_lens;           /* out [B]                                                      */
    int64_t q_batch_stride, k_batch_stride, v_batch_stride;
    int64_t kc_batch_stride, kc_tok_stride, kc_head_stride;
    int64_t vc_batch_stride, vc_tok_stride, vc_head_stride;
    int64_t pos_stride, cs_stride;
    int32_t dtype, B, Hq, Hkv, D, cache_len;
    int32_t max_pos;             /* rows of the cos/sin tables; positions are clamped to it (the  */
    int32_t reserved;            /*   host checks the range before launching: a kernel cannot raise) */
} hyd_rope_params;

HYD_API int hyd_rope_append_decode(const hyd_rope_params* p, void* stream);

/* ------------------------------------------------------------------------------------------
 * Elementwise glue of the decoder layer around the attention block (the model shell); rows are
 * tokens, n the hidden / intermediate size, every row 16-byte aligned (n and the row strides multiples of 8).
 *
 * hyd_add_rmsnorm: sum_out = residual + x rounded to dtype (the residual stream of llama.py:615-631: `hidden_states =
 * residual + hidden_states`), norm_out = sum_out * rsqrt(mean(sum_out^2) + eps) * weight with fp32 statistics and one
 * rounding (transformers' LlamaRMSNorm, constructed at llama.py:605-608,656) in one pass.  residual == NULL: plain
 * RMSNorm of x (sum_out ignored).  sum_out may alias residual or x.  n <= 16384.
 *
 * hyd_swiglu: out = silu(gate) * up (transformers' LlamaMLP, llama.py:2,604: down_proj(act_fn(gate_proj(x)) *
 * up_proj(x))), fp32 maths, one rounding; gate / up may be the column halves of one fused GEMM output.
 * ------------------------------------------------------------------------------------------ */
typedef struct hyd_add_rmsnorm_params {
    const void* x;         /* [rows, n] block output (o_proj / down_proj), row stride x_row_stride        */
    const void* residual;  /* [rows, n] or NULL                                                          */
    const void* weight;    /* [n], dtype                                                                 */
    void* sum_out;         /* [rows, n] or NULL                                                          */
    void* norm_out;        /* [rows, n]                                                                  */
    int64_t x_row_stride, residual_row_stride, sum_row_stride, norm_row_stride; /* elements              */
    int64_t rows;
    int32_t n;
    int32_t dtype;         /* HYD_F16 | HYD_BF16                                                         */
    float eps;
    int32_t reserved;
} hyd_add_rmsnorm_params;

HYD_API int hyd_add_rmsnorm(const hyd_add_rmsnorm_params* p, void* stream);

typedef struct hyd_swiglu_params {
    const void* gate;      /* [rows, n], row stride gate_row_stride                                       */
    const void* up;        /* [rows, n]                                                                  */
    void* out;             /* [rows, n]                                                                  */
    int64_t gate_row_stride, up_row_stride, out_row_stride; /* elements                                  */
    int64_t rows;
    int32_t n;
    int32_t dtype;
} hyd_swiglu_params;

HYD_API int hyd_swiglu(const hyd_swiglu_params* p, void* stream);

/* Next token of every sequence from its last-position logits: out[row] ~ softmax(logits[row] / temperature), what
 * `sample_from_logits` (llama.py: softmax(logits / temperature) + torch.multinomial(num_samples=1)) draws, by the
 * Gumbel-max identity argmax_v(logits_v / temperature + g_v) in one pass over the logits; temperature == 0: plain
 * argmax (lowest index on ties), as the reference's temperature-0 branch.  The noise is Philox4x32-10 keyed by
 * (seed, offset, row, column): the same (seed, offset) gives the same tokens on any device / launch geometry; the
 * caller advances `offset` by one per call.  logits: [rows, n] HYD_F16 | HYD_BF16 | HYD_F32, row stride in elements. */
typedef struct hyd_sample_params {
    const void* logits;
    int64_t* out;          /* [rows]                                                                     */
    int64_t row_stride;
    uint64_t seed, offset;
    int32_t rows, n;
    int32_t dtype;
    float temperature;     /* >= 0                                                                       */
} hyd_sample_params;

HYD_API int hyd_sample_tokens(const hyd_sample_params* p, void* stream);

/* ------------------------------------------------------------------------------------------
 * All-reduce(sum) of the tensor-parallel block output (hydragen/tp.py:83-87 after down_proj, :108-112 after
 * o_proj; the reference calls torch.distributed / NCCL there) as a two-shot direct exchange over
 * peer-mapped device memory: xGMI is a full mesh, so every rank reads its slice straight from every
 * peer (reduce-scatter), then every reduced slice from its owner (all-gather).  One process per GPU.
 *
 * Each rank owns one zero-initialised "shared block" of hyd_allreduce_block_bytes() bytes of UNCACHED device
 * memory -- hipExtMallocWithFlags(&p, bytes, hipDeviceMallocUncached), or hipDeviceMallocFinegrained where that is
 * refused; NEVER plain hipMalloc: peers write the block's flags and read its staged payload through the fabric,
 * which does not probe the owner's L2 for coarse-grained allocations (stale reads) -- exports it with
 * hyd_ipc_get_handle, and maps every peer's block with hyd_ipc_open_handle (handles travel over any host channel,
 * e.g. torch.distributed all_gather_object).
 * `blocks` is a HOST array of `world` device pointers: blocks[r] is rank r's block as mapped in THIS
 * process (blocks[rank] is the own block).  `in` / `out` are ordinary device buffers, 16-byte aligned,
 * count * sizeof(dtype) <= max_bytes; in == out is allowed.  Every rank must issue the same sequence of
 * calls.  Capture-safe: the call's epoch lives in the block, not in the arguments.
 * Waiting is bounded by a POLL COUNT, 2^timeout_log2_polls polls of ~0.3 us each per wait (0 = the default
 * 2^27, ~40 s -- longer than any lazy module load, graph capture or shard load between two ranks' calls):
 * a peer that never shows up makes the kernel give up instead of hanging the device; `out` is then NOT the
 * sum, and the block's status word (hyd_allreduce_status) is 1 / 2.  The caller MUST read that word before
 * trusting results of a run (hydragen_amd.tp.check_collectives does, at the end of every generate()).
 * ------------------------------------------------------------------------------------------ */
#define HYD_IPC_HANDLE_BYTES 64
#define HYD_ALLREDUCE_MAX_WORLD 8
HYD_API int hyd_ipc_get_handle(const void* dev_ptr, void* handle_out);
HYD_API int hyd_ipc_open_handle(const void* handle, void** dev_ptr_out);
HYD_API int hyd_ipc_close_handle(void* dev_ptr);

typedef struct hyd_allreduce_params {
    void* const* blocks;  /* HOST array [world] of device pointers (see above)                      */
    const void* in;
    void* out;
    int64_t count;        /* elements                                                              */
    size_t max_bytes;     /* the value the blocks were sized with                                  */
    int32_t dtype;        /* HYD_F16 | HYD_BF16 | HYD_F32; accumulation in fp32                    */
    int32_t rank, world;  /* world <= HYD_ALLREDUCE_MAX_WORLD                                      */
    int32_t timeout_log2_polls; /* 0 = default (27); 10..31: each wait gives up after 2^n polls    */
} hyd_allreduce_params;

HYD_API size_t hyd_allreduce_block_bytes(int32_t world, size_t max_bytes);
HYD_API int hyd_allreduce_sum(const hyd_allreduce_params* p, void* stream);
/* Device pointer of the status word inside a block (uint32: 0 = ok, 1 / 2 = a peer timed out in shot 1 / 2). */
HYD_API const uint32_t* hyd_allreduce_status(const void* own_block);

HYD_API int hyd_version(void);
HYD_API const char* hyd_last_error_string(void);

/* Planner query: the split-KV factor, grid and keys per split the prefix pass derives from these shapes (what
 * hyd_prefix_workspace_bytes sizes the scratch for). */
HYD_API int hyd_prefix_plan(const hyd_prefix_params* p, int32_t* num_splits, int32_t* grid, int32_t* split_len);

#ifdef __cplusplus
}
#endif
#endif /* HYDRAGEN_HIP_H */
