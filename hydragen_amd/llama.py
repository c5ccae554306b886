"""
Model shell around the hot path: the counterpart of /root/reference/hydragen/llama.py with the same
class / method names and semantics (`HydragenLlamaForCausalLM.setup_caches / graph / append_shared /
process_unique / generate / empty_shared_cache / truncate_shared_caches`), written from scratch so it
does not depend on a particular `transformers` version (the reference pins 4.37.2, llama.py:1-10).

What is native here: every attention call goes to the HIP kernels (hydragen_amd.attention / flash), and
the per-token RoPE + unique-KV append + seq_lens computation of the decode step is ONE HIP kernel
(`hyd_rope_append_decode`, replacing llama.py:236-262,317-330,485-501,565-569).  The dense layers
(q/k/v/o projections, SwiGLU MLP, lm_head) are plain torch matmuls (hipBLASLt) -- plumbing, as in the
reference.  Weights are random-initialised from a config (`from_config`); loading Hugging Face
checkpoints is out of scope here (no network), but `load_state_dict` of a HF Llama state dict works
because parameter names match (llama.py:1398-1422).

Decode runs under a HIP graph exactly like the reference's CUDA-graph wrapper (llama.py:781-866).
"""

from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Union

import torch
from torch import Tensor, nn

from . import layer_ops, placement
from .attention import hydragen_attention
from . import flash as _flash
from .flash import flash_attention, flash_attention_seqlen
from .tp import all_reduce_sum, check_collectives


@dataclass
class LlamaConfig:
    """The subset of transformers.LlamaConfig the reference reads (llama.py:423-462,635-653)."""

    hidden_size: int = 4096
    intermediate_size: int = 11008
    num_hidden_layers: int = 32
    num_attention_heads: int = 32
    num_key_value_heads: int = 32
    vocab_size: int = 32000
    rms_norm_eps: float = 1e-5
    rope_theta: float = 10000.0
    max_position_embeddings: int = 4096
    attention_bias: bool = False
    pad_token_id: Optional[int] = None

    @staticmethod
    def llama2_7b(**kw):
        return LlamaConfig(**kw)

    @staticmethod
    def llama3_70b(**kw):
        return LlamaConfig(hidden_size=8192, intermediate_size=28672, num_hidden_layers=80, num_attention_heads=64,
                           num_key_value_heads=8, vocab_size=128256, rope_theta=500000.0,
                           max_position_embeddings=8192, **kw)


def repeat_to_batch_size(tensors: list[Tensor], target_batch_size: int | None = None):
    """llama.py:32-46."""
    if target_batch_size is None:
        target_batch_size = max(t.shape[0] for t in tensors)
    out = []
    for t in tensors:
        assert target_batch_size % t.shape[0] == 0
        out.append(t.repeat_interleave(target_batch_size // t.shape[0], dim=0))
    return out


class RMSNorm(nn.Module):
    def __init__(self, hidden_size, eps=1e-6):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(hidden_size))
        self.variance_epsilon = eps

    def forward(self, x):
        # one fused kernel (fp32 statistics inside) instead of the seven elementwise / reduce launches of the spelled-out
        # form (float, pow, mean, add + rsqrt, mul, cast, mul): 2 norms x 32 layers of them were 13 % of a decode step.
        # Deviation from transformers' LlamaRMSNorm (stated): that one rounds x * rsqrt(var + eps) to the
        # 16-bit dtype BEFORE the multiply by `weight`; this form (and hyd_add_rmsnorm, csrc/layer_ops.hip) multiplies in
        # fp32 and rounds once -- at most one 16-bit ulp per element apart, the closer of the two to the fp32 formula
        # (tests/test_layer_ops_gpu.py bounds both against it).
        return nn.functional.rms_norm(x, (x.shape[-1],), self.weight, self.variance_epsilon)


def _fused_weight(cache: Optional[Tensor], linears) -> Optional[Tensor]:
    """One [sum of out_features, in_features] weight behind several column-parallel nn.Linear modules that read the same
    input (q|k|v, gate|up): the decode step then runs one GEMM instead of two or three (MI355X, batch 1024, Llama-2-7B:
    129 -> 86 us for q, k, v and 167 -> 154 us for gate, up per layer; tests/probes/gemm_fusion_probe.py).  The modules'
    own `weight` parameters become VIEWS of the fused tensor, so state dicts, in-place loading, `make_tp_files` and the
    reference's attribute names keep working and no memory is added; anything that replaces ANY of the weights (`apply_tp`,
    `.to()`, a quantised or merged single projection) is noticed by its address and everything is fused again.  The shared
    storage means that a per-tensor ownership check (safetensors `save_file`) sees aliases: save `state_dict()` clones or
    call `unfuse_weights(model)` first.  None when it does not apply (biases, meta / CPU weights, graph capture)."""
    w0 = linears[0].weight
    if cache is not None and w0.shape[1] == cache.shape[1]:
        # every module's weight must still BE its slice of the fused tensor: any of them may have been replaced on its own
        # (quantisation, a LoRA merge, load_state_dict(assign=True) on part of the modules)
        off, row = cache.data_ptr(), cache.shape[1] * cache.element_size()
        for l in linears:
            if l.weight.data_ptr() != off or l.weight.dtype != cache.dtype or not l.weight.is_contiguous():
                break
            off += l.weight.shape[0] * row
        else:
            if off == cache.data_ptr() + cache.shape[0] * row:
                return cache
    if any(l.bias is not None for l in linears) or not w0.is_cuda or torch.cuda.is_current_stream_capturing():
        return None
    with torch.no_grad():
        w = torch.cat([l.weight for l in linears], 0)
        o = 0
        for l in linears:
            n = l.weight.shape[0]
            l.weight.data = w[o:o + n]
            o += n
    return w


def unfuse_weights(module: nn.Module) -> None:
    """Give every projection its own storage again (undoes `_fused_weight` until the next forward fuses anew)."""
    for m in module.modules():
        for attr, names in (("_gate_up", ("gate_proj", "up_proj")), ("_qkv", ("q_proj", "k_proj", "v_proj"))):
            if getattr(m, attr, None) is not None:
                for n in names:
                    lin = getattr(m, n)
                    lin.weight.data = lin.weight.data.clone()
                setattr(m, attr, None)


class LlamaMLP(nn.Module):
    def __init__(self, config: LlamaConfig):
        super().__init__()
        self.gate_proj = nn.Linear(config.hidden_size, config.intermediate_size, bias=False)
        self.up_proj = nn.Linear(config.hidden_size, config.intermediate_size, bias=False)
        self.down_proj = nn.Linear(config.intermediate_size, config.hidden_size, bias=False)
        self.tp_reduce = False  # row-parallel down_proj needs the all-reduce of tp.py:83-87
        self._gate_up: Optional[Tensor] = None

    def forward(self, x):
        self._gate_up = w = _fused_weight(self._gate_up, (self.gate_proj, self.up_proj))
        if w is not None:
            g, u = nn.functional.linear(x, w).split(self.gate_proj.weight.shape[0], dim=-1)
        else:
            g, u = self.gate_proj(x), self.up_proj(x)
        # silu(gate) * up: one HIP kernel over the two column halves of the fused GEMM output (the strided silu and the
        # strided multiply were 15 + 19 us per layer at batch 1024); the torch form on CPU / fp32 / odd widths
        y = self.down_proj(layer_ops.swiglu(g, u) if layer_ops.supported(g) else nn.functional.silu(g) * u)
        return all_reduce_sum(y) if self.tp_reduce else y


class RotaryTable(nn.Module):
    """cos/sin cache [max_pos, head_dim] in the rotate-half convention the reference uses through HF's
    apply_rotary_pos_emb (llama.py:49-55,494-501)."""

    def __init__(self, dim, max_position_embeddings, base, device=None):
        super().__init__()
        inv_freq = 1.0 / (base ** (torch.arange(0, dim, 2, dtype=torch.float32, device=device) / dim))
        t = torch.arange(max_position_embeddings, dtype=torch.float32, device=device)
        freqs = torch.outer(t, inv_freq)
        emb = torch.cat((freqs, freqs), dim=-1)
        self.register_buffer("cos_cached", emb.cos(), persistent=False)
        self.register_buffer("sin_cached", emb.sin(), persistent=False)


def rotate_half(x):
    x1, x2 = x[..., : x.shape[-1] // 2], x[..., x.shape[-1] // 2 :]
    return torch.cat((-x2, x1), dim=-1)


def apply_rotary_pos_emb(q, k, cos, sin, position_ids):
    """q,k [b, s, h, d]; cos/sin [max_pos, d]; position_ids [b, s] absolute (llama.py:485-501)."""
    cos = cos[position_ids].unsqueeze(2).to(q.dtype)
    sin = sin[position_ids].unsqueeze(2).to(q.dtype)
    return (q * cos) + (rotate_half(q) * sin), (k * cos) + (rotate_half(k) * sin)


class SharedCache(nn.Module):
    """llama.py:58-170: one shared level, sequences packed back to back in [maxB*maxP, Hkv, D]."""

    def __init__(self, max_batch_size, max_seq_length, num_heads, head_dim, dtype, device):
        super().__init__()
        self.register_buffer("k_cache", torch.zeros((max_batch_size * max_seq_length, num_heads, head_dim), dtype=dtype, device=device), persistent=False)
        self.register_buffer("v_cache", torch.zeros_like(self.k_cache), persistent=False)
        self.register_buffer("seq_lens", torch.zeros((max_batch_size,), dtype=torch.int32, device=device), persistent=False)
        self.register_buffer("cumsum_lengths", torch.zeros((max_batch_size + 1,), dtype=torch.int32, device=device), persistent=False)
        self.max_batch_size = max_batch_size
        self.max_sequence_length = max_seq_length
        self.current_batch_size = 0
        self.use_varlen = False
        self.sliced_sequence_length = None

    def get_current_batch_size(self):
        return self.current_batch_size

    def fill(self, key_states: Tensor, value_states: Tensor, seq_lens: Tensor):
        bs = key_states.shape[0]
        if bs > self.max_batch_size:
            raise ValueError(f"Batch size {bs} exceeds max batch size {self.max_batch_size}")
        if key_states.shape[1] > self.max_sequence_length:
            raise ValueError(f"Sequence length {key_states.shape[1]} exceeds max sequence length {self.max_sequence_length}")
        lens = [int(x) for x in seq_lens.tolist()]  # host sync, as llama.py:159-167
        ks = torch.cat([key_states[i, : lens[i]] for i in range(bs)], dim=0)
        vs = torch.cat([value_states[i, : lens[i]] for i in range(bs)], dim=0)
        self.k_cache[: ks.shape[0]] = ks
        self.v_cache[: vs.shape[0]] = vs
        self.seq_lens[:bs] = seq_lens.to(torch.int32)
        self.cumsum_lengths[0] = 0
        self.cumsum_lengths[1 : bs + 1] = seq_lens.cumsum(0).to(torch.int32)
        self.use_varlen = max(lens) != min(lens)
        self.sliced_sequence_length = None if self.use_varlen else lens[0]
        self.current_batch_size = bs

    def get_used_cumsum_lengths(self):
        return self.cumsum_lengths[: self.current_batch_size + 1]


class PerLayerKVCache(nn.Module):
    """llama.py:173-346."""

    def __init__(self, max_unique_batch_size, max_unique_seq_length, max_shared_batch_sizes, max_shared_seq_lengths,
                 n_kv_heads, head_dim, device, dtype, arena: Optional[Tensor] = None):
        super().__init__()
        shape = (max_unique_batch_size, max_unique_seq_length, n_kv_heads, head_dim)
        # One allocation per layer (the reference's two attribute names, llama.py:186-198, stay as views of it): a sequence's K rows,
        # then its V rows (placement.kv_arena: the suffix pass streams that 1.5-3 % faster than all K, then all V).  WHERE the arena
        # sits in HBM matters too (profiles/r05_gqa_placement.md): `setup_caches` hands in arenas chosen by hydragen_amd/placement.py;
        # without one, a plain allocation of the same layout.
        if arena is None:
            arena = placement.kv_arena(shape, dtype, device, zero=True)
        assert tuple(arena.shape) == (2,) + shape and arena.dtype == dtype, f"{tuple(arena.shape)} {arena.dtype}"
        self.register_buffer("per_completion_k_cache", arena[0])
        self.register_buffer("per_completion_v_cache", arena[1])
        self.shared_caches = nn.ModuleList([
            SharedCache(b, s, n_kv_heads, head_dim, dtype, device)
            for b, s in zip(max_shared_batch_sizes, max_shared_seq_lengths)
        ])
        self.num_used_shared_caches = 0

    def empty_shared_cache(self):
        self.truncate_shared_caches(0)

    def get_num_total_shared_caches(self):
        return len(self.shared_caches)

    def truncate_shared_caches(self, n: int):
        assert n <= self.get_num_total_shared_caches(), f"{n} {self.get_num_total_shared_caches()}"
        self.num_used_shared_caches = n

    def update_per_completion_kvs(self, input_pos: Tensor, k_val: Tensor, v_val: Tensor):
        """input_pos [bs, sl]; k_val/v_val [bs, sl, h, d] -> scatter into the caches (llama.py:236-262)."""
        assert input_pos.shape[1] == k_val.shape[1], f"{input_pos.shape} {k_val.shape}"
        bs, sl, h, d = k_val.shape
        idx = input_pos[:, :, None, None].expand(bs, -1, h, d).to(torch.int64)
        self.per_completion_k_cache[:bs].scatter_(1, idx, k_val)
        self.per_completion_v_cache[:bs].scatter_(1, idx, v_val)
        return self.per_completion_k_cache[:bs], self.per_completion_v_cache[:bs]

    @torch.no_grad()
    def copy_shared_to_unique(self, total_num_sequences: int):
        """llama.py:264-298 (the no-sharing baseline materialises the prefix per sequence)."""
        assert self.num_used_shared_caches == 1, "Cannot copy shared without exactly one active shared cache"
        sc: SharedCache = self.shared_caches[0]
        sb = sc.get_current_batch_size()
        assert total_num_sequences % sb == 0
        rep = total_num_sequences // sb
        cu = sc.cumsum_lengths.tolist()
        for i in range(sb):
            n = cu[i + 1] - cu[i]
            self.per_completion_k_cache[i * rep : (i + 1) * rep, :n] = sc.k_cache[cu[i] : cu[i + 1]].unsqueeze(0)
            self.per_completion_v_cache[i * rep : (i + 1) * rep, :n] = sc.v_cache[cu[i] : cu[i + 1]].unsqueeze(0)

    @torch.no_grad()
    def repeat_per_completion_cache_for_num_samples(self, current_size: int, num_samples: int):
        if num_samples == 1:
            return
        self.per_completion_k_cache[: current_size * num_samples] = self.per_completion_k_cache[:current_size].repeat_interleave(num_samples, 0)
        self.per_completion_v_cache[: current_size * num_samples] = self.per_completion_v_cache[:current_size].repeat_interleave(num_samples, 0)

    def get_used_shared_caches(self) -> list[SharedCache]:
        return list(self.shared_caches)[: self.num_used_shared_caches]

    def get_shared_len(self, final_batch_size: int):
        """llama.py:317-330: per-sequence total shared length [B] (long)."""
        if self.num_used_shared_caches == 0:
            return torch.zeros((final_batch_size,), dtype=torch.long, device=self.per_completion_k_cache.device)
        lens = [c.seq_lens[: c.current_batch_size] for c in self.get_used_shared_caches()]
        return sum(repeat_to_batch_size(lens, final_batch_size)).long()

    def has_shared(self):
        return self.num_used_shared_caches > 0

    def append_shared(self, key_states, value_states, seq_lens):
        if self.num_used_shared_caches >= self.get_num_total_shared_caches():
            raise ValueError(f"No more available shared caches: {self.num_used_shared_caches} {self.get_num_total_shared_caches()}")
        self.shared_caches[self.num_used_shared_caches].fill(key_states, value_states, seq_lens)
        self.num_used_shared_caches += 1


class AttentionMode:
    SHARED_PREFILL = "shared-prefill"
    UNIQUE_PREFILL = "unique-prefill"
    DECODE = "decode"


def hydragen_attention_on_caches(q, k, v, shared_caches: list[SharedCache], seq_len: Optional[Tensor] = None):
    """llama.py:355-414: adapt the caches to the operator's arguments (views only, no copies)."""
    keys, values, cu_seqlens, max_seqlens, use_varlens = [], [], [], [], []
    for sc in shared_caches:
        if sc.use_varlen:
            keys.append(sc.k_cache)
            values.append(sc.v_cache)
            cu_seqlens.append(sc.get_used_cumsum_lengths())
            max_seqlens.append(sc.max_sequence_length)
        else:
            b, s = sc.get_current_batch_size(), sc.sliced_sequence_length
            keys.append(sc.k_cache[: b * s].view(b, s, *sc.k_cache.shape[1:]))
            values.append(sc.v_cache[: b * s].view(b, s, *sc.v_cache.shape[1:]))
            cu_seqlens.append(None)
            max_seqlens.append(None)
        use_varlens.append(sc.use_varlen)
    return hydragen_attention(q, k, v, shared_ks=keys, shared_vs=values, shared_cu_seq_lens=cu_seqlens,
                              shared_max_seq_lens=max_seqlens, use_varlens=use_varlens, seq_lens=seq_len)


class HydragenLlamaAttention(nn.Module):
    """llama.py:417-595."""

    def __init__(self, config: LlamaConfig):
        super().__init__()
        self.config = config
        self.hidden_size = config.hidden_size
        self.num_heads = config.num_attention_heads
        self.head_dim = self.hidden_size // self.num_heads
        self.num_key_value_heads = config.num_key_value_heads
        self.disable_hydragen = False
        self.disable_attention = False
        if self.head_dim * self.num_heads != self.hidden_size:
            raise ValueError(f"hidden_size must be divisible by num_heads (got `hidden_size`: {self.hidden_size} and `num_heads`: {self.num_heads}).")
        b = config.attention_bias
        self.q_proj = nn.Linear(self.hidden_size, self.num_heads * self.head_dim, bias=b)
        self.k_proj = nn.Linear(self.hidden_size, self.num_key_value_heads * self.head_dim, bias=b)
        self.v_proj = nn.Linear(self.hidden_size, self.num_key_value_heads * self.head_dim, bias=b)
        self.o_proj = nn.Linear(self.num_heads * self.head_dim, self.hidden_size, bias=b)
        self.kv_cache: Optional[PerLayerKVCache] = None
        self.mode: Optional[str] = None
        self.rotary_emb: Optional[RotaryTable] = None
        self.tp_reduce = False  # row-parallel o_proj needs the all-reduce of tp.py:108-112
        self.use_fused_decode = True
        self._qkv: Optional[Tensor] = None

    def forward(self, hidden_states: Tensor, position_ids: Tensor, shared_len: Optional[Tensor] = None):
        """shared_len: the per-sequence total shared length [B] (get_shared_len) when the caller already has it -- it is
        the same for every layer of a forward pass, and computing it is three small launches (~17 us) per layer."""
        bsz, q_len, _ = hidden_states.shape
        self._qkv = w = _fused_weight(self._qkv, (self.q_proj, self.k_proj, self.v_proj))
        if w is not None:
            nq, nk = self.q_proj.weight.shape[0], self.k_proj.weight.shape[0]
            q, k, v = nn.functional.linear(hidden_states, w).split([nq, nk, nk], dim=-1)
        else:
            q, k, v = self.q_proj(hidden_states), self.k_proj(hidden_states), self.v_proj(hidden_states)
        q = q.view(bsz, q_len, self.num_heads, self.head_dim)
        k = k.view(bsz, q_len, self.num_key_value_heads, self.head_dim)
        v = v.view(bsz, q_len, self.num_key_value_heads, self.head_dim)
        cos, sin = self.rotary_emb.cos_cached, self.rotary_emb.sin_cached

        fused = self.mode == AttentionMode.DECODE and self.use_fused_decode and q_len == 1
        if fused:
            # one HIP kernel: RoPE(q, k) at absolute positions, append k/v at (position - shared length),
            # seq_lens = that index + 1  (replaces llama.py:485-501,565-569 and the scatter_ of :236-262)
            from .fused_decode import rope_append_decode

            if self.disable_hydragen:
                shared_len = None
            elif shared_len is None:
                shared_len = self.kv_cache.get_shared_len(bsz)
            q, seq_lens = rope_append_decode(q, k, v, cos, sin, position_ids, shared_len,
                                             self.kv_cache.per_completion_k_cache, self.kv_cache.per_completion_v_cache)
            key_states = self.kv_cache.per_completion_k_cache[:bsz]
            value_states = self.kv_cache.per_completion_v_cache[:bsz]
            if self.disable_attention:
                # llama.py:433-437,503-504: attention replaced by the identity on the rotated q.  The same preamble kernel as
                # the other modes, so that this mode's step is the decode step minus exactly the attention kernels.
                attn_output = q
            elif not self.kv_cache.has_shared() or self.disable_hydragen:
                attn_output, _ = flash_attention_seqlen(q, key_states, value_states, seq_len=seq_lens)
            else:
                attn_output = hydragen_attention_on_caches(q, key_states, value_states,
                                                           self.kv_cache.get_used_shared_caches(), seq_len=seq_lens)
            out = self.o_proj(attn_output.reshape(bsz, q_len, -1))
            return all_reduce_sum(out) if self.tp_reduce else out

        if self.disable_hydragen:
            unique_position_ids = position_ids
        else:
            unique_position_ids = position_ids - self.kv_cache.get_shared_len(position_ids.shape[0]).unsqueeze(-1)
        q, k = apply_rotary_pos_emb(q, k, cos, sin, position_ids)

        if self.disable_attention:
            attn_output = q
        elif self.mode == AttentionMode.SHARED_PREFILL:
            if not self.kv_cache.has_shared():
                attn_output, _ = flash_attention(q, k, v, causal=True)
            else:
                attn_output = hydragen_attention_on_caches(q, k, v, self.kv_cache.get_used_shared_caches())
            self.kv_cache.append_shared(k, v, unique_position_ids.max(1).values + 1)
        elif self.mode == AttentionMode.UNIQUE_PREFILL:
            if self.disable_hydragen:
                ks, vs = self.kv_cache.update_per_completion_kvs(unique_position_ids, k, v)
                n = int(unique_position_ids.max().item()) + 1
                attn_output, _ = flash_attention(q, ks[:, :n], vs[:, :n], causal=True)
            else:
                if not self.kv_cache.has_shared():
                    attn_output, _ = flash_attention(q, k, v, causal=True)
                else:
                    attn_output = hydragen_attention_on_caches(q, k, v, self.kv_cache.get_used_shared_caches())
                self.kv_cache.update_per_completion_kvs(unique_position_ids, k, v)
        elif self.mode == AttentionMode.DECODE:
            ks, vs = self.kv_cache.update_per_completion_kvs(unique_position_ids, k, v)
            seq_lens = unique_position_ids.squeeze(-1) + 1
            if not self.kv_cache.has_shared() or self.disable_hydragen:
                attn_output, _ = flash_attention_seqlen(q, ks, vs, seq_len=seq_lens)
            else:
                attn_output = hydragen_attention_on_caches(q, ks, vs, self.kv_cache.get_used_shared_caches(), seq_len=seq_lens)
        else:
            raise ValueError(f"Unknown mode {self.mode}")

        out = self.o_proj(attn_output.reshape(bsz, q_len, -1))
        return all_reduce_sum(out) if self.tp_reduce else out


class HydragenLlamaDecoderLayer(nn.Module):
    def __init__(self, config: LlamaConfig):
        super().__init__()
        self.self_attn = HydragenLlamaAttention(config)
        self.mlp = LlamaMLP(config)
        self.input_layernorm = RMSNorm(config.hidden_size, eps=config.rms_norm_eps)
        self.post_attention_layernorm = RMSNorm(config.hidden_size, eps=config.rms_norm_eps)

    def forward(self, hidden_states, position_ids):
        hidden_states = hidden_states + self.self_attn(self.input_layernorm(hidden_states), position_ids)
        return hidden_states + self.mlp(self.post_attention_layernorm(hidden_states))


class HydragenLlamaModel(nn.Module):
    """llama.py:636-765."""

    def __init__(self, config: LlamaConfig):
        super().__init__()
        self.config = config
        self.embed_tokens = nn.Embedding(config.vocab_size, config.hidden_size, config.pad_token_id)
        self.layers = nn.ModuleList([HydragenLlamaDecoderLayer(config) for _ in range(config.num_hidden_layers)])
        self.norm = RMSNorm(config.hidden_size, eps=config.rms_norm_eps)
        self.rotary_emb = RotaryTable(config.hidden_size // config.num_attention_heads,
                                      config.max_position_embeddings, config.rope_theta)
        for layer in self.layers:
            layer.self_attn.rotary_emb = self.rotary_emb

    def set_disable_hydragen(self, disable=True):
        for layer in self.layers:
            layer.self_attn.disable_hydragen = disable

    def get_disable_hydragen(self):
        return self.layers[0].self_attn.disable_hydragen

    def set_disable_attention(self, disable=True):
        for layer in self.layers:
            layer.self_attn.disable_attention = disable

    def get_disable_attention(self):
        return self.layers[0].self_attn.disable_attention

    def copy_shared_cache_to_unique(self, total_num_sequences):
        for layer in self.layers:
            layer.self_attn.kv_cache.copy_shared_to_unique(total_num_sequences)

    def _used(self):
        return self.layers[0].self_attn.kv_cache.get_used_shared_caches()

    def get_shared_batch_sizes(self):
        return [c.get_current_batch_size() for c in self._used()]

    def get_shared_varlens(self):
        return [c.use_varlen for c in self._used()]

    def get_shared_slice_seq_lens(self):
        return [c.sliced_sequence_length for c in self._used()]

    def forward(self, input_ids, position_ids):
        h = self.embed_tokens(input_ids)
        layers = list(self.layers)
        w0 = self.norm.weight
        if not layers or not layer_ops.supported(h) or w0.dtype != h.dtype:
            for layer in layers:  # CPU / fp32 / odd widths: the spelled-out layer (llama.py:610-633)
                h = layer(h, position_ids=position_ids)
            return self.norm(h)
        # Same dataflow with every residual add fused into the norm that consumes it (llama.py:615-631: `residual +
        # hidden_states`, then the next LlamaRMSNorm): one kernel that reads the block output and the residual stream
        # once and writes the new stream and its normalised form, instead of an add and a norm launch per block.
        first = layers[0].input_layernorm
        _, normed = layer_ops.add_rms_norm(h, None, first.weight, first.variance_epsilon)
        # decode: the shared lengths (llama.py:317-330) once per forward instead of once per layer -- the layers' caches are
        # filled in lockstep, so every layer would compute the same [B] tensor with three small launches
        a0 = layers[0].self_attn
        shared_len = None
        if a0.mode == AttentionMode.DECODE and a0.kv_cache is not None and not a0.disable_hydragen and h.shape[1] == 1:
            shared_len = a0.kv_cache.get_shared_len(h.shape[0])
        for i, layer in enumerate(layers):
            post = layer.post_attention_layernorm
            h, normed = layer_ops.add_rms_norm(layer.self_attn(normed, position_ids, shared_len=shared_len), h, post.weight,
                                               post.variance_epsilon)
            nxt = layers[i + 1].input_layernorm if i + 1 < len(layers) else self.norm
            h, normed = layer_ops.add_rms_norm(layer.mlp(normed), h, nxt.weight, nxt.variance_epsilon)
        return normed


@dataclass
class CaptureData:
    graph: "torch.cuda.CUDAGraph"
    static_input_ids: Tensor
    static_position_ids: Tensor
    static_hidden: Tensor
    key: tuple


class GraphedHydragenLlamaModel(nn.Module):
    """HIP-graph replay of the decode forward; re-captures on the same invalidation keys as
    llama.py:791-823 (shapes, shared batch sizes, varlen flags, sliced lengths, disable flags)."""

    def __init__(self, model: HydragenLlamaModel):
        super().__init__()
        self.model = model
        self.capture_data: Optional[CaptureData] = None

    def invalidate(self):
        self.capture_data = None

    def _key(self, input_ids, position_ids):
        m = self.model
        order = _flash.current_seq_order()  # (its pointer is baked into the captured launches)
        return (tuple(input_ids.shape), tuple(position_ids.shape), tuple(m.get_shared_batch_sizes()),
                m.get_disable_hydragen(), m.get_disable_attention(), tuple(m.get_shared_varlens()),
                tuple(m.get_shared_slice_seq_lens()), None if order is None else (order.data_ptr(), order.numel()))

    def forward(self, input_ids, position_ids):
        key = self._key(input_ids, position_ids)
        if self.capture_data is not None and self.capture_data.key != key:
            self.invalidate()
        if self.capture_data is None:
            self.capture(input_ids, position_ids, key)
        self.capture_data.static_input_ids.copy_(input_ids)
        self.capture_data.static_position_ids.copy_(position_ids)
        self.capture_data.graph.replay()
        return self.capture_data.static_hidden

    def capture(self, input_ids, position_ids, key):
        static_input_ids = input_ids.clone()
        static_position_ids = position_ids.clone()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(3):  # warm-up on a side stream (llama.py:838-845)
                self.model(input_ids=static_input_ids, position_ids=static_position_ids)
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            static_hidden = self.model(input_ids=static_input_ids, position_ids=static_position_ids)
        self.capture_data = CaptureData(g, static_input_ids, static_position_ids, static_hidden, key)


class SharedCacheOp:
    WIPE = "wipe"
    EXTEND = "extend"
    PRESERVE = "preserve"


class HydragenLlamaForCausalLM(nn.Module):
    """llama.py:875-1422."""

    def __init__(self, config: LlamaConfig):
        super().__init__()
        self.config = config
        self.model = HydragenLlamaModel(config)
        self.vocab_size = config.vocab_size
        self.lm_head = nn.Linear(config.hidden_size, config.vocab_size, bias=False)
        self.kv_cache_allocated = False
        self.graphed_model: Optional[GraphedHydragenLlamaModel] = None
        self.mode: Optional[str] = None

    # ---- construction -----------------------------------------------------------------------------
    @classmethod
    def from_config(cls, config: LlamaConfig, dtype=torch.bfloat16, device="cuda", seed: int = 0, std: float = 0.02,
                    tp_shard: Optional[tuple] = None):
        """Random-weight model of the given architecture (synthetic benchmarks / tests).  `tp_shard=(rank, world)`
        builds only that rank's tensor-parallel shard (tp.py:115-132) without ever materialising the full model."""
        if tp_shard is not None:
            from copy import deepcopy
            from .tp import apply_tp

            with torch.device("meta"):
                model = cls(deepcopy(config))
            apply_tp(model.model, rank=tp_shard[0], world_size=tp_shard[1])
            model.to_empty(device=device)
            c = model.config
            model.model.rotary_emb = RotaryTable(c.hidden_size // c.num_attention_heads, c.max_position_embeddings,
                                                 c.rope_theta, device=device)
            for layer in model.model.layers:
                layer.self_attn.rotary_emb = model.model.rotary_emb
            with torch.no_grad():
                for prm in model.parameters():
                    if prm.ndim == 1:
                        prm.fill_(1.0)  # RMSNorm gains
        else:
            with torch.device(device):
                model = cls(config)
        model.to(dtype=dtype)
        model.model.rotary_emb.float()  # cos/sin tables stay fp32 (cast per use, like HF's rotary embedding)
        g = torch.Generator(device=device).manual_seed(seed)
        with torch.no_grad():
            for name, p in model.named_parameters():
                if p.ndim >= 2:
                    p.normal_(0.0, std, generator=g)
        model.device, model.dtype = torch.device(device), dtype
        return model

    @classmethod
    def from_pretrained(cls, model_name_or_path: str, **kwargs):
        """llama.py:1398-1422.  Needs `transformers` weights on disk; not exercised offline."""
        from transformers import LlamaForCausalLM  # pragma: no cover

        hf = LlamaForCausalLM.from_pretrained(model_name_or_path, **kwargs)  # pragma: no cover
        if hf.dtype not in (torch.float16, torch.bfloat16):  # pragma: no cover
            raise ValueError(f"Model must be in float16 or bfloat16, not {hf.dtype}")
        c = hf.config  # pragma: no cover
        rp = getattr(c, "rope_parameters", None) or {}
        cfg = LlamaConfig(c.hidden_size, c.intermediate_size, c.num_hidden_layers, c.num_attention_heads,
                          c.num_key_value_heads, c.vocab_size, c.rms_norm_eps,
                          getattr(c, "rope_theta", rp.get("rope_theta", 10000.0)), c.max_position_embeddings,
                          getattr(c, "attention_bias", False), c.pad_token_id)  # pragma: no cover
        scaling = getattr(c, "rope_scaling", None) or {}  # pragma: no cover
        kind = scaling.get("rope_type", scaling.get("type", "default")) or rp.get("rope_type", "default")  # pragma: no cover
        if kind not in ("default", None):  # pragma: no cover
            raise NotImplementedError(f"rope scaling '{kind}' is not implemented (plain theta-only RoPE tables)")
        model = cls(cfg)  # pragma: no cover
        report = model.load_state_dict(hf.state_dict(), strict=False)  # pragma: no cover
        extra = [k for k in report.unexpected_keys if "inv_freq" not in k]  # pragma: no cover
        if report.missing_keys or extra:  # pragma: no cover
            raise RuntimeError(f"checkpoint does not match: missing {report.missing_keys[:5]}, unexpected {extra[:5]}")
        model.to(device=hf.device, dtype=hf.dtype)  # pragma: no cover
        model.model.rotary_emb.float()  # pragma: no cover
        model.device, model.dtype = hf.device, hf.dtype  # pragma: no cover
        return model  # pragma: no cover

    # ---- cache / graph management -------------------------------------------------------------------
    def set_mode(self, mode):
        self.mode = mode
        for layer in self.model.layers:
            layer.self_attn.mode = mode

    def graph(self, do_graph: bool = True):
        """Controls whether decoding replays a HIP graph (llama.py:898-907)."""
        if do_graph:
            if self.graphed_model is None:
                self.graphed_model = GraphedHydragenLlamaModel(self.model)
        else:
            self.graphed_model = None

    def maybe_invalidate(self):
        if self.graphed_model is not None:
            self.graphed_model.invalidate()

    def get_num_heads(self):
        return self.config.num_attention_heads

    def setup_caches(self, max_unique_batch_size: int, max_unique_seq_length: int,
                     max_shared_batch_sizes: list[int], max_shared_seq_lengths: list[int]):
        """Allocate the unique KV cache and the shared cache levels at every layer (llama.py:921-955)."""
        self.maybe_invalidate()
        max_unique_seq_length = (max_unique_seq_length + 15) // 16 * 16
        head_dim = self.config.hidden_size // self.get_num_heads()
        device, dtype = self.lm_head.weight.device, self.lm_head.weight.dtype
        # the layers' unique caches, placed where the suffix pass streams them fastest (placement.py: more candidates than layers
        # are allocated, each timed once with the suffix pass over all of its keys, the slowest go back; off the data path)
        self.kv_cache_allocated = False
        for layer in self.model.layers:
            layer.self_attn.kv_cache = None  # a second setup_caches: the old arenas are free before the candidates are made
        arenas, self.kv_placement = placement.place_kv_arenas(
            len(self.model.layers), (max_unique_batch_size, max_unique_seq_length, self.config.num_key_value_heads, head_dim),
            dtype, device, self.config.num_attention_heads)
        for layer, arena in zip(self.model.layers, arenas):
            layer.self_attn.kv_cache = PerLayerKVCache(
                max_unique_batch_size=max_unique_batch_size, max_unique_seq_length=max_unique_seq_length,
                max_shared_batch_sizes=max_shared_batch_sizes, max_shared_seq_lengths=max_shared_seq_lengths,
                n_kv_heads=self.config.num_key_value_heads, head_dim=head_dim, device=device, dtype=dtype, arena=arena)
        # the decode loop's schedule hint (flash.seq_order): one buffer for the model's lifetime, so that a captured decode graph
        # keeps pointing at the current generation's order
        self.seq_order_buf = torch.arange(max_unique_batch_size, dtype=torch.int32, device=device)
        self.kv_cache_allocated = True

    def empty_shared_cache(self):
        for layer in self.model.layers:
            layer.self_attn.kv_cache.empty_shared_cache()

    def truncate_shared_caches(self, new_num_shared_caches: int):
        for layer in self.model.layers:
            layer.self_attn.kv_cache.truncate_shared_caches(new_num_shared_caches)

    def get_shared_cache_len(self, batch_size):
        return self.model.layers[0].self_attn.kv_cache.get_shared_len(batch_size)

    def get_num_used_shared_caches(self):
        return self.model.layers[0].self_attn.kv_cache.num_used_shared_caches

    def repeat_per_completion_cache_for_num_samples(self, current_size, num_samples):
        for layer in self.model.layers:
            layer.self_attn.kv_cache.repeat_per_completion_cache_for_num_samples(current_size, num_samples)

    # ---- forward -------------------------------------------------------------------------------------
    def forward(self, input_ids, position_ids, seq_lens=None, use_graph=False, full_logits=False, raw_logits=False):
        """Logits in fp32 as the reference returns them (llama.py:1063-1070); raw_logits=True keeps the lm_head's own
        dtype (the decode loop's fused sampler reads them once, without the fp32 copy)."""
        model = self.graphed_model if use_graph else self.model
        assert model is not None
        hidden = model(input_ids=input_ids, position_ids=position_ids)
        if full_logits:
            to_lm_head = hidden
        elif seq_lens is not None:
            to_lm_head = hidden[torch.arange(hidden.shape[0], device=hidden.device), seq_lens - 1].unsqueeze(1)
        else:
            to_lm_head = hidden[:, -1:]
        logits = self.lm_head(to_lm_head)
        return logits if raw_logits else logits.float()

    def apply_top_p(self, logits, top_p, min_tokens_to_keep=1, filter_value=-float("Inf")):
        sorted_logits, sorted_indices = torch.sort(logits, descending=False)
        cumulative_probs = sorted_logits.softmax(dim=-1).cumsum(dim=-1)
        remove = cumulative_probs <= (1 - top_p)
        remove[..., -min_tokens_to_keep:] = 0
        return logits.masked_fill(remove.scatter(1, sorted_indices, remove), filter_value)

    def sample_from_logits(self, logits, temperature, num_samples=1, top_p=None):
        if top_p is not None:
            logits = self.apply_top_p(logits, top_p)
        if logits.is_cuda and logits.ndim == 2 and num_samples == 1 and logits.stride(-1) == 1 and temperature >= 0:
            # one HIP kernel: argmax(logits / T + Gumbel noise) draws exactly from softmax(logits / T) -- what the softmax +
            # torch.multinomial chain below draws, in one pass over the logits instead of ~12 launches over [B, vocab]
            return layer_ops.sample_tokens(logits, temperature)
        if temperature == 0:
            assert logits.ndim == 2
            return logits.argmax(dim=-1, keepdim=True).repeat_interleave(num_samples, dim=-1)
        probs = nn.functional.softmax(logits / temperature, dim=-1)
        return torch.multinomial(probs, num_samples=num_samples, replacement=True)

    def _positions(self, input_ids):
        shared_lens = self.get_shared_cache_len(input_ids.shape[0])
        n = input_ids.shape[1]
        return shared_lens[:, None] + torch.arange(n, device=input_ids.device, dtype=torch.long)[None, :]

    @torch.no_grad()
    def append_shared(self, input_ids, seq_lens=None, full_logits=False):
        """Add a new level of shared cache (llama.py:1073-1116).  Padding on the right."""
        self.set_mode(AttentionMode.SHARED_PREFILL)
        position_ids = self._positions(input_ids)
        if seq_lens is not None:
            last = position_ids.gather(1, (seq_lens.long() - 1)[:, None])
            ar = torch.arange(input_ids.shape[1], device=input_ids.device)[None, :]
            position_ids = torch.where(ar >= seq_lens.long()[:, None], last, position_ids)
        return self(input_ids=input_ids, position_ids=position_ids, seq_lens=seq_lens, full_logits=full_logits)

    @torch.no_grad()
    def process_unique(self, input_ids, seq_lens=None):
        """Prefill of per-sequence (unique) prompts (llama.py:1118-1145)."""
        self.set_mode(AttentionMode.UNIQUE_PREFILL)
        return self(input_ids=input_ids, position_ids=self._positions(input_ids), seq_lens=seq_lens)

    # ---- generation -----------------------------------------------------------------------------------
    # Contract (README.md:183-289 of the reference; llama.py:1156-1396): prompts form a hierarchy, outermost level
    # first.  With num_return_sequences > 1 every given level is shared and the completions hang off the last one;
    # otherwise the last given level is each sequence's own (unique) prompt.  shared_cache_op: "wipe" empties the
    # shared caches first, "preserve" drops the levels this call added when it returns, "extend" keeps them.
    def _prompt_levels(self, input_ids, seq_lens, fan_out, flatten_last):
        """-> (shared levels [(ids, lens)], unique level (ids, lens) or None)."""
        levels = [] if input_ids is None else ([input_ids] if isinstance(input_ids, Tensor) else list(input_ids))
        if seq_lens is None:
            lens = [torch.full((x.shape[0],), x.shape[1], device=x.device, dtype=torch.long) for x in levels]
        else:
            lens = [seq_lens] if isinstance(seq_lens, Tensor) else list(seq_lens)
        pairs = list(zip(levels, lens))
        if not pairs or (fan_out and not flatten_last):
            return pairs, None
        return pairs[:-1], pairs[-1]

    def _check_room(self, start_positions, new_tokens):
        """The fused decode preamble cannot raise from the device: refuse on the host anything that would index past
        the unique cache or the rotary tables (the reference's scatter_/index ops device-assert)."""
        if new_tokens < 2:
            return  # the first token comes from the prefill logits: no decode step runs
        room = self.model.layers[0].self_attn.kv_cache.per_completion_k_cache.shape[1]
        last_pos = int(start_positions.max().item()) + new_tokens - 2
        # cache index = position - shared length, PER SEQUENCE (llama.py:487-492 of the reference): with ragged shared
        # levels the longest position and the shortest shared length belong to different sequences
        starts = start_positions.reshape(start_positions.shape[0], -1)[:, 0]
        if self.model.get_disable_hydragen():
            first_idx = starts
        else:
            first_idx = starts - self.get_shared_cache_len(start_positions.shape[0]).to(starts.device)
        last_idx = int(first_idx.max().item()) + new_tokens - 2
        if last_idx >= room:
            raise ValueError(f"unique cache holds {room} tokens per sequence, decoding needs {last_idx + 1}")
        if last_pos >= self.config.max_position_embeddings:
            raise ValueError(f"position {last_pos} exceeds max_position_embeddings = {self.config.max_position_embeddings}")

    @torch.no_grad()
    def generate(self, input_ids: Optional[Union[Tensor, list[Tensor]]] = None,
                 seq_lens: Optional[Union[Tensor, list[Tensor]]] = None, starting_logits: Optional[Tensor] = None,
                 num_return_sequences: int = 1, max_new_tokens: int = 5, temperature: float = 1.0,
                 top_p: Optional[float] = None, eos_token_id: Optional[int] = None, return_logits: bool = False,
                 shared_cache_op: str = SharedCacheOp.PRESERVE, disable_hydragen: bool = False,
                 disable_attention: bool = False, disable_hierarchy: bool = False,
                 token_overrides: Optional[Tensor] = None):
        if not self.kv_cache_allocated:
            raise RuntimeError("call setup_caches() before generate()")
        if (input_ids is None) == (starting_logits is None):
            raise ValueError("pass exactly one of input_ids and starting_logits")
        if temperature < 0:
            raise ValueError(f"temperature must be non-negative, {temperature} is invalid")
        fan_out = num_return_sequences > 1
        flatten = disable_hierarchy or disable_hydragen  # the baselines keep the last level per sequence
        if shared_cache_op == SharedCacheOp.WIPE:
            self.empty_shared_cache()
        levels_before = self.get_num_used_shared_caches()
        shared, unique = self._prompt_levels(input_ids, seq_lens, fan_out, flatten)
        n_given = len(shared) + (unique is not None)
        depth = levels_before + n_given + (1 if fan_out else 0)
        if disable_hydragen and (depth != 2 or (n_given == 2 and shared[0][0].shape[0] != 1)):
            raise ValueError("disable_hydragen compares against ONE shared prompt: exactly two levels, the first of batch 1")
        if disable_hierarchy and not (depth == 3 and fan_out):
            raise ValueError("disable_hierarchy flattens a three-level hierarchy with num_return_sequences > 1")

        self.model.set_disable_attention(bool(disable_attention))
        logits = None if starting_logits is None else starting_logits.unsqueeze(1)
        for ids, lens in shared:
            logits = self.append_shared(ids, lens)
        leaf_batch = (unique[0].shape[0] if unique is not None else logits.shape[0])
        batch = leaf_batch * num_return_sequences
        if disable_hydragen:
            self.model.set_disable_hydragen(True)
            if self.get_num_used_shared_caches() > 0:
                self.model.copy_shared_cache_to_unique(batch)
        if unique is not None:
            logits = self.process_unique(*unique)
            self.repeat_per_completion_cache_for_num_samples(unique[0].shape[0], num_return_sequences)

        try:
            return self._decode(logits[:, -1], unique, num_return_sequences, max_new_tokens, temperature, top_p,
                                eos_token_id, return_logits, token_overrides)
        finally:
            if shared_cache_op == SharedCacheOp.PRESERVE:
                self.truncate_shared_caches(levels_before)
            self.model.set_disable_hydragen(False)
            self.model.set_disable_attention(False)

    def _decode(self, prefill_logits, unique, fan, max_new_tokens, temperature, top_p, eos_token_id, return_logits,
                token_overrides):
        first = self.sample_from_logits(prefill_logits, temperature=temperature, num_samples=fan, top_p=top_p).reshape(-1, 1)
        kept_logits = [prefill_logits.repeat_interleave(fan, 0)] if return_logits else None
        start = self.get_shared_cache_len(first.shape[0])[:, None]
        if unique is not None:
            start = start + unique[1].long().repeat_interleave(fan, 0)[:, None]
        self._check_room(start, max_new_tokens)
        done = (first == eos_token_id) if eos_token_id is not None else None
        tokens = [first]
        feed = first if token_overrides is None else token_overrides[:, 0:1]
        self.set_mode(AttentionMode.DECODE)
        graphed = self.graphed_model is not None
        # Ragged unique prompts: hand the longest sequences to the chip first.  Every length grows by one per step, so the order of
        # this generation's first step is the order of all of them (C2 heads, lengths 1..128 at random: suffix pass 184 -> 169 us).
        order = None
        if unique is not None:
            lens0 = unique[1].repeat_interleave(fan, 0)
            if self.schedule_longest_first and lens0.numel() > 1 and bool((lens0 != lens0[0]).any()):
                order = self.seq_order_buf[: lens0.numel()]
                order.copy_(_flash.longest_first(lens0))
        with _flash.seq_order(order, check=False):
            return self._decode_steps(feed, start, tokens, kept_logits, done, graphed, max_new_tokens, temperature, top_p,
                                      eos_token_id, return_logits, token_overrides)

    schedule_longest_first = True  # (tests switch it off to compare: only the schedule may depend on it, never a token)

    def _decode_steps(self, feed, start, tokens, kept_logits, done, graphed, max_new_tokens, temperature, top_p, eos_token_id,
                      return_logits, token_overrides):
        for step in range(max_new_tokens - 1):
            # 16-bit logits straight into the sampler unless the caller wants them (fp32, as the reference returns them)
            logits = self(input_ids=feed, position_ids=start + step, use_graph=graphed,
                          raw_logits=not return_logits and top_p is None)[:, -1]
            if return_logits:
                kept_logits.append(logits)
            nxt = self.sample_from_logits(logits, temperature=temperature, top_p=top_p)
            if done is not None:
                done = done | (nxt == eos_token_id)
                if bool(done.all()):
                    break
            tokens.append(nxt)
            feed = nxt if token_overrides is None else token_overrides[:, step + 1 : step + 2]
        out = torch.cat(tokens, dim=-1)
        check_collectives()  # no-op without the direct xGMI all-reduce; raises if a rank ever gave up on a peer
        return (out, kept_logits) if return_logits else out
