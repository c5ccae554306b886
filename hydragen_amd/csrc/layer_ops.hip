// Elementwise glue of the decoder layer around the attention block (the model shell), HBM-bound:
//   * add_rmsnorm_kernel: h = residual + x (rounded to the storage dtype, as the reference's bf16 add at
//     /root/reference/hydragen/llama.py:624,631 rounds it) and normed = RMSNorm(h) * weight (transformers' LlamaRMSNorm
//     as used at llama.py:605-608,656: fp32 statistics, one rounding) in ONE pass over the row: the torch form is an add
//     kernel (read 2, write 1) followed by a norm kernel (read 1, write 1); fused it is read 2, write 2 and one launch.
//   * swiglu_kernel: silu(gate) * up (transformers' LlamaMLP act_fn(gate_proj(x)) * up_proj(x), imported at
//     llama.py:2,604), gate and up being the two column halves of ONE fused GEMM output (row-strided views): fp32 maths,
//     one rounding, one launch instead of a strided silu and a strided multiply.
//   * sample_kernel: one token per row from softmax(logits / temperature) (llama.py `sample_from_logits`: softmax +
//     torch.multinomial, ~12 launches over the [B, vocab] fp32 matrix per decode step) by the Gumbel-max identity
//     argmax_v(logits_v / T + g_v), g_v = -ln(-ln u_v): one pass over the 16-bit logits, counter-based Philox4x32-10
//     noise keyed by (seed, call offset, row, column) -- reproducible for a seed, independent of the launch geometry.
#include "hyd_kernels.h"

namespace hyd {

namespace {

template <typename T>
__device__ __forceinline__ void unpack8(const u32x4& u, float (&f)[8]) {
    using TR = Traits<T>;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f[2 * i] = TR::lo(u[i]);
        f[2 * i + 1] = TR::hi(u[i]);
    }
}
template <typename T>
__device__ __forceinline__ u32x4 pack8(const float (&f)[8]) {
    using TR = Traits<T>;
    u32x4 u;
#pragma unroll
    for (int i = 0; i < 4; ++i) u[i] = TR::pack2(f[2 * i], f[2 * i + 1]);
    return u;
}
__device__ __forceinline__ float wave_sum(float x) {
    x = group_sum<32>(x);
    return pair_sum(x);
}

}  // namespace

// One workgroup of 256 threads per row; a thread keeps its NV 16-byte chunks (8 elements each, chunk c of the row at
// c = threadIdx.x + 256 j) of the summed row in registers between the statistic and the scaling, so the row is read
// once.  n % 8 == 0, n <= 256 * 8 * NV.
template <typename T, int NV>
__global__ __launch_bounds__(256) void add_rmsnorm_kernel(const NormArgs a) {
    __shared__ float part[4];
    const int64_t row = blockIdx.x;
    const int nchunk = a.n >> 3;
    const uint16_t* x = static_cast<const uint16_t*>(a.x) + row * a.x_rs;
    const uint16_t* r = a.residual ? static_cast<const uint16_t*>(a.residual) + row * a.r_rs : nullptr;
    const uint16_t* w = static_cast<const uint16_t*>(a.weight);
    uint16_t* so = a.sum_out ? static_cast<uint16_t*>(a.sum_out) + row * a.s_rs : nullptr;
    uint16_t* no = static_cast<uint16_t*>(a.norm_out) + row * a.o_rs;
    u32x4 hx[NV], hr[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int c = threadIdx.x + 256 * j;
        hx[j] = u32x4{0u, 0u, 0u, 0u};
        hr[j] = u32x4{0u, 0u, 0u, 0u};
        if (c < nchunk) {
            hx[j] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(x + 8 * c));
            if (r) hr[j] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(r + 8 * c));
        }
    }
    float h[NV][8];
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int c = threadIdx.x + 256 * j;
        float fx[8], fr[8];
        unpack8<T>(hx[j], fx);
        unpack8<T>(hr[j], fr);
        if (r) {
            // the residual stream is stored in the 16-bit dtype: the statistic sees the ROUNDED sum, like a separate norm kernel would
#pragma unroll
            for (int i = 0; i < 8; ++i) fx[i] += fr[i];
            const u32x4 s = pack8<T>(fx);
            if (so && c < nchunk) *reinterpret_cast<u32x4*>(so + 8 * c) = s;
            unpack8<T>(s, fx);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            h[j][i] = fx[i];
            ss += fx[i] * fx[i];
        }
    }
    ss = wave_sum(ss);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = ss;
    __syncthreads();
    const float tot = part[0] + part[1] + part[2] + part[3];
    const float inv = rsqrtf(tot / (float)a.n + a.eps);
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int c = threadIdx.x + 256 * j;
        if (c < nchunk) {
            float fw[8], o[8];
            unpack8<T>(*reinterpret_cast<const u32x4*>(w + 8 * c), fw);
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = h[j][i] * inv * fw[i];
            *reinterpret_cast<u32x4*>(no + 8 * c) = pack8<T>(o);
        }
    }
}

// One thread per 8 output elements.
template <typename T>
__global__ __launch_bounds__(256) void swiglu_kernel(const SwigluArgs a) {
    const int nchunk = a.n >> 3;
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= a.rows * nchunk) return;
    const int64_t row = gid / nchunk;
    const int c = (int)(gid - row * nchunk);
    const u32x4 g = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(static_cast<const uint16_t*>(a.gate) + row * a.g_rs + 8 * c));
    const u32x4 u = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(static_cast<const uint16_t*>(a.up) + row * a.u_rs + 8 * c));
    float fg[8], fu[8], o[8];
    unpack8<T>(g, fg);
    unpack8<T>(u, fu);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float sig = __builtin_amdgcn_rcpf(1.0f + fast_exp2(-kLog2e * fg[i]));
        o[i] = fg[i] * sig * fu[i];
    }
    *reinterpret_cast<u32x4*>(static_cast<uint16_t*>(a.out) + row * a.o_rs + 8 * c) = pack8<T>(o);
}

namespace {

__device__ __forceinline__ void philox4x32_10(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t lo0 = 0xD2511F53u * c[0], hi0 = __umulhi(0xD2511F53u, c[0]);
        const uint32_t lo1 = 0xCD9E8D57u * c[2], hi1 = __umulhi(0xCD9E8D57u, c[2]);
        c[0] = hi1 ^ c[1] ^ k0;
        c[1] = lo1;
        c[2] = hi0 ^ c[3] ^ k1;
        c[3] = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
}
// standard Gumbel noise from 23 random bits: u = (k + 1/2) 2^-23 for k in [0, 2^23), strictly inside (0, 1) and exact in
// fp32 (24 significant bits).  (The former 24-bit form (k 2^-24 + 2^-25) rounded to exactly 1.0 for k = 2^24 - 1 -- a tie
// that rounds to even -- where -ln u = 0 and the noise is +inf: that column won the argmax whatever its logit, about two
// tokens per decode step at a batch of 1024 and a vocabulary of 32000.)  g = -ln(-ln u), finite for every k -- also on
// hardware whose approximate log2 returns 0 for the largest u (1 - 2^-24, true -ln u = 2^-24): -ln u is held at >= 2^-25.
__device__ __forceinline__ float gumbel(uint32_t bits) {
    const float u = ((float)(bits >> 9) + 0.5f) * 0x1p-23f;
    const float e = fmaxf(-kLn2 * fast_log2(u), 0x1p-25f);
    return -kLn2 * fast_log2(e);
}

}  // namespace

// One workgroup of 256 threads per row; thread t walks the 8-element chunks t, t + 256, ...; ties go to the lowest index.
template <int DT>
__global__ __launch_bounds__(256) void sample_kernel(const SampleArgs a) {
    __shared__ float bv[4];
    __shared__ int bi[4];
    const int row = blockIdx.x;
    const int nchunk = (a.n + 7) >> 3;
    const char* base = static_cast<const char*>(a.logits) + (int64_t)row * a.row_stride * (DT == HYD_F32 ? 4 : 2);
    float best = -INFINITY;
    int besti = 0x7fffffff;
    for (int c = threadIdx.x; c < nchunk; c += 256) {
        float f[8];
        const bool full = 8 * c + 8 <= a.n;
        if (DT == HYD_F32) {
#pragma unroll
            for (int i = 0; i < 8; ++i) f[i] = (full || 8 * c + i < a.n) ? reinterpret_cast<const float*>(base)[8 * c + i] : -INFINITY;
        } else if (full && a.vec_ok) {
            const u32x4 u = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(base) + c);
            if (DT == HYD_F16) unpack8<F16>(u, f);
            else unpack8<BF16>(u, f);
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const uint32_t h = 8 * c + i < a.n ? reinterpret_cast<const uint16_t*>(base)[8 * c + i] : (DT == HYD_F16 ? 0xfc00u : 0xff80u);
                f[i] = DT == HYD_F16 ? Traits<F16>::lo(h) : Traits<BF16>::lo(h);
            }
        }
        if (a.inv_temperature > 0.f) {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                uint32_t ctr[4] = {(uint32_t)(2 * c + k), (uint32_t)row, (uint32_t)a.offset, (uint32_t)(a.offset >> 32)};
                philox4x32_10(ctr, (uint32_t)a.seed, (uint32_t)(a.seed >> 32));
#pragma unroll
                for (int i = 0; i < 4; ++i) f[4 * k + i] = f[4 * k + i] * a.inv_temperature + gumbel(ctr[i]);
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (f[i] > best) {  // strictly greater: the lowest index of equal keys wins inside a thread (indices ascend)
                best = f[i];
                besti = 8 * c + i;
            }
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const float ov = __shfl_xor(best, off);
        const int oi = __shfl_xor(besti, off);
        if (ov > best || (ov == best && oi < besti)) {
            best = ov;
            besti = oi;
        }
    }
    if ((threadIdx.x & 63) == 0) {
        bv[threadIdx.x >> 6] = best;
        bi[threadIdx.x >> 6] = besti;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            if (bv[w] > best || (bv[w] == best && bi[w] < besti)) {
                best = bv[w];
                besti = bi[w];
            }
        }
        a.out[row] = besti == 0x7fffffff ? 0 : besti;  // a row of NaNs / -inf: token 0
    }
}

int launch_sample(const SampleArgs& a, int dtype, hipStream_t s) {
    if (a.rows == 0) return 0;
    const dim3 grid((unsigned)a.rows), block(256);
    if (dtype == HYD_F16) hipLaunchKernelGGL((sample_kernel<HYD_F16>), grid, block, 0, s, a);
    else if (dtype == HYD_BF16) hipLaunchKernelGGL((sample_kernel<HYD_BF16>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((sample_kernel<HYD_F32>), grid, block, 0, s, a);
    return (int)hipGetLastError();
}

int launch_add_rmsnorm(const NormArgs& a, int dtype, hipStream_t s) {
    if (a.rows == 0) return 0;
    const int nv = (a.n + 2047) / 2048;
    const dim3 grid((unsigned)a.rows), block(256);
#define HYD_NORM(TT, NV) hipLaunchKernelGGL((add_rmsnorm_kernel<TT, NV>), grid, block, 0, s, a)
#define HYD_NORM_T(TT)                                    \
    switch (nv) {                                         \
        case 1: HYD_NORM(TT, 1); break;                   \
        case 2: HYD_NORM(TT, 2); break;                   \
        case 3: case 4: HYD_NORM(TT, 4); break;           \
        case 5: case 6: case 7: case 8: HYD_NORM(TT, 8); break; \
        default: return (int)hipErrorInvalidValue;        \
    }
    if (dtype == HYD_F16) { HYD_NORM_T(F16) } else { HYD_NORM_T(BF16) }
#undef HYD_NORM_T
#undef HYD_NORM
    return (int)hipGetLastError();
}

int launch_swiglu(const SwigluArgs& a, int dtype, hipStream_t s) {
    const int64_t threads = a.rows * (a.n >> 3);
    if (threads == 0) return 0;
    const dim3 grid((unsigned)((threads + 255) / 256)), block(256);
    if (dtype == HYD_F16) hipLaunchKernelGGL((swiglu_kernel<F16>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((swiglu_kernel<BF16>), grid, block, 0, s, a);
    return (int)hipGetLastError();
}

}  // namespace hyd
