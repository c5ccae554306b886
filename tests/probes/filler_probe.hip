// Hardware probe 9: how many independent single-issue instructions hide in the shadow of one v_mfma_f32_32x32x16_bf16
// when ONE wave runs per SIMD (256-thread workgroups, one per CU), and what each instruction beyond that costs.
// Variants: N plain v_fma_f32 per MFMA; N v_exp_f32; MFMA destination in AGPRs vs VGPRs; 1 LDS-DMA per 4 MFMAs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NF, int KIND>  // KIND 0: fma fillers, acc in AGPR; 1: exp fillers; 2: fma fillers, MFMA dst in VGPRs; 3: fma + one DMA per 4 MFMAs
__global__ __launch_bounds__(256) void probe(const u32x4* src, float* out, unsigned long long* cyc, int iters) {
    __shared__ __attribute__((aligned(1024))) char lds[16384];
    u32x4 a = src[threadIdx.x], b = src[threadIdx.x + 256];
    float f[8];
    for (int i = 0; i < 8; ++i) f[i] = 1.0f + threadIdx.x * 1e-6f * (i + 1);
    f32x16 s0, s1, s2, s3;
    for (int i = 0; i < 16; ++i) s0[i] = s1[i] = s2[i] = s3[i] = 0.f;
    asm volatile("v_accvgpr_write_b32 a0, 0" ::: "a0","a15","a16","a31","a32","a47","a48","a63");
    const uint64_t base = (uint64_t)src;
    u32x4 rsrc;
    rsrc[0] = __builtin_amdgcn_readfirstlane((unsigned)base);
    rsrc[1] = __builtin_amdgcn_readfirstlane((unsigned)(base >> 32) & 0xffffu);
    rsrc[2] = 1u << 20;
    rsrc[3] = 0x00020000u;
    const unsigned ldsaddr = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds + (threadIdx.x >> 6) * 4096);
    const unsigned voff = (threadIdx.x & 63) * 16;
    const unsigned ldsv = ldsaddr + voff;
    u32x4 stage = {0u, 0u, 0u, 0u}, r0 = stage;
    unsigned long long r1 = 0, r2 = 0;
    __syncthreads();
    uint64_t t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            if (KIND == 2) {
                if ((m & 3) == 0) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(s0) : "v"(a), "v"(b));
                if ((m & 3) == 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(s1) : "v"(a), "v"(b));
                if ((m & 3) == 2) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(s2) : "v"(a), "v"(b));
                if ((m & 3) == 3) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(s3) : "v"(a), "v"(b));
            } else {
                if ((m & 3) == 0) asm volatile("v_mfma_f32_32x32x16_bf16 a[0:15], %0, %1, a[0:15]" ::"v"(a), "v"(b));
                if ((m & 3) == 1) asm volatile("v_mfma_f32_32x32x16_bf16 a[16:31], %0, %1, a[16:31]" ::"v"(a), "v"(b));
                if ((m & 3) == 2) asm volatile("v_mfma_f32_32x32x16_bf16 a[32:47], %0, %1, a[32:47]" ::"v"(a), "v"(b));
                if ((m & 3) == 3) asm volatile("v_mfma_f32_32x32x16_bf16 a[48:63], %0, %1, a[48:63]" ::"v"(a), "v"(b));
            }
#pragma unroll
            for (int k = 0; k < NF; ++k) {
                if (KIND == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(f[k & 7]));
                else asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f[k & 7]) : "v"(f[(k + 3) & 7]));
            }
            if (KIND == 4 && (m & 3) == 1) {  // register staging: one 16-B global load per lane ...
                asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(stage) : "v"(voff), "s"(rsrc) : "memory");
            }
            if (KIND == 4 && (m & 3) == 3) {  // ... and one ds_write_b128 of the data loaded an iteration earlier
                asm volatile("s_waitcnt vmcnt(1)\n\tds_write_b128 %0, %1" ::"v"(ldsv), "v"(stage) : "memory");
            }
            if (KIND == 5 && (m & 1) == 1) {  // LDS fragment reads: one ds_read_b128 + two ds_read_b64_tr_b16 per 2 MFMAs
                asm volatile("ds_read_b128 %0, %3\n\tds_read_b64_tr_b16 %1, %3 offset:2048\n\tds_read_b64_tr_b16 %2, %3 offset:4096"
                             : "=v"(r0), "=v"(r1), "=v"(r2) : "v"(ldsv) : "memory");
            }
            if (KIND == 3 && (m & 3) == 3)
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" ::"v"(voff), "s"(rsrc), "s"(ldsaddr) : "memory");
        }
        if (KIND == 3) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        if (KIND == 5) asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(r0), "+v"(r1), "+v"(r2));
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15\n\ts_nop 7\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    float acc = 0.f;
    for (int i = 0; i < 8; ++i) acc += f[i];
    acc += s0[0] + s1[1] + s2[2] + s3[3] + (float)stage[0] + (float)r0[1] + (float)r1 + (float)r2;
    out[blockIdx.x * 256 + threadIdx.x] = acc;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int NF, int KIND>
void run(const u32x4* s, float* o, unsigned long long* c, const char* what) {
    const int iters = 2000;
    hipLaunchKernelGGL((probe<NF, KIND>), dim3(256), dim3(256), 0, 0, s, o, c, iters);
    hipLaunchKernelGGL((probe<NF, KIND>), dim3(256), dim3(256), 0, 0, s, o, c, iters);
    unsigned long long h;
    hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
    printf("%-28s fillers/MFMA %2d : %7.1f ticks per MFMA\n", what, NF, (double)h / (iters * 8.0));
}

int main() {
    u32x4* s; float* o; unsigned long long* c;
    hipMalloc(&s, 1 << 20); hipMalloc(&o, 256 * 256 * 4); hipMalloc(&c, 8);
    hipMemset(s, 0x3c, 1 << 20);
    run<0, 0>(s, o, c, "fma, acc AGPR"); run<2, 0>(s, o, c, "fma, acc AGPR"); run<4, 0>(s, o, c, "fma, acc AGPR");
    run<5, 0>(s, o, c, "fma, acc AGPR"); run<6, 0>(s, o, c, "fma, acc AGPR"); run<8, 0>(s, o, c, "fma, acc AGPR");
    run<12, 0>(s, o, c, "fma, acc AGPR"); run<16, 0>(s, o, c, "fma, acc AGPR");
    run<1, 1>(s, o, c, "exp"); run<2, 1>(s, o, c, "exp"); run<4, 1>(s, o, c, "exp"); run<8, 1>(s, o, c, "exp");
    run<0, 2>(s, o, c, "fma, MFMA dst VGPR"); run<4, 2>(s, o, c, "fma, MFMA dst VGPR"); run<8, 2>(s, o, c, "fma, MFMA dst VGPR");
    run<0, 3>(s, o, c, "fma + 1 DMA / 4 MFMA"); run<4, 3>(s, o, c, "fma + 1 DMA / 4 MFMA"); run<8, 3>(s, o, c, "fma + 1 DMA / 4 MFMA");
    run<0, 4>(s, o, c, "fma + reg-staged 1KB / 4 MFMA"); run<4, 4>(s, o, c, "fma + reg-staged 1KB / 4 MFMA"); run<8, 4>(s, o, c, "fma + reg-staged 1KB / 4 MFMA");
    run<0, 5>(s, o, c, "fma + LDS frag reads"); run<4, 5>(s, o, c, "fma + LDS frag reads"); run<6, 5>(s, o, c, "fma + LDS frag reads");
    return 0;
}
