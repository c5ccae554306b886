// Hardware probe 2: can VALU work of wave B run in the shadow of wave A's MFMAs (same SIMD) when
//  (a) A leaves issue gaps (s_nop) after each MFMA, (b) A chains one accumulator, (c) B has higher s_setprio?
// 512-thread workgroups, 1 per CU: waves 0-3 = A (one per SIMD), waves 4-7 = B (their SIMD partners).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NOPS, int CHAIN>
__device__ __forceinline__ void mfma_loop(int n, float* out, int lane) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(lane * 0.001f + i); b[i] = (__bf16)(1.0f - lane * 0.002f); }
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    for (int i = 0; i < n; ++i) {
#define GAP() do { if (NOPS >= 1) asm volatile("s_nop 7"); if (NOPS >= 2) asm volatile("s_nop 7"); if (NOPS >= 3) asm volatile("s_nop 7"); } while (0)
        if (CHAIN) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0); GAP();
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0); GAP();
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0); GAP();
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0); GAP();
        } else {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0); GAP();
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0); GAP();
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0); GAP();
            c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0); GAP();
        }
    }
    out[lane] = c0[0] + c1[1] + c2[2] + c3[3];
}
template <int KIND>  // 0: exp2 + fma mix, 1: fma only
__device__ __forceinline__ void valu_loop(int n, float* out, int lane) {
    float x[8];
    for (int j = 0; j < 8; ++j) x[j] = lane * 0.01f + j;
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (KIND == 0) x[j] = __builtin_amdgcn_exp2f(x[j] * 0.5f - 1.0f) + x[j] * 0.25f;
            if (KIND == 1) { x[j] = x[j] * 0.5f - 1.0f; x[j] = x[j] * 0.25f + 3.0f; x[j] = x[j] * 1.5f - 0.5f; }
        }
    }
    float s = 0; for (int j = 0; j < 8; ++j) s += x[j];
    out[lane] = s;
}
// A: 0 none, 1 plain, 2 nop1, 3 nop2, 4 nop3, 5 chain, 6 chain+nop1   B: 0 none, 1 exp+fma, 2 fma   prio: B's s_setprio
__global__ __launch_bounds__(512) void probe(float* out, int A, int B, int prioA, int prioB, int nm, int nv) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    float* o = out + (blockIdx.x * 8 + wave) * 64;
    if (wave < 4) {
        if (prioA == 3) asm volatile("s_setprio 3");
        if (A == 1) mfma_loop<0, 0>(nm, o, lane);
        if (A == 2) mfma_loop<1, 0>(nm, o, lane);
        if (A == 3) mfma_loop<2, 0>(nm, o, lane);
        if (A == 4) mfma_loop<3, 0>(nm, o, lane);
        if (A == 5) mfma_loop<0, 1>(nm, o, lane);
        if (A == 6) mfma_loop<1, 1>(nm, o, lane);
    } else {
        if (prioB == 3) asm volatile("s_setprio 3");
        if (B == 1) valu_loop<0>(nv, o, lane);
        if (B == 2) valu_loop<1>(nv, o, lane);
    }
}
int main() {
    float* d; hipMalloc(&d, 256 * 8 * 64 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int nm = 4000, nv = 4000;
    for (int B : {0, 1, 2}) for (int A : {0, 1, 2, 3, 4, 5, 6}) for (int pr : {0, 1, 2}) {
        if (A == 0 && B == 0) continue;
        if ((A == 0 || B == 0) && pr) continue;
        int pa = pr == 2 ? 3 : 0, pb = pr == 1 ? 3 : 0;
        hipLaunchKernelGGL(probe, dim3(256), dim3(512), 0, 0, d, A, B, pa, pb, nm, nv);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(probe, dim3(256), dim3(512), 0, 0, d, A, B, pa, pb, nm, nv);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("A=%d B=%d prioA=%d prioB=%d: %.1f us\n", A, B, pa, pb, ms * 1e3);
    }
    return 0;
}
