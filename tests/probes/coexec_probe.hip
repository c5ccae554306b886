// Hardware probe: do MFMA work of one wave and VALU work of ANOTHER wave on the same SIMD overlap?
// 512-thread workgroups, 1 per CU: waves 0-3 = one per SIMD, waves 4-7 = their SIMD partners.
// mode bit0: waves 0-3 run an MFMA loop, bit1: waves 4-7 run a VALU (exp2 + fma) loop,
// bit2: waves 4-7 run the MFMA loop too, bit3: waves 0-3 ALSO run the VALU loop interleaved in-stream.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ void mfma_loop(int n, float* out, int lane) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(lane * 0.001f + i); b[i] = (__bf16)(1.0f - lane * 0.002f); }
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    for (int i = 0; i < n; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
    }
    out[lane] = c0[0] + c1[1] + c2[2] + c3[3];
}
template <int KIND>  // 0: exp2 + fma mix, 1: fma only, 2: exp2 only
__device__ __forceinline__ void valu_loop(int n, float* out, int lane) {
    float x[8];
    for (int j = 0; j < 8; ++j) x[j] = lane * 0.01f + j;
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (KIND == 0) x[j] = __builtin_amdgcn_exp2f(x[j] * 0.5f - 1.0f) + x[j] * 0.25f;
            if (KIND == 1) { x[j] = x[j] * 0.5f - 1.0f; x[j] = x[j] * 0.25f + 3.0f; x[j] = x[j] * 1.5f - 0.5f; }
            if (KIND == 2) x[j] = __builtin_amdgcn_exp2f(x[j]);
        }
    }
    float s = 0; for (int j = 0; j < 8; ++j) s += x[j];
    out[lane] = s;
}
// same wave: one MFMA followed by independent VALU work in its shadow
template <int NV>
__device__ __forceinline__ void mixed_loop(int n, float* out, int lane) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(lane * 0.001f + i); b[i] = (__bf16)(1.0f - lane * 0.002f); }
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    float x[NV];
    for (int j = 0; j < NV; ++j) x[j] = lane * 0.01f + j;
    for (int i = 0; i < n; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NV; ++j) x[j] = x[j] * 0.5f - 1.0f;
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NV; ++j) x[j] = x[j] * 0.25f + 3.0f;
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NV; ++j) x[j] = x[j] * 1.5f - 0.5f;
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NV; ++j) x[j] = x[j] * 0.75f + 0.5f;
    }
    float s = c0[0] + c1[1] + c2[2] + c3[3];
    for (int j = 0; j < NV; ++j) s += x[j];
    out[lane] = s;
}
__global__ __launch_bounds__(512) void probe(float* out, int mode, int nm, int nv) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    float* o = out + (blockIdx.x * 8 + wave) * 64;
    if (wave < 4) {
        if (mode & 1) mfma_loop(nm, o, lane);
        if (mode == 64) mixed_loop<4>(nm, o, lane);
        if (mode == 65) mixed_loop<8>(nm, o, lane);
        if (mode == 66) mixed_loop<12>(nm, o, lane);
    } else {
        if (mode & 2) valu_loop<0>(nv, o, lane);
        if (mode & 4) mfma_loop(nm, o, lane);
        if (mode & 8) valu_loop<1>(nv, o, lane);
        if (mode & 16) valu_loop<2>(nv, o, lane);
    }
}
int main() {
    float* d; hipMalloc(&d, 256 * 8 * 64 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int nm = 4000, nv = 4000;   // 16000 MFMAs ~ 512k cycles; 32000 exp+64000 valu
    for (int mode : {1, 2, 3, 8, 9, 16, 17, 64, 65, 66}) {
        hipLaunchKernelGGL(probe, dim3(256), dim3(512), 0, 0, d, mode, nm, nv);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(probe, dim3(256), dim3(512), 0, 0, d, mode, nm, nv);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("mode %2d (%s%s%s%s%s%s): %.1f us\n", mode, (mode & 1) && mode < 64 ? "A:mfma " : "", mode & 2 ? "B:exp+fma " : "", mode & 4 ? "B:mfma " : "", mode & 8 ? "B:fma " : "", mode & 16 ? "B:exp " : "", mode >= 64 ? "A: mfma with 4/8/12 in-stream fma per mfma" : "", ms * 1e3);
    }
    return 0;
}
