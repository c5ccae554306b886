// Hardware probe 3: how much softmax-like VALU work (fma + exp2 + add + max [+ cvt_pk]) hides in the shadow of
// one v_mfma_f32_32x32x16_bf16 when it is interleaved IN THE SAME WAVE's instruction stream?
// E = softmax elements per MFMA (the pipelined prefix kernel needs 1 element per MFMA: 16 MFMAs per 32x32 tile).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int E, int WAVES>
__global__ __launch_bounds__(512) void probe(float* out, int n) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (wave >= WAVES) return;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(lane * 0.001f + i); b[i] = (__bf16)(1.0f - lane * 0.002f); }
    f32x16 c0 = {0}, c1 = {0};
    float x[8], mx = -1e30f, sm = 0.f;
    unsigned pk = 0;
    for (int j = 0; j < 8; ++j) x[j] = lane * 0.01f + j;
    const float sc = 0.125f;
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (u & 1) c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
            else c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int j = (u * E + e) & 7;
                mx = fmaxf(mx, x[j]);
                float p = __builtin_amdgcn_exp2f(__builtin_fmaf(x[j], sc, -mx));
                sm += p;
                x[j] = p + (float)j;
                if (e & 1) { pk ^= __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(p, sm)); }
            }
        }
    }
    out[(blockIdx.x * 8 + wave) * 64 + lane] = c0[0] + c1[1] + sm + mx + (float)pk;
}
template <int E, int WAVES>
void run(float* d, int n) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<E, WAVES>), dim3(256), dim3(512), 0, 0, d, n);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<E, WAVES>), dim3(256), dim3(512), 0, 0, d, n);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("E=%d waves/SIMD=%d: %.1f us  -> %.2f ns per MFMA per wave-slot\n", E, WAVES / 4, ms * 1e3, ms * 1e6 / (4.0 * n) / (WAVES / 4));
}
int main() {
    float* d; hipMalloc(&d, 256 * 8 * 64 * 4);
    const int n = 4000;
    run<0, 4>(d, n); run<1, 4>(d, n); run<2, 4>(d, n); run<3, 4>(d, n); run<4, 4>(d, n);
    run<0, 8>(d, n); run<1, 8>(d, n); run<2, 8>(d, n); run<3, 8>(d, n); run<4, 8>(d, n);
    return 0;
}
