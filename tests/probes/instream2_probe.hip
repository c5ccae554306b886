// Hardware probe 4: in-stream capacity of the MFMA shadow for (a) v_exp_f32 only, (b) v_fma_f32 only,
// (c) v_pk_fma_f32 only.  2 waves per SIMD, each: loop { MFMA; K independent VALU ops } x 4 accumulators.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int KIND, int K>
__global__ __launch_bounds__(512) void probe(float* out, int n) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(lane * 0.001f + i); b[i] = (__bf16)(1.0f - lane * 0.002f); }
    f32x16 c0 = {0}, c1 = {0};
    float x[16];
    f32x2 y[8];
    for (int j = 0; j < 16; ++j) x[j] = lane * 0.01f + j;
    for (int j = 0; j < 8; ++j) y[j] = f32x2{lane * 0.01f + j, 1.0f + j};
    const f32x2 s2 = {0.999f, 1.001f}, t2 = {0.001f, -0.001f};
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (u & 1) c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
            else c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
#pragma unroll
            for (int e = 0; e < K; ++e) {
                const int j = (u * K + e) & 15;
                if (KIND == 0) x[j] = __builtin_amdgcn_exp2f(x[j]);
                if (KIND == 1) x[j] = __builtin_fmaf(x[j], 0.999f, 0.001f);
                if (KIND == 2) y[j & 7] = y[j & 7] * s2 + t2;
                if (KIND == 3) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(x[j]) : "v"(x[j]), "v"(x[(j + 1) & 15]));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = c0[0] + c1[1];
    for (int j = 0; j < 16; ++j) s += x[j];
    for (int j = 0; j < 8; ++j) s += y[j][0] + y[j][1];
    out[(blockIdx.x * 8 + wave) * 64 + lane] = s;
}
template <int KIND, int K>
void run(float* d, int n) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<KIND, K>), dim3(256), dim3(512), 0, 0, d, n);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<KIND, K>), dim3(256), dim3(512), 0, 0, d, n);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const char* names[] = {"exp2", "fma", "pk_fma", "cvt_pk_bf16"};
    printf("%-12s K=%d per MFMA: %.2f ns per MFMA (2 waves/SIMD)\n", names[KIND], K, ms * 1e6 / (4.0 * n) / 2);
}
int main() {
    float* d; hipMalloc(&d, 256 * 8 * 64 * 4);
    const int n = 3000;
    run<0, 0>(d, n);
    run<0, 1>(d, n); run<0, 2>(d, n); run<0, 3>(d, n); run<0, 4>(d, n);
    run<1, 2>(d, n); run<1, 4>(d, n); run<1, 6>(d, n); run<1, 8>(d, n); run<1, 12>(d, n);
    run<2, 2>(d, n); run<2, 4>(d, n); run<2, 6>(d, n); run<2, 8>(d, n);
    run<3, 2>(d, n); run<3, 4>(d, n); run<3, 8>(d, n);
    return 0;
}
