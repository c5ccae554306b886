// Probe: does an MFMA + VALU kernel that leaves half of every SIMD's registers free co-run with the HBM-bound suffix
// pass, and what does each lose?  Built as a shared object with one C entry point (tools/_corun_exp.py launches it on a
// second stream beside hyd_decode_attn_fused(phase = UNIQUE)).  256 workgroups x 4 waves, <= 256 registers per wave,
// 128 KB of dynamic LDS (one workgroup per CU, like the prefix pass), no global traffic inside the loop.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

template <int BIG>
__global__ __launch_bounds__(256) void corun_kernel(float* out, int iters, int nvalu) {
    extern __shared__ char smem[];
    // allocate a wave of a given register count, like a prefix-pass variant of that size would
    if constexpr (BIG == 1) asm volatile("" ::: "v250");
    if constexpr (BIG == 2) asm volatile("" ::: "v180");
    if constexpr (BIG == 3) asm volatile("" ::: "v120");
    if constexpr (BIG == 4) asm volatile("" ::: "v90");
    f32x16_t acc0 = {0}, acc1 = {0};
    bf16x8_t a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)(float)(i - 3); }
    float x0 = threadIdx.x, x1 = 1.f, x2 = 2.f, x3 = 3.f;
    volatile float* l = reinterpret_cast<volatile float*>(smem);
    l[threadIdx.x] = x0;
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
#pragma unroll
        for (int k = 0; k < 3; ++k) { x0 = __builtin_fmaf(x0, 1.0001f, x1); x1 = __builtin_fmaf(x1, 0.9999f, x2); x2 = __builtin_amdgcn_exp2f(x2 * 0.001f); }
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
#pragma unroll
        for (int k = 0; k < 3; ++k) { x3 = __builtin_fmaf(x3, 1.0001f, x0); x1 = __builtin_fmaf(x1, 0.9999f, x3); x2 = __builtin_amdgcn_exp2f(x2 * 0.001f); }
        if (nvalu) x0 += l[(threadIdx.x + it) & 255];
    }
    float s = x0 + x1 + x2 + x3;
    for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
    if (s == 12345.678f) out[0] = s;
}

extern "C" int corun_launch(void* out, int iters, int nvalu, void* stream) {
    static bool once = (hipFuncSetAttribute(reinterpret_cast<const void*>(corun_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) == hipSuccess);
    (void)once;
    hipLaunchKernelGGL(corun_kernel<1>, dim3(256), dim3(256), 128 * 1024, (hipStream_t)stream, (float*)out, iters, nvalu);
    return (int)hipGetLastError();
}
// the small variant (~50 registers, 16 KB of LDS): two of these fit on a CU many times over -- the control experiment
extern "C" int corun_launch_small(void* out, int iters, int nvalu, void* stream) {
    hipLaunchKernelGGL(corun_kernel<0>, dim3(256), dim3(256), 16 * 1024, (hipStream_t)stream, (float*)out, iters, nvalu);
    return (int)hipGetLastError();
}

// any combination: big = 0 / 4 / 3 / 2 / 1 -> ~50 / 91 / 121 / 181 / 251 registers per wave; lds = bytes of dynamic LDS per workgroup
extern "C" int corun_launch_cfg(void* out, int iters, int nvalu, void* stream, int big, int lds) {
    static bool once = (hipFuncSetAttribute(reinterpret_cast<const void*>(corun_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) == hipSuccess) &&
                       (hipFuncSetAttribute(reinterpret_cast<const void*>(corun_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) == hipSuccess);
    (void)once;
    switch (big) {
#define CORUN(N) case N: hipLaunchKernelGGL(corun_kernel<N>, dim3(256), dim3(256), lds, (hipStream_t)stream, (float*)out, iters, nvalu); break;
        CORUN(1) CORUN(2) CORUN(3) CORUN(4)
        default: hipLaunchKernelGGL(corun_kernel<0>, dim3(256), dim3(256), lds, (hipStream_t)stream, (float*)out, iters, nvalu);
#undef CORUN
    }
    return (int)hipGetLastError();
}
