// Hardware probe: which waves of a 512-thread workgroup share a SIMD (HW_REG_HW_ID.SIMD_ID)?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(512) void probe(unsigned* out) {
    extern __shared__ char smem[];
    unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);  // HW_REG_HW_ID
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = hw;
    if (threadIdx.x == 9999) smem[0] = 1;
}
int main() {
    unsigned* d; hipMalloc(&d, 4 * 8 * 64);
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 133120);
    hipLaunchKernelGGL(probe, dim3(16), dim3(512), 133120, 0, d);
    unsigned h[8 * 16]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int b = 0; b < 16; b += 5) {
        printf("block %2d:", b);
        for (int w = 0; w < 8; ++w) printf("  w%d: simd %u wave_slot %u cu %u", w, (h[b * 8 + w] >> 4) & 3, h[b * 8 + w] & 15, (h[b * 8 + w] >> 8) & 15);
        printf("\n");
    }
    return 0;
}
