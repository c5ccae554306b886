// Hardware probe 5: buffer_load_dwordx4 ... lds (LDS-DMA through a buffer resource) on gfx950:
// destination layout (M0 base + lane * 16?), and what out-of-range lanes (offset >= num_records) do to LDS.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(const unsigned* src, unsigned* out, unsigned num_records, unsigned soff) {
    __shared__ unsigned lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = 0xdeadbeefu;
    __syncthreads();
    const uint64_t base = (uint64_t)src;
    u32x4 rsrc;
    rsrc[0] = (unsigned)base;
    rsrc[1] = (unsigned)(base >> 32) & 0xffffu;
    rsrc[2] = num_records;
    rsrc[3] = 0x00020000u;
    rsrc[0] = __builtin_amdgcn_readfirstlane(rsrc[0]);
    rsrc[1] = __builtin_amdgcn_readfirstlane(rsrc[1]);
    rsrc[2] = __builtin_amdgcn_readfirstlane(rsrc[2]);
    rsrc[3] = __builtin_amdgcn_readfirstlane(rsrc[3]);
    const unsigned ldsaddr = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned*)lds + 256);
    const unsigned voff = (63 - threadIdx.x) * 16;  // reversed lanes: shows the LDS image is lane-linear
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds\n\ts_waitcnt vmcnt(0)"
                 :: "v"(voff), "s"(rsrc), "s"(ldsaddr), "s"(soff) : "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += 64) out[i] = lds[i];
}
int main() {
    unsigned *s, *o, h[1024], ho[1024];
    for (int i = 0; i < 1024; ++i) h[i] = i;
    hipMalloc(&s, 4096); hipMalloc(&o, 4096);
    hipMemcpy(s, h, 4096, hipMemcpyHostToDevice);
    for (unsigned nr : {4096u, 512u, 0u}) for (unsigned so : {0u, 256u}) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, s, o, nr, so);
        hipMemcpy(ho, o, 4096, hipMemcpyDeviceToHost);
        printf("num_records=%u soffset=%u\n", nr, so);
        for (int i = 56; i < 336; i += 4) {
            if ((i - 56) % 32 == 0) printf("  lds[%3d..]:", i);
            printf(" %x", ho[i]);
            if ((i - 56) % 32 == 28) printf("\n");
        }
        printf("\n");
    }
    return 0;
}
