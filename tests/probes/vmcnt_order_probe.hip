// Hardware probe 6: does s_waitcnt vmcnt(N) order an LDS-DMA (buffer_load ... lds) against a later ordinary
// buffer_load into VGPRs?  Each wave issues a DMA from a COLD address (fresh 1 KiB per wave, never touched before)
// followed by a VGPR load from a HOT address (the same line re-read by everybody), waits vmcnt(1) -- "everything
// but the newest operation has completed" -- and then reads the DMA's LDS destination.  If the counter retires in
// issue order the LDS holds the cold data; a stale sentinel means the younger VGPR load retired first and the
// counted wait let the wave through early.  (The mirror experiment swaps the two roles.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(const unsigned* cold, const unsigned* hot, unsigned* bad, int mirror) {
    __shared__ unsigned lds[4 * 256];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    unsigned* my = lds + wave * 256;
    for (int i = lane; i < 256; i += 64) my[i] = 0xdeadbeefu;
    __syncthreads();
    const uint64_t cb = (uint64_t)(cold + ((size_t)blockIdx.x * 4 + wave) * 256), hb = (uint64_t)hot;
    u32x4 rc = {(unsigned)cb, (unsigned)(cb >> 32) & 0xffffu, 1024u, 0x00020000u};
    u32x4 rh = {(unsigned)hb, (unsigned)(hb >> 32) & 0xffffu, 1024u, 0x00020000u};
    for (int i = 0; i < 4; ++i) { rc[i] = __builtin_amdgcn_readfirstlane(rc[i]); rh[i] = __builtin_amdgcn_readfirstlane(rh[i]); }
    const unsigned ldsaddr = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned*)my);
    const unsigned voff = lane * 16;
    u32x4 r;
    unsigned got;
    if (!mirror) {
        asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\t"
                     "buffer_load_dwordx4 %2, %4, 0 offen lds\n\t"   // older: DMA, cold
                     "buffer_load_dwordx4 %0, %2, %5, 0 offen\n\t"  // younger: VGPR load, hot
                     "s_waitcnt vmcnt(1)\n\t"
                     "ds_read_b32 %1, %6\n\t"
                     "s_waitcnt vmcnt(0) lgkmcnt(0)"
                     : "=&v"(r), "=&v"(got)
                     : "v"(voff), "s"(ldsaddr), "s"(rc), "s"(rh), "v"(ldsaddr + lane * 16)
                     : "memory");
        const unsigned want = cold[((size_t)blockIdx.x * 4 + wave) * 256 + lane * 4];
        if (got != want) atomicAdd(bad, 1u);
    } else {
        // older: VGPR load, cold; younger: DMA, hot; after vmcnt(1) the VGPR must hold the cold data -- always true by
        // construction (the register is only read after the wait), so the mirror only checks the DMA afterwards
        asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\t"
                     "buffer_load_dwordx4 %0, %2, %4, 0 offen\n\t"
                     "buffer_load_dwordx4 %2, %5, 0 offen lds\n\t"
                     "s_waitcnt vmcnt(0)\n\t"
                     "ds_read_b32 %1, %6\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     : "=&v"(r), "=&v"(got)
                     : "v"(voff), "s"(ldsaddr), "s"(rc), "s"(rh), "v"(ldsaddr + lane * 16)
                     : "memory");
        if (got != hot[lane * 4]) atomicAdd(bad, 1u);
    }
    if (r[0] == 0x12345678u) bad[1] = 1;  // keep r alive
}
int main() {
    const int nblk = 4096;
    unsigned *cold, *hot, *bad;
    hipMalloc(&cold, (size_t)nblk * 4 * 1024 * 8); hipMalloc(&hot, 4096); hipMalloc(&bad, 8);
    unsigned* h = (unsigned*)malloc((size_t)nblk * 4 * 1024);
    for (size_t i = 0; i < (size_t)nblk * 1024; ++i) h[i] = (unsigned)(i * 2654435761u) | 1u;
    for (int rep = 0; rep < 8; ++rep) {  // a fresh (cold) region every repetition
        unsigned* c = cold + (size_t)rep * nblk * 1024;
        hipMemcpy(c, h, (size_t)nblk * 4 * 1024, hipMemcpyHostToDevice);
        hipMemcpy(hot, h, 4096, hipMemcpyHostToDevice);
        hipMemset(bad, 0, 8);
        hipLaunchKernelGGL(probe, dim3(nblk), dim3(256), 0, 0, c, hot, bad, rep & 1);
        unsigned b[2]; hipMemcpy(b, bad, 8, hipMemcpyDeviceToHost);
        printf("rep %d (%s): %u of %d lanes read stale / wrong LDS data\n", rep, rep & 1 ? "control: vmcnt(0)" : "DMA(cold) then VGPR load(hot), vmcnt(1)", b[0], nblk * 256);
    }
    return 0;
}
