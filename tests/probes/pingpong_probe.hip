// Hardware probe (round 4): does the guide's 8-wave role-alternating attention body (two waves per SIMD: one in an
// MFMA + softmax-VALU "compute" segment while its partner is in an LDS-read / LDS-DMA "load" segment, s_barrier between
// segments) run the matrix pipe faster than one wave per SIMD that carries MFMA, VALU, LDS reads and DMA in one stream?
// Synthetic bodies with the instruction mix of the prefix pass at D = 128 (per MFMA: fma, exp2, add, 1/2 pack, 1/2 max3;
// LDS fragment reads: per 64-key tile 16 ds_read_b128 (K) + 32 ds_read_b64_tr_b16 (V) per wave, feeding 32 MFMAs of a
// 32-row wave or 64 MFMAs of a 64-row wave; LDS-DMA: DPS KiB per wave per 16 MFMAs).
//   MODE 0: 8 waves, ping-pong: [K reads | QK 16 MFMAs | V reads | PV 16 MFMAs], partner shifted by one segment, 4 barriers per tile
//   MODE 1: MODE 0 + static s_setprio 1 for waves 4-7
//   MODE 2: 4 waves (one per SIMD), everything in one stream, 64 MFMAs per tile per wave, one barrier per tile
//   MODE 3: 8 waves, everything in each wave's own stream (32 MFMAs per tile per wave), one barrier per tile
//   MODE 4: MODE 0 with 32-MFMA segments (QK + PV of a tile back to back), 2 barriers per tile
//   MODE 5: MODE 0 without the softmax VALU (what the matrix pipe does under the barriers alone)
// Prints shader cycles per MFMA per SIMD (32 = the matrix pipe's floor) and the wall-clock TFLOP/s equivalent.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// the fragment block lives in fixed registers v[64:127] (kf[I] = v[64 + 4 I : 67 + 4 I]) so that the two 8-byte transpose
// reads of a V fragment land in the halves of the MFMA operand without compiler copies
#define KR0 "{v[64:67]}"
#define KL0 "{v[64:65]}"
#define KH0 "{v[66:67]}"
#define KR1 "{v[68:71]}"
#define KL1 "{v[68:69]}"
#define KH1 "{v[70:71]}"
#define KR2 "{v[72:75]}"
#define KL2 "{v[72:73]}"
#define KH2 "{v[74:75]}"
#define KR3 "{v[76:79]}"
#define KL3 "{v[76:77]}"
#define KH3 "{v[78:79]}"
#define KR4 "{v[80:83]}"
#define KL4 "{v[80:81]}"
#define KH4 "{v[82:83]}"
#define KR5 "{v[84:87]}"
#define KL5 "{v[84:85]}"
#define KH5 "{v[86:87]}"
#define KR6 "{v[88:91]}"
#define KL6 "{v[88:89]}"
#define KH6 "{v[90:91]}"
#define KR7 "{v[92:95]}"
#define KL7 "{v[92:93]}"
#define KH7 "{v[94:95]}"
#define KR8 "{v[96:99]}"
#define KL8 "{v[96:97]}"
#define KH8 "{v[98:99]}"
#define KR9 "{v[100:103]}"
#define KL9 "{v[100:101]}"
#define KH9 "{v[102:103]}"
#define KR10 "{v[104:107]}"
#define KL10 "{v[104:105]}"
#define KH10 "{v[106:107]}"
#define KR11 "{v[108:111]}"
#define KL11 "{v[108:109]}"
#define KH11 "{v[110:111]}"
#define KR12 "{v[112:115]}"
#define KL12 "{v[112:113]}"
#define KH12 "{v[114:115]}"
#define KR13 "{v[116:119]}"
#define KL13 "{v[116:117]}"
#define KH13 "{v[118:119]}"
#define KR14 "{v[120:123]}"
#define KL14 "{v[120:121]}"
#define KH14 "{v[122:123]}"
#define KR15 "{v[124:127]}"
#define KL15 "{v[124:125]}"
#define KH15 "{v[126:127]}"
#define MFMA(acc, I, b) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : KR##I(kf[I]), "v"(b))
#define FMA(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(c1), "v"(c2))
#define EXP(x) asm volatile("v_exp_f32 %0, %0" : "+v"(x))
#define ADD(x) asm volatile("v_add_f32 %0, %0, %1" : "+v"(sum) : "v"(x))
#define PACK(d, x, y) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(d) : "v"(x), "v"(y))
#define MAX3(x, y, z) asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(tm) : "v"(x), "v"(y), "v"(z))
#define LDS128(I, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=" KR##I(kf[I]) : "v"(laddr), "n"(off))
#define LDSTR2(I, OFF) { u32x2 lo_, hi_; asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=" KL##I(lo_) : "v"(laddr8), "n"(OFF)); asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=" KH##I(hi_) : "v"(laddr8), "n"((OFF) + 512)); kf[I] = u32x4{lo_[0], lo_[1], hi_[0], hi_[1]}; }
#define DMA(off) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen offset:%3 lds" ::"v"(voff), "s"(rsrc), "s"(ldsdst), "n"(off) : "memory")
#define WAITL(N) asm volatile("s_waitcnt lgkmcnt(%16)" : "+" KR0(kf[0]), "+" KR1(kf[1]), "+" KR2(kf[2]), "+" KR3(kf[3]), "+" KR4(kf[4]), "+" KR5(kf[5]), "+" KR6(kf[6]), "+" KR7(kf[7]), "+" KR8(kf[8]), "+" KR9(kf[9]), "+" KR10(kf[10]), "+" KR11(kf[11]), "+" KR12(kf[12]), "+" KR13(kf[13]), "+" KR14(kf[14]), "+" KR15(kf[15]) : "n"(N))
#define WAITV(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")
#define BAR() __builtin_amdgcn_s_barrier()

template <int MODE, int DPS>  // DPS: LDS-DMA KiB per wave per 16 MFMAs (2: 256 rows per workgroup, 4: 128 rows)
__global__ __launch_bounds__(MODE == 2 ? 256 : 512) void probe(const u32x4* src, float* out, unsigned long long* cyc, int tiles) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    constexpr bool SM = MODE != 5;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    u32x4 q = src[tid & 255];
    u32x4 kf[16];
    f32x16 acc[4];
    float s[16], sum = 0.f, tm = 0.f, c1 = 1e-4f, c2 = -1.0f;
    unsigned p[8];
#pragma unroll
    for (int i = 0; i < 16; ++i) { s[i] = -1.f - lane * 1e-3f; kf[i] = src[(tid + 64 * i) & 1023]; }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) p[i] = 0;
    const uint64_t base = (uint64_t)src;
    u32x4 rsrc;
    rsrc[0] = __builtin_amdgcn_readfirstlane((unsigned)base);
    rsrc[1] = __builtin_amdgcn_readfirstlane((unsigned)(base >> 32) & 0xffffu);
    rsrc[2] = 1u << 20;
    rsrc[3] = 0x00020000u;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
    const unsigned laddr = lds0 + lane * 16;
    const unsigned laddr8 = lds0 + lane * 8;
    const unsigned ldsdst = __builtin_amdgcn_readfirstlane(lds0 + 32768 + wave * 4096);
    const unsigned voff = lane * 16 + wave * 8192;
    for (int i = tid; i < 16384; i += blockDim.x) reinterpret_cast<unsigned*>(lds)[i] = 0x3c003c00u;
    __syncthreads();

// one MFMA + its share of the softmax: 4 VALU (one of them an exp2); J = position inside a 16-group
#define CGROUP(J, I)                                                                       \
    MFMA(acc[(J) & 3], I, q);                                                              \
    if (SM) {                                                                               \
        FMA(s[(J)]);                                                                        \
        EXP(s[((J) + 15) & 15]);                                                            \
        ADD(s[((J) + 14) & 15]);                                                            \
        if ((J) & 1) MAX3(s[((J) + 5) & 15], s[((J) + 6) & 15], s[((J) + 7) & 15]);         \
        else PACK(p[(J) >> 1], s[((J) + 12) & 15], s[((J) + 13) & 15]);                     \
    }
#define COMPUTE16()                                                                                       \
    CGROUP(0, 0) CGROUP(1, 1) CGROUP(2, 2) CGROUP(3, 3) CGROUP(4, 4) CGROUP(5, 5) \
    CGROUP(6, 6) CGROUP(7, 7) CGROUP(8, 8) CGROUP(9, 9) CGROUP(10, 10) CGROUP(11, 11) \
    CGROUP(12, 12) CGROUP(13, 13) CGROUP(14, 14) CGROUP(15, 15)
#define DMAS_A() { if (DPS >= 1) DMA(0); if (DPS >= 4) DMA(1024); }
#define DMAS_B() { if (DPS >= 2) DMA(2048); if (DPS >= 4) DMA(3072); }
// load segment, K: 16 ds_read_b128 (1 KiB each) + this wave's DMA share; everything landed before the segment ends
#define LOADK()                                                                                           \
    LDS128(0, 0); LDS128(1, 1024); LDS128(2, 2048); LDS128(3, 3072);                      \
    LDS128(4, 4096); LDS128(5, 5120); LDS128(6, 6144); LDS128(7, 7168);                   \
    DMAS_A()                                                                                              \
    LDS128(8, 8192); LDS128(9, 9216); LDS128(10, 10240); LDS128(11, 11264);               \
    LDS128(12, 12288); LDS128(13, 13312); LDS128(14, 14336); LDS128(15, 15360);           \
    DMAS_B()                                                                                              \
    WAITL(0);
// load segment, V: 32 ds_read_b64_tr_b16 (512 B each)
#define LOADV()                                                                                           \
    LDSTR2(0, 0) LDSTR2(1, 1024) LDSTR2(2, 2048) LDSTR2(3, 3072) LDSTR2(4, 4096) LDSTR2(5, 5120) LDSTR2(6, 6144) LDSTR2(7, 7168) \
    DMAS_A()                                                                                              \
    LDSTR2(8, 8192) LDSTR2(9, 9216) LDSTR2(10, 10240) LDSTR2(11, 11264) LDSTR2(12, 12288) LDSTR2(13, 13312) LDSTR2(14, 14336) LDSTR2(15, 15360) \
    DMAS_B()                                                                                              \
    WAITL(0);

    uint64_t t0, t1;
    if (MODE == 1 && wave >= 4) __builtin_amdgcn_s_setprio(1);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    if constexpr (MODE == 0 || MODE == 1 || MODE == 5) {
        if (wave < 4) {
            for (int t = 0; t < tiles; ++t) {
                COMPUTE16() BAR(); LOADV() BAR(); COMPUTE16() WAITV(2 * DPS); BAR(); LOADK() BAR();
            }
        } else {
            for (int t = 0; t < tiles; ++t) {
                LOADK() BAR(); COMPUTE16() BAR(); LOADV() WAITV(2 * DPS); BAR(); COMPUTE16() BAR();
            }
        }
    } else if constexpr (MODE == 4) {
        if (wave < 4) {
            for (int t = 0; t < tiles; ++t) {
                COMPUTE16() COMPUTE16() WAITV(2 * DPS); BAR(); LOADK() LOADV() BAR();
            }
        } else {
            for (int t = 0; t < tiles; ++t) {
                LOADK() LOADV() WAITV(2 * DPS); BAR(); COMPUTE16() COMPUTE16() BAR();
            }
        }
    } else if constexpr (MODE == 3) {
        // 32-row waves, own stream: per tile 32 MFMAs; K fragment read behind each QK MFMA, two V reads behind each PV MFMA
        for (int t = 0; t < tiles; ++t) {
#define IK(J) CGROUP(J, J) LDS128(J, 1024 * (J));
#define IV(J) CGROUP(J, J) LDSTR2(J, 1024 * (J))
            IK(0) IK(1) IK(2) IK(3) DMAS_A() IK(4) IK(5) IK(6) IK(7) WAITL(4); IK(8) IK(9) IK(10) IK(11) DMAS_B() IK(12) IK(13) IK(14) IK(15) WAITL(4);
            IV(0) IV(1) IV(2) IV(3) DMAS_A() IV(4) IV(5) IV(6) IV(7) WAITL(4); IV(8) IV(9) IV(10) IV(11) DMAS_B() IV(12) IV(13) IV(14) IV(15) WAITL(4);
            WAITV(2 * DPS);
            BAR();
        }
    } else {
        // MODE 2: 64-row waves, one per SIMD: every fragment feeds two MFMAs; per tile 64 MFMAs, 48 reads, 4 DPS KiB
        for (int t = 0; t < tiles; ++t) {
#define PK(J) CGROUP(J, J) CGROUP(((J) + 8) & 15, J) LDS128(J, 1024 * (J));
#define PVV(J) CGROUP(J, J) CGROUP(((J) + 8) & 15, J) LDSTR2(J, 1024 * (J))
            PK(0) PK(1) DMAS_A() PK(2) PK(3) PK(4) PK(5) DMAS_B() PK(6) PK(7) WAITL(4); PK(8) PK(9) DMAS_A() PK(10) PK(11) PK(12) PK(13) DMAS_B() PK(14) PK(15) WAITL(4);
            PVV(0) PVV(1) DMAS_A() PVV(2) PVV(3) PVV(4) PVV(5) DMAS_B() PVV(6) PVV(7) WAITL(4); PVV(8) PVV(9) DMAS_A() PVV(10) PVV(11) PVV(12) PVV(13) DMAS_B() PVV(14) PVV(15) WAITL(4);
            WAITV(4 * DPS);
            BAR();
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 15\n\ts_nop 7\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    float r = sum + tm;
#pragma unroll
    for (int i = 0; i < 16; ++i) r += s[i] + (float)kf[i][0];
#pragma unroll
    for (int i = 0; i < 4; ++i) r += acc[i][i];
#pragma unroll
    for (int i = 0; i < 8; ++i) r += (float)p[i];
    out[blockIdx.x * blockDim.x + tid] = r;
    if (blockIdx.x == 0 && lane == 0) cyc[wave] = t1 - t0;
}

template <int MODE, int DPS>
void run(const u32x4* s, float* o, unsigned long long* c, const char* what) {
    const int tiles = 512, threads = MODE == 2 ? 256 : 512;
    hipFuncSetAttribute((const void*)probe<MODE, DPS>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((probe<MODE, DPS>), dim3(256), dim3(threads), 96 * 1024, 0, s, o, c, tiles);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<MODE, DPS>), dim3(256), dim3(threads), 96 * 1024, 0, s, o, c, tiles);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[8]; hipMemcpy(h, c, sizeof(h), hipMemcpyDeviceToHost);
    const double mfma_per_simd = 64.0 * tiles;  // per tile and SIMD: 2 waves x 32 or 1 wave x 64
    const double flops = 256.0 * 4 * mfma_per_simd * 32 * 32 * 16 * 2;
    printf("%-46s DMA %d KiB/16 MFMA: %6.1f cyc/MFMA/SIMD (wave 0) %6.1f (last wave)  wall %7.1f us  %6.0f TFLOP/s  clock %.2f GHz\n", what, DPS,
           h[0] / mfma_per_simd, h[threads / 64 - 1] / mfma_per_simd, ms * 1e3, flops / (ms * 1e-3) / 1e12, h[0] / (ms * 1e-3) / 1e9);
}

int main() {
    u32x4* s; float* o; unsigned long long* c;
    hipMalloc(&s, 1 << 20); hipMalloc(&o, 256 * 512 * 4); hipMalloc(&c, 64);
    hipMemset(s, 0x3c, 1 << 20);
    run<2, 4>(s, o, c, "4 waves, one stream (64-row waves)");
    run<2, 2>(s, o, c, "4 waves, one stream (64-row waves)");
    run<2, 0>(s, o, c, "4 waves, one stream (64-row waves)");
    run<3, 2>(s, o, c, "8 waves, own streams (32-row waves)");
    run<3, 0>(s, o, c, "8 waves, own streams (32-row waves)");
    run<0, 4>(s, o, c, "8 waves, ping-pong 16-MFMA segments");
    run<0, 2>(s, o, c, "8 waves, ping-pong 16-MFMA segments");
    run<0, 0>(s, o, c, "8 waves, ping-pong 16-MFMA segments");
    run<1, 2>(s, o, c, "8 waves, ping-pong + setprio(young half)");
    run<4, 2>(s, o, c, "8 waves, ping-pong 32-MFMA segments");
    run<4, 0>(s, o, c, "8 waves, ping-pong 32-MFMA segments");
    run<5, 2>(s, o, c, "8 waves, ping-pong, no softmax VALU");
    return 0;
}
