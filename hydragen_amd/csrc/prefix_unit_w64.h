#pragma once
// Prefix pass (unit body; the kernel and its launcher are in prefix_attn_w64.hip): batched-query attention of all queries of a group against the group's single shared K/V, on the
// gfx950 matrix cores.  Replaces flash-attn's _flash_attn_forward / _flash_attn_varlen_forward as called from
// /root/reference/hydragen/flash.py:284-351 and hydragen/attention.py:270,313,344.
//
// Work decomposition: one unit body, two wave geometries over the SAME rows per workgroup, LDS rings and pipeline (NW below):
//   NW = 8 (the default for D = 128): 512 threads = 8 waves, two per SIMD, 32 folded query rows (b_local, iq, gqa head) per
//           wave, 256 registers per wave, O / Q in the literal VGPRs v[160:255] (RegsV); the SIMD issues one wave's VALU /
//           LDS / DMA instructions beside the other wave's MFMAs.
//   NW = 4 (D = 64 / 256, persistent launches): 256 threads = 4 waves, one per SIMD, 64 rows = two 32-row query blocks per
//           wave, so that each K / V^T fragment read from LDS feeds TWO MFMAs and the two blocks' softmax chains interleave
//           in the MFMA shadow; 512 registers per wave, O / Q in the literal AGPRs a[0:191] (RegsA).
//   KG = 2: 128 rows per workgroup, wave = (row sub-block, key half of every 128-key tile); the two key halves keep
//           independent (m, l, O) and are merged through LDS at the end.
//   KG = 1: 256 rows per workgroup, every wave walks all keys (shapes with enough rows to fill the chip that way).
//   S^T = K.Q^T and O^T += V^T.P^T with v_mfma_f32_32x32x16: computing the transposed products puts one query row per
//   lane, so the softmax reductions are in-lane plus one v_permlane32_swap, and P^T is already the B operand.
//
// Pipeline (per wave, over 32-key blocks b): iteration i interleaves, instruction by instruction,
//   MFMA stream: QK(i+1) and PV(i-1), alternating
//   VALU stream: online softmax of block i (exp2, row sums, pack to 16-bit P^T), one group of 4-5 instructions behind
//                every MFMA, the order pinned by sched_barriers
// The running maximum is only raised when a block exceeds it by more than 2^kTau (the O rescale is a cold wave-uniform
// branch): NW = 4 tests the block's maximum (max3 tree) against a threshold; NW = 8 has no maximum tree at all -- it tests
// the lane's SUM of the block's probabilities (LAZY, see the iteration) and recomputes the block on the cold path.
// K and V live in LDS rings of four 32-key block slots filled by LDS-DMA (buffer_load ... lds, zero fill past the end of
// the keys): iteration i issues K block i+4 and V block i+2 and its barrier only waits for the DMAs issued during
// iteration i-1.
// Register ownership (the contract tests/test_build_quality.py and csrc/regcheck.py enforce): the asm statements of RegsV /
// RegsA name their registers literally and carry no operands for them; hipcc is kept out by the kernels' register budgets
// (prefix_launch_w64.h) and because nothing spills.
#include <type_traits>
#include <utility>

#include "hyd_kernels.h"

namespace hyd {

namespace {

typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr_w;

template <int... Is, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, Is...>, F&& f) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f));
}

__device__ __forceinline__ u32x2 lds_tr16_w(unsigned lds_byte_addr) {
    s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr_w)(uintptr_t)lds_byte_addr);
    return __builtin_bit_cast(u32x2, t);
}

// One LDS-DMA instruction: 64 lanes x 16 B from global memory to LDS [lds_dst, lds_dst + 1 KiB) (lds_dst wave-uniform,
// lane l lands at lds_dst + 16 l).  The source is a raw buffer resource plus a per-lane byte offset (+ a scalar byte
// offset): lanes at or past num_records write zeros instead of faulting (tests/probes/bufdma_probe.hip), which is how
// rows past the end of the keys, and whole blocks that do not exist, are handled without address arithmetic.
// Issued from inline asm on purpose: hipcc then neither drains it (vmcnt(0)) in front of the LDS reads of the current
// blocks nor counts it; the loop waits for it explicitly (dma_wait_w<N>) before its barrier.  M0 is not used by any
// compiler-generated instruction of this kernel (gfx9 LDS instructions do not read it), so it is not saved.
// A wave's NLB pieces of one block sit 1 KiB apart in LDS: M0 is written once per tensor and block (dma_m0) and piece i
// is addressed through the instruction's immediate offset 1024 i, which the hardware adds to the LDS address AND to the
// global offset -- the per-lane source offsets of piece i are made 1024 i smaller to compensate (koffb / voffb below).
__device__ __forceinline__ void dma_m0(unsigned lds_dst) { asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" ::"s"(lds_dst) : "memory"); }
// PAD: hipcc does not see inside the asm, so it inserts no wait states between a VALU instruction that writes one of the
// statement's scalar operands and the buffer_load that reads it (5 wait states on gfx9).  In the one-unit kernel those
// operands are written by scalar instructions long before; the persistent kernel's unit loop spills scalars to VGPR lanes
// and restores them (v_readlane) wherever it likes, also right in front of this statement: it pads.
// tests/test_build_quality.py checks the distance in the compiled code of every instantiation.
template <int OFF, bool PAD>
__device__ __forceinline__ void dma16w(u32x4 rsrc, unsigned voff, unsigned soff) {
    static_assert(OFF >= 0 && OFF < 4096, "12-bit immediate offset");
    if constexpr (PAD) asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2 offen offset:%3 lds" ::"v"(voff), "s"(rsrc), "s"(soff), "n"(OFF) : "memory");
    else asm volatile("buffer_load_dwordx4 %0, %1, %2 offen offset:%3 lds" ::"v"(voff), "s"(rsrc), "s"(soff), "n"(OFF) : "memory");
}
__device__ __forceinline__ u32x4 make_rsrc_w(const char* base, unsigned bytes) {
    const uint64_t b = (uint64_t)(uintptr_t)base;
    u32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((unsigned)b);
    r[1] = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32) & 0xffffu);
    r[2] = __builtin_amdgcn_readfirstlane(bytes);
    r[3] = 0x00020000u;
    return r;
}
template <int N>
__device__ __forceinline__ void dma_wait_w() { asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory"); }

// MFMAs as inline asm with the O accumulators and the Q fragments in LITERAL accumulator registers that only these
// statements touch: O block I = qb * NDB + db in a[16 I : 16 I + 15] (a[0:127]), Q fragment I = qb * NC + c in
// a[128 + 4 I : 131 + 4 I] (a[128:191]); scores / P / K / V fragments stay in compiler-allocated VGPRs, where the VALU
// works on them.  (Given 512 registers and MFMA builtins, or asm operands with register-class or even physical-register
// constraints, hipcc routes scores and spills through AGPRs and migrates accumulators between the files across
// iterations: 4-9 extra v_accvgpr_* per MFMA, scratch traffic and a vmcnt(0) inside the loop.)  claim_agprs() makes
// the kernel descriptor allocate the registers; tests/test_build_quality.py asserts that no compiler-generated
// instruction names an AGPR (the kernel has no spills, so hipcc has no use for them).
// Hazards hipcc does not see (cdna_hip_programming.md 5.7): a score block is read by the VALU an iteration after the
// MFMAs that wrote it were issued; P is consumed an iteration after the VALU packed it; every VALU access to an
// accumulator (cold rescale, epilogue) is preceded by acc_drain(); v_accvgpr_write -> MFMA is padded inside the asm.
// The register NUMBERS are assembler expressions of a template constant (`a[%2:%3]`, `a[%0+\r]` inside .irp): one generic
// definition serves the 8 O blocks and the 16 Q fragments.  These statements carry no clobber lists; claim_agprs()
// below names a[0:191] once per unit, which is what sizes the kernel descriptor's accumulator file.
#define HYD_A10(p) "a" #p "0", "a" #p "1", "a" #p "2", "a" #p "3", "a" #p "4", "a" #p "5", "a" #p "6", "a" #p "7", "a" #p "8", "a" #p "9"
__device__ __forceinline__ void claim_agprs() {  // a[0:191] of the 4-wave unit
    asm volatile("" ::: "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", HYD_A10(1), HYD_A10(2), HYD_A10(3), HYD_A10(4),
                 HYD_A10(5), HYD_A10(6), HYD_A10(7), HYD_A10(8), HYD_A10(9), HYD_A10(10), HYD_A10(11), HYD_A10(12), HYD_A10(13),
                 HYD_A10(14), HYD_A10(15), HYD_A10(16), HYD_A10(17), HYD_A10(18), "a190", "a191");
}
#undef HYD_A10
#define HYD_IRP16 ".irp r,0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15\n\t"

template <int I>
struct OAcc {  // O block I = qb * NDB + db: a[16 I : 16 I + 15]
    static constexpr int R0 = 16 * I;
    template <bool BF>
    static __device__ __forceinline__ void pv(const u32x4& a, const u32x4& b) {
        if constexpr (BF) asm volatile("v_mfma_f32_32x32x16_bf16 a[%2:%3], %0, %1, a[%2:%3]" ::"v"(a), "v"(b), "n"(R0), "n"(R0 + 15));
        else asm volatile("v_mfma_f32_32x32x16_f16 a[%2:%3], %0, %1, a[%2:%3]" ::"v"(a), "v"(b), "n"(R0), "n"(R0 + 15));
    }
    static __device__ __forceinline__ void zero() {
        asm volatile(HYD_IRP16 "v_accvgpr_write_b32 a[%0+\\r], 0\n\t.endr\n\ts_nop 1" ::"n"(R0));
    }
    static __device__ __forceinline__ void scale(float f) {
        float t;
        asm volatile(HYD_IRP16 "v_accvgpr_read_b32 %0, a[%2+\\r]\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a[%2+\\r], %0\n\t.endr\n\ts_nop 1"
                     : "=&v"(t)
                     : "v"(f), "n"(R0));
    }
    static __device__ __forceinline__ void read(float (&x)[16]) {
        asm volatile("v_accvgpr_read_b32 %0, a[%16]\n\tv_accvgpr_read_b32 %1, a[%16+1]\n\tv_accvgpr_read_b32 %2, a[%16+2]\n\t"
                     "v_accvgpr_read_b32 %3, a[%16+3]\n\tv_accvgpr_read_b32 %4, a[%16+4]\n\tv_accvgpr_read_b32 %5, a[%16+5]\n\t"
                     "v_accvgpr_read_b32 %6, a[%16+6]\n\tv_accvgpr_read_b32 %7, a[%16+7]\n\tv_accvgpr_read_b32 %8, a[%16+8]\n\t"
                     "v_accvgpr_read_b32 %9, a[%16+9]\n\tv_accvgpr_read_b32 %10, a[%16+10]\n\tv_accvgpr_read_b32 %11, a[%16+11]\n\t"
                     "v_accvgpr_read_b32 %12, a[%16+12]\n\tv_accvgpr_read_b32 %13, a[%16+13]\n\tv_accvgpr_read_b32 %14, a[%16+14]\n\t"
                     "v_accvgpr_read_b32 %15, a[%16+15]"
                     : "=v"(x[0]), "=v"(x[1]), "=v"(x[2]), "=v"(x[3]), "=v"(x[4]), "=v"(x[5]), "=v"(x[6]), "=v"(x[7]), "=v"(x[8]),
                       "=v"(x[9]), "=v"(x[10]), "=v"(x[11]), "=v"(x[12]), "=v"(x[13]), "=v"(x[14]), "=v"(x[15])
                     : "n"(R0));
    }
};

template <int I>
struct QFrag {  // Q fragment I = qb * NC + c: a[128 + 4 I : 131 + 4 I]
    static constexpr int R0 = 128 + 4 * I;
    template <int OFF>
    static __device__ __forceinline__ void load(const void* p) {
        asm volatile("global_load_dwordx4 a[%1:%2], %0, off offset:%3" ::"v"(p), "n"(R0), "n"(R0 + 3), "n"(OFF) : "memory");
    }
    template <bool BF, bool FIRST>
    static __device__ __forceinline__ void qk(f32x16& s, const u32x4& a) {
        if constexpr (BF && FIRST) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, a[%2:%3], 0" : "=&v"(s) : "v"(a), "n"(R0), "n"(R0 + 3));
        else if constexpr (BF) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, a[%2:%3], %0" : "+v"(s) : "v"(a), "n"(R0), "n"(R0 + 3));
        else if constexpr (FIRST) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, a[%2:%3], 0" : "=&v"(s) : "v"(a), "n"(R0), "n"(R0 + 3));
        else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, a[%2:%3], %0" : "+v"(s) : "v"(a), "n"(R0), "n"(R0 + 3));
    }
};
// every MFMA issued so far has written its result (8-pass XDL: 18 wait states cover any reader)
__device__ __forceinline__ void acc_drain() { asm volatile("s_nop 15\n\ts_nop 7" ::: "memory"); }

// RegsA -- one wave per SIMD (4-wave workgroups; 64-row waves, or one 32-row block at D = 256): O block I lives in the literal
// accumulator registers a[16 I : 16 I + 15], Q fragment I in a[128 + 4 I : 131 + 4 I]; a[0:191] of the wave's 512 registers are
// claimed for the kernel descriptor.
struct RegsA {
    __device__ __forceinline__ void claim() { claim_agprs(); }
    template <int I, bool BF>
    __device__ __forceinline__ void pv(const u32x4& a, const u32x4& b) { OAcc<I>::template pv<BF>(a, b); }
    template <int I>
    __device__ __forceinline__ void zero() { OAcc<I>::zero(); }
    // cold path of the softmax: every O block of query block qb times alpha[qb] (pend is wave-uniform)
    template <int NQB, int NDB_>
    __device__ __forceinline__ void rescale(bool pend, const float (&alpha)[NQB]) {
        if (pend) {
            acc_drain();
            static_for<NQB * NDB_>([&](auto I_) { constexpr int I = decltype(I_)::value; OAcc<I>::scale(alpha[I / NDB_]); });
        }
    }
    template <int I>
    __device__ __forceinline__ void read(float (&x)[16]) { OAcc<I>::read(x); }
    template <int I, int OFF>
    __device__ __forceinline__ void qload(const void* p) { QFrag<I>::template load<OFF>(p); }
    template <int I, bool BF, bool FIRST>
    __device__ __forceinline__ void qk(f32x16& s, const u32x4& a) { QFrag<I>::template qk<BF, FIRST>(s, a); }
    __device__ __forceinline__ void drain() { acc_drain(); }
};
// RegsV -- two waves per SIMD (8-wave workgroups, 32-row waves, 256 registers per wave): no accumulator registers at all
// (on gfx950 every MFMA operand may be an architectural VGPR).  O block I lives in the literal registers
// v[160 + 16 I : 175 + 16 I], Q fragment I in v[224 + 4 I : 227 + 4 I], named in the asm text only, exactly as RegsA names
// its accumulator registers.  What keeps hipcc out of them is the kernel's register budget: the 8-wave kernel is declared
// amdgpu_num_vgpr(80), which for a function that uses no accumulator register means "at most 160 architectural VGPRs"
// (v0..v159 are all it can allocate; a function that does use accumulator registers is held to a 128 / 128 split, too few
// for this loop, and hipcc then parks values in whatever accumulator registers it believes free).  claim() names v255
// once so that the kernel descriptor allocates the whole file; tests/test_build_quality.py asserts that no
// compiler-generated instruction names a register above v159 and that nothing spills.
template <int I>
struct OAccV {  // O block I: v[160 + 16 I : 175 + 16 I]
    static constexpr int R0 = 160 + 16 * I;
    template <bool BF>
    static __device__ __forceinline__ void pv(const u32x4& a, const u32x4& b) {
        if constexpr (BF) asm volatile("v_mfma_f32_32x32x16_bf16 v[%2:%3], %0, %1, v[%2:%3]" ::"v"(a), "v"(b), "n"(R0), "n"(R0 + 15));
        else asm volatile("v_mfma_f32_32x32x16_f16 v[%2:%3], %0, %1, v[%2:%3]" ::"v"(a), "v"(b), "n"(R0), "n"(R0 + 15));
    }
    static __device__ __forceinline__ void zero() { asm volatile(HYD_IRP16 "v_mov_b32 v[%0+\\r], 0\n\t.endr\n\ts_nop 1" ::"n"(R0)); }
    static __device__ __forceinline__ void scale(float f) {
        asm volatile(HYD_IRP16 "v_mul_f32 v[%1+\\r], v[%1+\\r], %0\n\t.endr\n\ts_nop 1" ::"v"(f), "n"(R0));
    }
    static __device__ __forceinline__ void read(float (&x)[16]) {
        asm volatile("v_mov_b32 %0, v[%16]\n\tv_mov_b32 %1, v[%16+1]\n\tv_mov_b32 %2, v[%16+2]\n\tv_mov_b32 %3, v[%16+3]\n\t"
                     "v_mov_b32 %4, v[%16+4]\n\tv_mov_b32 %5, v[%16+5]\n\tv_mov_b32 %6, v[%16+6]\n\tv_mov_b32 %7, v[%16+7]\n\t"
                     "v_mov_b32 %8, v[%16+8]\n\tv_mov_b32 %9, v[%16+9]\n\tv_mov_b32 %10, v[%16+10]\n\tv_mov_b32 %11, v[%16+11]\n\t"
                     "v_mov_b32 %12, v[%16+12]\n\tv_mov_b32 %13, v[%16+13]\n\tv_mov_b32 %14, v[%16+14]\n\tv_mov_b32 %15, v[%16+15]"
                     : "=v"(x[0]), "=v"(x[1]), "=v"(x[2]), "=v"(x[3]), "=v"(x[4]), "=v"(x[5]), "=v"(x[6]), "=v"(x[7]), "=v"(x[8]),
                       "=v"(x[9]), "=v"(x[10]), "=v"(x[11]), "=v"(x[12]), "=v"(x[13]), "=v"(x[14]), "=v"(x[15])
                     : "n"(R0));
    }
};
template <int I>
struct QFragV {  // Q fragment I: v[224 + 4 I : 227 + 4 I]
    static constexpr int R0 = 224 + 4 * I;
    template <int OFF>
    static __device__ __forceinline__ void load(const void* p) {
        asm volatile("global_load_dwordx4 v[%1:%2], %0, off offset:%3" ::"v"(p), "n"(R0), "n"(R0 + 3), "n"(OFF) : "memory");
    }
    template <bool BF, bool FIRST>
    static __device__ __forceinline__ void qk(f32x16& s, const u32x4& a) {
        if constexpr (BF && FIRST) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, v[%2:%3], 0" : "=&v"(s) : "v"(a), "n"(R0), "n"(R0 + 3));
        else if constexpr (BF) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, v[%2:%3], %0" : "+v"(s) : "v"(a), "n"(R0), "n"(R0 + 3));
        else if constexpr (FIRST) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, v[%2:%3], 0" : "=&v"(s) : "v"(a), "n"(R0), "n"(R0 + 3));
        else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, v[%2:%3], %0" : "+v"(s) : "v"(a), "n"(R0), "n"(R0 + 3));
    }
};
struct RegsV {
    __device__ __forceinline__ void claim() { asm volatile("" ::: "v255"); }
    template <int I, bool BF>
    __device__ __forceinline__ void pv(const u32x4& a, const u32x4& b) { OAccV<I>::template pv<BF>(a, b); }
    template <int I>
    __device__ __forceinline__ void zero() { OAccV<I>::zero(); }
    template <int NQB, int NDB_>
    __device__ __forceinline__ void rescale(bool pend, const float (&alpha)[NQB]) {
        if (pend) {
            acc_drain();
            static_for<NQB * NDB_>([&](auto I_) { constexpr int I = decltype(I_)::value; OAccV<I>::scale(alpha[I / NDB_]); });
        }
    }
    template <int I>
    __device__ __forceinline__ void read(float (&x)[16]) { OAccV<I>::read(x); }
    template <int I, int OFF>
    __device__ __forceinline__ void qload(const void* p) { QFragV<I>::template load<OFF>(p); }
    template <int I, bool BF, bool FIRST>
    __device__ __forceinline__ void qk(f32x16& s, const u32x4& a) { QFragV<I>::template qk<BF, FIRST>(s, a); }
    __device__ __forceinline__ void drain() { acc_drain(); }
};

#undef HYD_IRP16
// Element phase of one query block's softmax: 54 operations on the 8 element pairs p (elements 2p, 2p + 1) -- fa / fb: the
// fma that applies scale and reference maximum, ea / eb: exp2, sa / sb: the two row-sum chains, pk: pack to 16-bit P^T --
// as 14 groups, one behind each of the block's MFMAs 2..15.  An MFMA's shadow holds 5 issue slots and exp2 takes two: the
// 54 operations are 70 slots = 14 x 5, so the table packs every group to exactly five (one exp2 + three plain operations,
// two exp2 + one, or five plain ones), with every exp2 at least two instructions (in practice: one MFMA) behind the
// fma that feeds it -- back to back hipcc pads the pair with an s_nop -- and every add / pack a group behind its exp2s.
// Code = kind * 8 + pair; kinds: 0 fa, 1 fb, 2 ea, 3 eb, 4 sa, 5 sb, 6 pk.
#define HYD_OP(kind, pair) ((kind) * 8 + (pair))
constexpr int kElemOps[54] = {
    HYD_OP(0, 0), HYD_OP(1, 0), HYD_OP(0, 1), HYD_OP(2, 0),  // g2
    HYD_OP(3, 0), HYD_OP(1, 1), HYD_OP(0, 2), HYD_OP(1, 2),  // g3
    HYD_OP(2, 1), HYD_OP(6, 0), HYD_OP(0, 3), HYD_OP(1, 3),  // g4
    HYD_OP(3, 1), HYD_OP(0, 4), HYD_OP(1, 4), HYD_OP(4, 1),  // g5   (sa of pair 1: S[0] + S[2])
    HYD_OP(2, 2), HYD_OP(6, 1), HYD_OP(5, 1), HYD_OP(0, 5),  // g6
    HYD_OP(3, 2), HYD_OP(1, 5), HYD_OP(4, 2), HYD_OP(0, 6),  // g7
    HYD_OP(2, 3), HYD_OP(6, 2), HYD_OP(5, 2), HYD_OP(1, 6),  // g8
    HYD_OP(3, 3), HYD_OP(4, 3), HYD_OP(0, 7), HYD_OP(1, 7),  // g9
    HYD_OP(2, 4), HYD_OP(6, 3), HYD_OP(3, 4),                // g10
    HYD_OP(2, 5), HYD_OP(5, 3), HYD_OP(4, 4), HYD_OP(5, 4),  // g11
    HYD_OP(3, 5), HYD_OP(6, 4), HYD_OP(2, 6),                // g12
    HYD_OP(3, 6), HYD_OP(4, 5), HYD_OP(5, 5), HYD_OP(6, 5),  // g13
    HYD_OP(2, 7), HYD_OP(4, 6), HYD_OP(3, 7),                // g14
    HYD_OP(5, 6), HYD_OP(6, 6), HYD_OP(4, 7), HYD_OP(5, 7), HYD_OP(6, 7),  // g15  (no exp2 left: nothing here reads a result of its own group)
};
constexpr int kElemGroupStart[15] = {0, 4, 8, 12, 16, 20, 24, 28, 32, 35, 39, 42, 46, 49, 54};
// The same 54 operations for the 8-wave unit's softmax, which has no maximum tree in front (see LAZY in the iteration): all
// 16 groups carry element work, 4 / 5 issue slots alternating, every consumer at least one group behind its producer, at
// most six exponentials alive at a time (list-scheduled, tools-free: the rule is in the comment of kElemOps).
constexpr int kElemOpsL[54] = {
    HYD_OP(0, 0), HYD_OP(1, 0), HYD_OP(0, 1), HYD_OP(1, 1),                // g0
    HYD_OP(0, 2), HYD_OP(2, 0), HYD_OP(3, 0),                              // g1
    HYD_OP(6, 0), HYD_OP(2, 1), HYD_OP(1, 2),                              // g2
    HYD_OP(4, 1), HYD_OP(3, 1), HYD_OP(2, 2),                              // g3
    HYD_OP(6, 1), HYD_OP(3, 2), HYD_OP(5, 1),                              // g4
    HYD_OP(6, 2), HYD_OP(4, 2), HYD_OP(5, 2), HYD_OP(0, 3), HYD_OP(1, 3),  // g5
    HYD_OP(2, 3), HYD_OP(3, 3),                                            // g6
    HYD_OP(6, 3), HYD_OP(4, 3), HYD_OP(5, 3), HYD_OP(0, 4), HYD_OP(1, 4),  // g7
    HYD_OP(2, 4), HYD_OP(3, 4),                                            // g8
    HYD_OP(6, 4), HYD_OP(4, 4), HYD_OP(5, 4), HYD_OP(0, 5), HYD_OP(1, 5),  // g9
    HYD_OP(2, 5), HYD_OP(3, 5),                                            // g10
    HYD_OP(6, 5), HYD_OP(4, 5), HYD_OP(5, 5), HYD_OP(0, 6), HYD_OP(1, 6),  // g11
    HYD_OP(2, 6), HYD_OP(3, 6),                                            // g12
    HYD_OP(6, 6), HYD_OP(4, 6), HYD_OP(5, 6), HYD_OP(0, 7), HYD_OP(1, 7),  // g13
    HYD_OP(2, 7), HYD_OP(3, 7),                                            // g14
    HYD_OP(6, 7), HYD_OP(4, 7), HYD_OP(5, 7),                              // g15
};
constexpr int kElemGroupStartL[17] = {0, 4, 7, 10, 13, 16, 21, 23, 28, 30, 35, 37, 42, 44, 49, 51, 54};
#undef HYD_OP

}  // namespace

// ABL: development-only timing ablations (bit0 no in-loop DMA, bit2 no softmax VALU, bit3 no LDS fragment reads,
// bit4 no barrier in the loop, bit5 no exp2, bit6 no element phase, bit7 no row sums / pack); only ABL = 0 ships.
// QB: 32-row query blocks per wave.  2 for D <= 128 (64 rows per wave); 1 for D = 256, where one block's O accumulators
// (8 x 16) and Q fragments (16 x 4) fill the same a[0:191] that two blocks fill at D = 128 (then always KG = 1: 128
// rows per workgroup, every wave walks all keys, 128 KB of rings).
// PERSIST: the unit runs inside a persistent workgroup's unit loop (see dma16w's PAD).
// NW: waves per workgroup.  4 = one wave per SIMD, 64-row waves (512 registers per wave, RegsA); 8 = two waves per SIMD, 32-row
// waves (256 registers per wave, all architectural VGPRs: v[160:255] for O / Q (RegsV) + at most 160 of hipcc's): the same rows
// per workgroup, the same rings, the same pipeline per wave -- half the MFMAs per wave and iteration, and the SIMD issues one wave's VALU /
// LDS / DMA instructions in the shadow of the other wave's MFMAs (tests/probes/pingpong_probe.hip: 36.5 against 46.7 - 53
// cycles per MFMA and SIMD for the instruction mix of this loop).
template <typename T, int D, bool CAUSAL, int KG, int ABL = 0, bool PERSIST = false, int NW = 4, int QB = ((D > 128 || NW == 8) ? 1 : 2)>
__device__ __forceinline__ void prefix_unit_w64(const PrefixArgs& a, const int vblock, const int vgrid, char* smem) {
    using TR = Traits<T>;
    static_assert(NW == 4 || (NW == 8 && QB == 1 && D <= 128), "two waves per SIMD: 32-row waves, D <= 128");
    std::conditional_t<NW == 8, RegsV, RegsA> regs;
    regs.claim();
    if constexpr ((ABL & 4096) != 0) asm volatile("s_nop 0");  // development: shifts the whole stream by 4 bytes (code-placement probe)
    static_assert(QB == 1 || QB == 2, "query blocks per wave");
    static_assert(D <= 128 || (QB == 1 && KG == 1), "D = 256 runs one query block per wave and unsplit key tiles");
    constexpr int WROWS = 32 * QB;       // query rows per wave
    constexpr int RB = D * 2;            // bytes per K/V row
    constexpr int NC = D / 16;           // k-chunks of the QK^T contraction
    constexpr int NDB = D / 32;          // 32-wide d blocks of O^T
    constexpr int NRW = NW / KG;                              // row sub-blocks (waves per key half)
    constexpr int RWG = NRW * WROWS;                          // query rows per workgroup
    constexpr int BROWS = KG == 2 ? 64 : 32;                  // K (or V) rows staged per iteration
    constexpr int NLB = BROWS * RB / 1024 / NW;               // DMA instructions per wave per tensor per iteration
    constexpr int RPI = 1024 / RB;                            // rows per DMA instruction
    static_assert(NLB >= 1, "every wave issues at least one DMA instruction per tensor and iteration");
    constexpr int RING_BYTES = (KG == 2 ? 512 : 256) * RB;
    float* mlbuf = reinterpret_cast<float*>(smem + RING_BYTES);  // [NW waves][QB][2][64] (KG = 2 merge)

    // ABL bit 11 (development builds only): time stamps of EVERY workgroup's wave 0 -- s_memtime (shader-clock cycles) at six
    // points, s_memrealtime (the constant 100 MHz counter) at the first and the last: where a workgroup's time goes, when it
    // started relative to the others, and the clock it ran at (cycles / real time), tools/prefix_timeline.py.
    unsigned tst[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    auto stampk = [&](int k) __attribute__((always_inline)) {
        if constexpr ((ABL & 2048) != 0) {
            uint64_t t;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
            tst[k] = (unsigned)t;
            if (k == 0 || k == 5) {
                asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
                tst[k == 0 ? 6 : 7] = (unsigned)t;
            }
        }
    };
    stampk(0);
    // Opaque per unit: a persistent workgroup calls this in a loop, and everything derived from the lane index is
    // loop-invariant there -- hoisted out of the unit loop it would stay live across the whole pipeline and push the
    // allocation past 256 VGPRs (hipcc then parks values in AGPRs, which the asm statements own).
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kg = KG == 2 ? wave / NRW : 0;  // key half of every 128-key tile this wave computes on
    const int rw = KG == 2 ? wave % NRW : wave;  // row sub-block (WROWS rows)
    const int l31 = lane & 31, hi = lane >> 5;

    // ---- which (group, kv head, split, row block) ------------------------------------------
    // (no divide on the device: magic numbers from the host, hyd_kernels.h FastDiv)
    auto fdiv = [](unsigned n, const FastDiv& f) __attribute__((always_inline)) -> unsigned {
        const unsigned t = __umulhi(n, f.mul);
        return (t + ((n - t) >> (f.sh & 0xffu))) >> (f.sh >> 8);
    };
    const int lin = xcd_remap(vblock, vgrid);
    int t_ = (int)fdiv((unsigned)lin, a.div_row_blocks);
    const int rb = lin - t_ * a.row_blocks;
    const int t2_ = (int)fdiv((unsigned)t_, a.div_nsplit);
    const int sp = t_ - t2_ * a.nsplit;
    const int gi = (int)fdiv((unsigned)t2_, a.div_hkv);
    const int hk = t2_ - gi * a.Hkv;

    int q_tok0, nqtok, nq_eff;
    if (a.cu_q) {
        q_tok0 = a.cu_q[gi];
        nqtok = a.cu_q[gi + 1] - q_tok0;
        nq_eff = nqtok;
    } else {
        q_tok0 = gi * a.per * a.nq;
        nqtok = a.per * a.nq;
        nq_eff = a.nq;
    }
    const int Mrows = nqtok * a.g;
    if (rb * RWG >= Mrows) return;  // block-uniform

    const uint16_t* k16 = static_cast<const uint16_t*>(a.k);
    const uint16_t* v16 = static_cast<const uint16_t*>(a.v);
    int L;
    if (a.cu_k) {
        const int t0 = a.cu_k[gi];
        L = a.cu_k[gi + 1] - t0;
        k16 += (int64_t)t0 * a.k_ts;
        v16 += (int64_t)t0 * a.v_ts;
    } else {
        L = a.kv_len;
        k16 += (int64_t)gi * a.k_gs;
        v16 += (int64_t)gi * a.v_gs;
    }
    k16 += (int64_t)hk * a.k_hs;
    v16 += (int64_t)hk * a.v_hs;

    const int kbeg = sp * a.split_len;
    int kend = min(L, kbeg + a.split_len);
    if (CAUSAL && a.per == 1) {
        // rows of this block only see keys <= iq_max + L - nq
        const int rmax = min(Mrows, rb * RWG + RWG) - 1;
        kend = min(kend, rmax / a.g + L - nq_eff + 1);
    }
    const int nkeys = kend > kbeg ? kend - kbeg : 0;
    // 32-key blocks per wave: its half of every 128-key tile (KG = 2) or all keys (KG = 1); even, the loop runs in pairs
    const int NB = KG == 2 ? 2 * ((nkeys + 127) >> 7) : 2 * ((nkeys + 63) >> 6);

    // ---- LDS map (bytes) ------------------------------------------------------------------------------------
    // KG = 2: KA[2 tiles x 64 rows] | KB[2 tiles x 64 rows] | V[2 tiles x 128 rows] | mlbuf.  KA: rows 0..63 of a
    // 128-key K tile (kg = 0 waves), KB: rows 64..127 (kg = 1 waves).  Ring slot s (block b, s = b & 3): tile buffer
    // s >> 1, rows [(s & 1) * 32, +32) of each 64-row half.
    // KG = 1: K[4 slots of 32 rows] | V[4 slots of 32 rows] | mlbuf.
    constexpr int KH_BYTES = 64 * RB;
    constexpr int V_BYTES = KG == 2 ? 128 * RB : 64 * RB;  // two ring slots of V
    constexpr int KA_OFF = 0, KB_OFF = 2 * KH_BYTES, V_OFF = KG == 2 ? 4 * KH_BYTES : 2 * KH_BYTES;
    typedef const __attribute__((address_space(3))) char* lptr_c;
    auto slot_k = [](int s) { return KG == 2 ? (s >> 1) * KH_BYTES + (s & 1) * 32 * RB : s * 32 * RB; };
    auto slot_v = [](int s) { return KG == 2 ? (s >> 1) * V_BYTES + (s & 1) * 32 * RB : s * 32 * RB; };

    // ---- staging: global -> LDS DMA, BROWS rows per tensor and iteration, NLB instructions per wave ----------
    // The LDS image of a wave instruction is lane-linear (base + lane * 16), so the XOR swizzles are applied to the
    // per-lane SOURCE chunk (involutions inside a row).  Instruction q = wave * NLB + i covers rows
    // rr = q * RPI + [0, RPI) of the staged rows: half h = rr >> 5 (KG = 2), row r32 = rr & 31.
    // ORDER OF THE PROLOGUE: a workgroup's first 4 us are ~300 set-up instructions and then the launch's cold burst (every
    // workgroup fetches its 32 KB of Q rows at once: 8 MB at ~5 TB/s; profiles/r05_prefix_timeline.md).  What the first
    // iteration waits for -- K block 0 and the Q rows -- is issued first, from the K half of the staging set-up alone; V's
    // offsets, the fragment addresses and the softmax state are computed under the burst (sched_barrier below: hipcc may not
    // hoist them back).  Stamps, start -> first barrier passed: C2 4.13 -> 3.81 us, C3 3.79 -> 3.58 us per workgroup.
    const int drow = (lane * 16) / RB;        // row inside the instruction
    const int dcp = ((lane * 16) % RB) >> 4;  // 16-byte slot inside the row (LDS side)
    unsigned koffb[NLB], voffb[NLB];  // per-lane source byte offsets relative to the block's first row
    unsigned kdst[NLB], vdst[NLB];    // wave-uniform LDS byte address of the instruction in ring slot 0 (piece i = piece 0 + 1024 i)
    const unsigned lds0 = (unsigned)(uintptr_t)(lptr_c)smem;
#pragma unroll
    for (int i = 0; i < NLB; ++i) {
        const int q = wave * NLB + i;
        const int rr = q * RPI + drow, h = KG == 2 ? rr >> 5 : 0, r32 = rr & 31;
        const int kch = D >= 128 ? (dcp ^ (r32 & 15)) : (dcp ^ ((r32 >> 1) & 7));
        const int drr = h * 64 + r32;  // row inside the 128-key tile (KG = 2) / the block (KG = 1)
        // piece i is issued with the immediate offset 1024 i (dma16w): drr >= RPI i and a row is >= RB bytes, so >= 0
        koffb[i] = (unsigned)(((int64_t)drr * a.k_ts + kch * 8) * 2) - 1024u * i;
        const int qh = KG == 2 ? (q * RPI) >> 5 : 0, qr = (q * RPI) & 31;  // wave-uniform
        kdst[i] = __builtin_amdgcn_readfirstlane(lds0 + (qh ? KB_OFF : KA_OFF) + qr * RB);
    }
    // One buffer resource per tensor for the whole pass: rows [kbeg, kend) of this (group, head); a block's first row
    // goes into the scalar offset.  Blocks / rows at or past the end read as zeros.
    const unsigned k_ts2 = (unsigned)(a.k_ts * 2), v_ts2 = (unsigned)(a.v_ts * 2);
    const u32x4 krs = make_rsrc_w(reinterpret_cast<const char*>(k16) + (int64_t)kbeg * a.k_ts * 2,
                                  nkeys > 0 ? (unsigned)(nkeys - 1) * k_ts2 + RB : 0u);
    auto row0_of = [&](int b) -> int { return KG == 2 ? (b >> 1) * 128 + (b & 1) * 32 : b * 32; };
    // a block whose first row is past the keys gets an offset past num_records: the whole instruction zero-fills
    auto soff_of = [&](int b, unsigned ts2) -> unsigned {
        const int r0 = row0_of(b);
        return __builtin_amdgcn_readfirstlane(r0 < nkeys + 128 ? (unsigned)r0 * ts2 : 0x7fff0000u);
    };
    auto dma_kblock = [&](int b) __attribute__((always_inline)) {
        dma_m0(kdst[0] + slot_k(b & 3));
        static_for<NLB>([&](auto I_) __attribute__((always_inline)) {
            constexpr int i = decltype(I_)::value;
            dma16w<1024 * i, true>(krs, koffb[i], soff_of(b, k_ts2));  // padded: the prologue's scalars may come straight from a v_readlane / v_readfirstlane
        });
    };
    if (NB > 0) dma_kblock(0);

    // ---- this lane's query rows (one per query block), fetched straight into AGPRs ---------------------------------
    // (B operands of the QK^T MFMAs for the whole kernel.)  Rows past the end are clamped to the last valid row: every
    // lane computes an independent query row, and an invalid lane never stores.
    // Row decode as a function of the lane index: evaluated here for the Q fetch and AGAIN in the epilogue from an opaque
    // copy of the lane index, so that the row's token / head / output offset are not carried across the pipeline (the
    // loop runs within a handful of registers of the 256-VGPR line; beyond it hipcc parks values in the asm-owned AGPRs).
    auto row_of = [&](int qb, int l31_, bool& valid, int& tok, int& hqv, int64_t& off) __attribute__((always_inline)) {
        const int r = rb * RWG + rw * WROWS + qb * 32 + l31_;
        valid = r < Mrows;
        const int rc = min(r, Mrows - 1);
        tok = (int)fdiv((unsigned)rc, a.div_g);  // query token inside the group
        hqv = hk * a.g + (rc - tok * a.g);
        off = ((int64_t)(q_tok0 + tok) * a.Hq + hqv) * D;
    };
    int row_lim[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) row_lim[qb] = 0x3fffffff;
    {
        const uint16_t* qrow_p[QB];
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            bool valid;
            int tok, hqv;
            int64_t off;
            row_of(qb, l31, valid, tok, hqv, off);
            if (CAUSAL) row_lim[qb] = (tok % nq_eff) + L - nq_eff;  // last visible key
            qrow_p[qb] = static_cast<const uint16_t*>(a.q) + off + 8 * hi;
        }
        static_for<QB * NC>([&](auto I_) {
            constexpr int I = decltype(I_)::value;
            regs.template qload<I, 32 * (I % NC)>(qrow_p[I / NC]);
        });
    }

    __builtin_amdgcn_sched_barrier(0);  // everything below is computed under the flight of K block 0 and Q
#pragma unroll
    for (int i = 0; i < NLB; ++i) {
        const int q = wave * NLB + i;
        const int rr = q * RPI + drow, h = KG == 2 ? rr >> 5 : 0, r32 = rr & 31;
        const int vs_ = D >= 128 ? (r32 & 3) : ((r32 >> 1) & 1);
        const int vch = (((dcp >> 2) ^ vs_) << 2) | (dcp & 3);
        const int drr = h * 64 + r32;
        voffb[i] = (unsigned)(((int64_t)drr * a.v_ts + vch * 8) * 2) - 1024u * i;
        const int qh = KG == 2 ? (q * RPI) >> 5 : 0, qr = (q * RPI) & 31;  // wave-uniform
        vdst[i] = __builtin_amdgcn_readfirstlane(lds0 + V_OFF + (qh * 64 + qr) * RB);
    }
    const u32x4 vrs = make_rsrc_w(reinterpret_cast<const char*>(v16) + (int64_t)kbeg * a.v_ts * 2,
                                  nkeys > 0 ? (unsigned)(nkeys - 1) * v_ts2 + RB : 0u);
    auto dma_block = [&](int b, bool isv) __attribute__((always_inline)) {
        dma_m0(isv ? vdst[0] + slot_v(b & 3) : kdst[0] + slot_k(b & 3));
        static_for<NLB>([&](auto I_) __attribute__((always_inline)) {
            constexpr int i = decltype(I_)::value;
            if (isv) dma16w<1024 * i, true>(vrs, voffb[i], soff_of(b, v_ts2));
            else dma16w<1024 * i, true>(krs, koffb[i], soff_of(b, k_ts2));
        });
    };
    if (NB > 0) {  // the rest of the cold start (see "Cold start" at the pipeline): behind K block 0 and Q, not waited for with them
        dma_block(1, false);
        dma_block(2, false);
        dma_block(0, true);
    }
    __builtin_amdgcn_sched_barrier(0);

    // ---- per-lane LDS byte addresses of the MFMA fragments (ring slot 0) -------------------------------
    const int ksw = D >= 128 ? (l31 & 15) : ((l31 >> 1) & 7);
    const int kx = hi ^ ksw;
    unsigned kaddr[NC];  // K fragment c of row l31 in slot 0 of this wave's key half (KA or KB)
#pragma unroll
    for (int c = 0; c < NC; ++c)
        kaddr[c] = (unsigned)(uintptr_t)(lptr_c)(smem + (kg ? KB_OFF : KA_OFF) + l31 * RB + (((2 * c) ^ kx) << 4));
    const int i16 = lane & 15, g16 = lane >> 4;
    const int vsw = D >= 128 ? (i16 >> 2) : ((i16 >> 3) & 1);
    unsigned vaddr[NDB];  // V^T fragment address in slot 0 for key slot 0 of this wave's half
#pragma unroll
    for (int db = 0; db < NDB; ++db)
        vaddr[db] = (unsigned)(uintptr_t)(lptr_c)(smem + V_OFF + (kg * 64 + 4 * hi + (i16 >> 2)) * RB +
                                                  ((db ^ vsw) << 6) + 32 * (g16 & 1) + 8 * (i16 & 3));

    // Online-softmax state per query block, in RAW score units (before the scale): the reference maximum m_raw (equal in
    // the two lanes of a row), nms = -sc * m_raw and the threshold thr = m_raw + kTau / sc above which a block's lane
    // maximum forces a new reference.  kMinit is a finite "minus infinity": exp2(sc * (m_old - m_new)) never sees inf - inf.
    // LAZY (the 8-wave unit): no maximum tree in the hot path at all.  A block's probabilities are computed against the
    // reference as it stands; if any lane's 16 of them sum to more than 2^kTau (which also catches an overflow to inf, and
    // -- nms starts at +1e30 sc -- the first valid key of a row), the wave takes the cold branch: true block maximum, new
    // reference = max(old, block), the block's probabilities recomputed from the raw scores (which the hot path leaves
    // intact: its fma / exp2 work on temporaries), O and l rescaled at the end of the iteration as before.  Below the
    // threshold every probability is <= its lane's sum <= 2^kTau: the same bound the threshold on the maximum gives.
    constexpr bool LAZY = NW == 8;
    constexpr float kMinit = -1.0e30f;
    float m_raw[QB], nms[QB], thr[QB], l_run[QB];
    f32x16 S0[QB], S1[QB];
    u32x4 P0[QB][2], P1[QB][2];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        m_raw[qb] = kMinit;
        thr[qb] = kMinit;
        nms[qb] = LAZY ? -kMinit * a.scale_log2e : 0.f;  // LAZY: exp2(score + huge) = inf trips the sum test at the row's first valid key
        l_run[qb] = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) S0[qb][i] = S1[qb][i] = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) P0[qb][i] = P1[qb][i] = u32x4{0u, 0u, 0u, 0u};
    }
    const float sc = a.scale_log2e;
    constexpr float kTau = 8.0f;
    const float tau_raw = kTau / sc;

    // ---- one pipeline iteration i: QK(i+1) | SM(i) | PV(i-1) -------------------------------------------
    // KOFF / VOFF: compile-time ring-slot byte offsets of K block i+1 and V block i-1;
    // NKOFF / NVOFF: the same for iteration i+1, whose first fragments are prefetched in the tail of this one.
    // FL: bit0 QK, bit1 SM, bit2 PV, bit3 the softmax may need masking, bit4 / bit5: iteration i+1 has QK / PV.
    // DM: 1 = issue the DMAs of K block i+4 and V block i+2 into ring slots kslot / vslot, spread between the MFMAs.
    // bvalid: block i exists.  Sw: scores written by QK; Sr: scores consumed by the softmax, which writes Pw; PV reads Pr.
    // kw: first key of block i for this wave (masking).
    constexpr int PDK = 3, PDV = 2;  // LDS prefetch distance (in fragment reads of the own stream)
    u32x4 kfr[PDK];
    u32x2 vfr[PDV][2];
#pragma unroll
    for (int c = 0; c < PDK; ++c) kfr[c] = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
    for (int p = 0; p < PDV; ++p) vfr[p][0] = vfr[p][1] = u32x2{0u, 0u};
    auto ldk_at = [&](int c, int off) -> u32x4 {
        return *reinterpret_cast<const __attribute__((address_space(3))) u32x4*>((uintptr_t)(kaddr[c] + off));
    };
    auto ldv_at = [&](int p, int h, int off) -> u32x2 {  // PV MFMA p = ks * NDB + db; h = 8-key half of the slot
        return lds_tr16_w(vaddr[p % NDB] + off + (16 * (p / NDB) + 8 * h) * RB);
    };
    auto iter = [&](auto KOFF_C, auto VOFF_C, auto NKOFF_C, auto NVOFF_C, auto FL_C, auto DM_C, f32x16(&Sw)[QB],
                    f32x16(&Sr)[QB], u32x4(&Pw)[QB][2], u32x4(&Pr)[QB][2], int kw, bool bvalid, unsigned ksoff,
                    unsigned vsoff, int kslot, int vslot) __attribute__((always_inline)) {
        constexpr int KOFF = decltype(KOFF_C)::value;
        constexpr int VOFF = decltype(VOFF_C)::value;
        constexpr int NKOFF = decltype(NKOFF_C)::value;
        constexpr int NVOFF = decltype(NVOFF_C)::value;
        constexpr int FL = decltype(FL_C)::value;
        constexpr int DM = decltype(DM_C)::value;
        constexpr bool QK = FL & 1, SM = FL & 2, PV = FL & 4, MASK = (FL & 8) || CAUSAL;
        constexpr bool NQK = FL & 16, NPVF = FL & 32;
        constexpr int NPV = 2 * NDB;        // PV fragment steps (2 key slots x NDB d blocks)
        constexpr int NSLOT = NC + NPV;     // MFMA slots (QB MFMAs each); QK and PV alternate
        constexpr int NG = 16;              // VALU groups of one query block's softmax
        auto ldk = [&](int c) -> u32x4 { return ldk_at(c, KOFF); };
        auto ldv = [&](int p, int h) -> u32x2 { return ldv_at(p, h, VOFF); };
        // softmax state of this iteration
        float t0[QB], t1[QB], t2[QB], t3[QB], t4[QB], tmax[QB], alpha[QB], su0[QB], su1[QB];
        uint64_t upb[QB];
        bool pend = false;  // a new reference maximum was adopted in this iteration: O and l are rescaled at its end
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            t0[qb] = t1[qb] = t2[qb] = t3[qb] = t4[qb] = tmax[qb] = su0[qb] = su1[qb] = 0.f;
            // alpha is only written by the cold branch that sets `pend` and only read under `pend`: no per-iteration 1.0
            upb[qb] = 0ull;
        }
        if constexpr (SM && MASK) {
            bool need_mask = !bvalid || (kw + 32 > kend);
            if (CAUSAL) need_mask = need_mask || __builtin_amdgcn_ballot_w64(min(row_lim[0], row_lim[QB - 1]) < kw + 31) != 0ull;
            if (need_mask) {
                asm volatile("" ::: "memory");  // keep this a (cold) branch
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) {
                    int lim = kend - 1;
                    if (CAUSAL) lim = min(lim, row_lim[qb]);
                    const int lr = bvalid ? lim - kw - 4 * hi : -1;
#pragma unroll
                    for (int i = 0; i < 16; ++i)
                        if (8 * (i >> 2) + (i & 3) > lr) Sr[qb][i] = -INFINITY;
                }
            }
        }
        // Softmax of one query block as NG = 16 groups of 4-5 VALU instructions, one group behind each of the block's 16
        // MFMAs (an MFMA's shadow holds ~5 issue slots; every instruction beyond it costs its full issue time).
        //   g0-g1 lane-local maximum of the 16 raw scores (max3 tree) and ONE compare against the threshold.  Only when
        //         some lane exceeds it (cold, wave-uniform) are the half-wave exchange, the new reference, alpha and the
        //         O / l rescale executed -- the hot path never touches m, nms or alpha.
        //   g2-g15: 54 element operations in the order of kElemOps, five issue slots per group (scalar f32 ops on purpose: packed
        //         f32 VALU beside MFMAs costs more than the two scalar instructions it replaces).
        float Y[QB][16];  // LAZY: the block's scaled scores, then probabilities (the raw scores stay in Sr for the cold branch)
        auto elem_op = [&](auto K_, int qb) __attribute__((always_inline)) {
            f32x16& S = Sr[qb];
            constexpr int code = LAZY ? kElemOpsL[decltype(K_)::value] : kElemOps[decltype(K_)::value];
            constexpr int kind = code >> 3, pr = code & 7, e0 = 2 * pr, e1 = 2 * pr + 1;
            if constexpr ((ABL & 64) || ((ABL & 32) && (kind == 2 || kind == 3)) || ((ABL & 128) && kind >= 4))
                return;
            else if constexpr (LAZY) {
                float(&y)[16] = Y[qb];
                if constexpr (kind == 0) { y[e0] = __builtin_fmaf(S[e0], sc, nms[qb]); asm volatile("" : "+v"(y[e0])); }
                else if constexpr (kind == 1) { y[e1] = __builtin_fmaf(S[e1], sc, nms[qb]); asm volatile("" : "+v"(y[e1])); }
                else if constexpr (kind == 2) { y[e0] = fast_exp2(y[e0]); asm volatile("" : "+v"(y[e0])); }
                else if constexpr (kind == 3) { y[e1] = fast_exp2(y[e1]); asm volatile("" : "+v"(y[e1])); }
                else if constexpr (kind == 4) { if constexpr (pr == 1) su0[qb] = y[0] + y[2]; else su0[qb] += y[e0]; asm volatile("" : "+v"(su0[qb])); }
                else if constexpr (kind == 5) { if constexpr (pr == 1) su1[qb] = y[1] + y[3]; else su1[qb] += y[e1]; asm volatile("" : "+v"(su1[qb])); }
                else {
                    Pw[qb][pr >> 2][pr & 3] = TR::pack2(y[e0], y[e1]);
                    asm volatile("" ::"v"(Pw[qb][pr >> 2][pr & 3]));
                }
            }
            else if constexpr (kind == 0) { S[e0] = __builtin_fmaf(S[e0], sc, nms[qb]); asm volatile("" : "+v"(S[e0])); }
            else if constexpr (kind == 1) { S[e1] = __builtin_fmaf(S[e1], sc, nms[qb]); asm volatile("" : "+v"(S[e1])); }
            else if constexpr (kind == 2) { S[e0] = fast_exp2(S[e0]); asm volatile("" : "+v"(S[e0])); }
            else if constexpr (kind == 3) { S[e1] = fast_exp2(S[e1]); asm volatile("" : "+v"(S[e1])); }
            // row sums: the block's first add is S[0] + S[2] (one instruction when pair 1 comes by), not 0 + S[0] and then + S[2]
            else if constexpr (kind == 4) { if constexpr (pr == 1) su0[qb] = S[0] + S[2]; else su0[qb] += S[e0]; asm volatile("" : "+v"(su0[qb])); }
            else if constexpr (kind == 5) { if constexpr (pr == 1) su1[qb] = S[1] + S[3]; else su1[qb] += S[e1]; asm volatile("" : "+v"(su1[qb])); }
            else {  // element pair pr -> P^T slot (pr >> 2), word (pr & 3)
                Pw[qb][pr >> 2][pr & 3] = TR::pack2(S[e0], S[e1]);
                asm volatile("" ::"v"(Pw[qb][pr >> 2][pr & 3]));
            }
        };
        auto valu_group = [&](auto G_, int qb) __attribute__((always_inline)) {
            constexpr int g = decltype(G_)::value;
            if constexpr (SM && !(ABL & 4) && LAZY) {
                constexpr int k0 = kElemGroupStartL[g], k1 = kElemGroupStartL[g + 1];
                static_for<k1 - k0>([&](auto K_) __attribute__((always_inline)) {
                    elem_op(std::integral_constant<int, k0 + decltype(K_)::value>{}, qb);
                });
            } else if constexpr (SM && !(ABL & 4)) {
                f32x16& S = Sr[qb];
                if constexpr (g == 0) {
                    t0[qb] = fmaxf(fmaxf(S[0], S[1]), S[2]);
                    t1[qb] = fmaxf(fmaxf(S[3], S[4]), S[5]);
                    t2[qb] = fmaxf(fmaxf(S[6], S[7]), S[8]);
                    t3[qb] = fmaxf(fmaxf(S[9], S[10]), S[11]);
                    t4[qb] = fmaxf(fmaxf(S[12], S[13]), S[14]);
                    asm volatile("" ::"v"(t0[qb]), "v"(t1[qb]), "v"(t2[qb]), "v"(t3[qb]), "v"(t4[qb]));
                } else if constexpr (g == 1) {
                    t0[qb] = fmaxf(fmaxf(t0[qb], t1[qb]), S[15]);
                    t2[qb] = fmaxf(fmaxf(t2[qb], t3[qb]), t4[qb]);
                    tmax[qb] = fmaxf(t0[qb], t2[qb]);
                    upb[qb] = __builtin_amdgcn_ballot_w64(tmax[qb] > thr[qb]);  // lanes above the threshold (a scalar pair: no VALU select)
                    asm volatile("" ::"v"(tmax[qb]));
                } else {
                    constexpr int k0 = kElemGroupStart[g - 2], k1 = kElemGroupStart[g - 1];
                    static_for<k1 - k0>([&](auto K_) __attribute__((always_inline)) {
                        elem_op(std::integral_constant<int, k0 + decltype(K_)::value>{}, qb);
                    });
                }
            }
        };
        constexpr bool BF = std::is_same<T, BF16>::value;
        // An in-order wave that issues two MFMAs back to back sits at the second one until the matrix pipe frees (32
        // cycles) and nothing behind it issues: only ONE MFMA's shadow per pair would carry VALU work (measured: the
        // softmax then adds its full issue time to the loop).  So every MFMA is followed by its own share of the other
        // streams: MFMA(qb 0) | softmax group of qb 0 | MFMA(qb 1) | softmax group of qb 1 | LDS prefetch, DMA.
        static_for<NSLOT>([&](auto J_) __attribute__((always_inline)) {
            constexpr int j = decltype(J_)::value;
            // MFMAs of this slot: even -> QK chunk j/2, odd -> PV step j/2; one fragment, QB MFMAs
            constexpr int idx = j >> 1;
            constexpr bool isqk = (j & 1) == 0;
            u32x4 vf = {0u, 0u, 0u, 0u};
            if constexpr (!isqk && PV) vf = u32x4{vfr[idx % PDV][0][0], vfr[idx % PDV][0][1], vfr[idx % PDV][1][0], vfr[idx % PDV][1][1]};
            static_for<QB>([&](auto B_) __attribute__((always_inline)) {
                constexpr int qb = decltype(B_)::value;
                if constexpr (isqk) {
                    if constexpr (QK) regs.template qk<qb * NC + idx, BF, idx == 0>(Sw[qb], kfr[idx % PDK]);
                } else {
                    if constexpr (PV) regs.template pv<qb * NDB + idx % NDB, BF>(vf, Pr[qb][idx / NDB]);
                }
                __builtin_amdgcn_sched_barrier(0);
                constexpr int g0 = (j * NG) / NSLOT, g1 = ((j + 1) * NG) / NSLOT;
                static_for<g1 - g0>([&](auto G_) __attribute__((always_inline)) {
                    valu_group(std::integral_constant<int, g0 + decltype(G_)::value>{}, qb);
                });
                __builtin_amdgcn_sched_barrier(0);
            });
            if constexpr (SM && !(ABL & 4) && !LAZY && (j * NG) / NSLOT <= 1 && 1 < ((j + 1) * NG) / NSLOT) {
                // both query blocks know their lane maxima: adopt a new reference maximum?  (cold, wave-uniform)
                if ((upb[0] | upb[QB - 1]) != 0ull) {
                    asm volatile("" ::: "memory");  // keep this a branch
#pragma unroll
                    for (int qb = 0; qb < QB; ++qb) {
                        const float tm = pair_max(tmax[qb]);  // the row's maximum: both lanes of a row decide alike
                        const float newm = tm > thr[qb] ? tm : m_raw[qb];
                        alpha[qb] = fast_exp2((m_raw[qb] - newm) * sc);
                        m_raw[qb] = newm;
                        nms[qb] = -newm * sc;
                        thr[qb] = newm + tau_raw;
                    }
                    pend = true;
                }
            }
            if constexpr (isqk) {
                if constexpr (QK) {
                    if (idx + PDK < NC && !(ABL & 8)) kfr[idx % PDK] = ldk(idx + PDK);
                }
            } else {
                if constexpr (PV) {
                    if (idx + PDV < NPV && !(ABL & 8)) { vfr[idx % PDV][0] = ldv(idx + PDV, 0); vfr[idx % PDV][1] = ldv(idx + PDV, 1); }
                }
            }
            if constexpr (DM == 1 && !(ABL & 1)) {
                // 2 * NLB DMA instructions spread over the iteration (odd slots first)
                constexpr int EVERY = NSLOT / (2 * NLB) > 1 ? 2 : 1;
                if constexpr ((j % EVERY) == EVERY - 1 && j / EVERY < 2 * NLB) {
                    constexpr int i = j / EVERY;
                    if constexpr (i == 0) dma_m0(kdst[0] + kslot);
                    if constexpr (i == NLB) dma_m0(vdst[0] + vslot);
                    if constexpr (i < NLB) dma16w<1024 * i, PERSIST>(krs, koffb[i], ksoff);
                    else dma16w<1024 * (i - NLB), PERSIST>(vrs, voffb[i - NLB], vsoff);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        // first fragments of iteration i+1 (its blocks are already visible, see the header)
        if constexpr (NQK) {
#pragma unroll
            for (int c = 0; c < PDK; ++c) kfr[c] = ldk_at(c, NKOFF);
        }
        if constexpr (NPVF) {
#pragma unroll
            for (int p = 0; p < PDV; ++p) { vfr[p][0] = ldv_at(p, 0, NVOFF); vfr[p][1] = ldv_at(p, 1, NVOFF); }
        }
        if constexpr (SM && !(ABL & 4) && LAZY) {
            static_assert(!LAZY || QB == 1, "the 8-wave unit runs one query block per wave");
            const float bsum = su0[0] + su1[0];
            if (__builtin_amdgcn_ballot_w64(bsum > 256.0f /* 2^kTau */) != 0ull) {  // cold, wave-uniform
                asm volatile("" ::: "memory");  // keep this a branch
                f32x16& S = Sr[0];
                float tm = fmaxf(fmaxf(fmaxf(S[0], S[1]), fmaxf(S[2], S[3])), fmaxf(fmaxf(S[4], S[5]), fmaxf(S[6], S[7])));
                tm = fmaxf(tm, fmaxf(fmaxf(fmaxf(S[8], S[9]), fmaxf(S[10], S[11])), fmaxf(fmaxf(S[12], S[13]), fmaxf(S[14], S[15]))));
                tm = pair_max(tm);                            // the row's maximum: both lanes of a row decide alike
                const float newm = fmaxf(tm, m_raw[0]);       // (tm = -inf: every key of the block masked for this row)
                alpha[0] = fast_exp2((m_raw[0] - newm) * sc);
                m_raw[0] = newm;
                nms[0] = -newm * sc;
                float r0 = 0.f, r1 = 0.f;
#pragma unroll
                for (int pr = 0; pr < 8; ++pr) {
                    const float pa = fast_exp2(__builtin_fmaf(S[2 * pr], sc, nms[0]));
                    const float pb = fast_exp2(__builtin_fmaf(S[2 * pr + 1], sc, nms[0]));
                    r0 += pa;
                    r1 += pb;
                    Pw[0][pr >> 2][pr & 3] = TR::pack2(pa, pb);
                }
                su0[0] = r0;
                su1[0] = r1;
                pend = true;
            }
        }
        if constexpr (SM && !(ABL & 4)) {
            // cold: at most a handful of times per row block.  Every PV(i-1) MFMA has been issued: all of O and l is still
            // at the old reference and is rescaled exactly once; P(i) is at the new one.
            regs.template rescale<QB, NDB>(pend, alpha);
            if (pend) {
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) l_run[qb] *= alpha[qb];
            }
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) {  // pinned scalar adds: left alone, hipcc packs the two blocks' sums into v_pk_add_f32 (+ an s_nop)
                float t = su0[qb] + su1[qb];
                asm volatile("" : "+v"(t));
                l_run[qb] += t;
                asm volatile("" : "+v"(l_run[qb]));
            }
        }
    };

    // ---- pipeline over 32-key blocks ---------------------------------------------------------------------
    using std::integral_constant;
#define HYD_IC(x) integral_constant<int, (x)>{}
    const int kwave = kbeg + kg * 64;  // first key of this wave's half of tile 0
    // Iterations i = -1 .. NB; the pipeline's fill (i = -1: QK(0) only; i = 0: QK(1) + softmax(0)) and drain
    // (i = NB-1: softmax + PV, no QK; i = NB: PV(NB-1) only) run their own, shorter instantiations of the iteration.
    // Cold start (issued in the prologue, above): the first iteration (QK(0) alone) needs the Q fragments and K block 0 only.  K blocks 1, 2 and V
    // block 0 are issued behind them and the wait leaves those three blocks in flight (vector-memory operations retire
    // in issue order): the prologue is a bandwidth burst of every workgroup of the launch at once (~11 B / cycle / CU),
    // so every KiB not waited for is ~90 cycles.  The counted wait that ends iteration -1 covers them; that iteration
    // therefore does not prefetch block 1's first fragments in its tail (FL_FIRST), they are read behind its barrier.
    static_for<QB * NDB>([&](auto I_) { regs.template zero<decltype(I_)::value>(); });  // under the first loads' flight
    stampk(1);
    if (NB > 0) dma_wait_w<3 * NLB>();
    else dma_wait_w<0>();
    __syncthreads();
    stampk(2);
    // Static issue priority for one of the two waves that share a SIMD (MI355X_MICROARCH.md, "two waves per SIMD", item 4): waves w
    // and w + 4 of a workgroup sit on the same SIMD; raised once in front of the key loop, no flips inside it.  Development
    // builds: ABL bit 13 = the younger half (waves 4-7), bit 14 = the older half (A/B in profiles/r06_prefix_setprio.md).
    if constexpr (NW == 8 && (ABL & (8192 | 16384)) != 0) {
        if (((ABL & 8192) != 0) == (wave >= 4)) __builtin_amdgcn_s_setprio(1);
    }
    if (NB > 0) {
#pragma unroll
        for (int c = 0; c < PDK; ++c) kfr[c] = ldk_at(c, slot_k(0));
        // ii = i0 + R = -1 + R (mod 4), so every ring slot is a compile-time constant:
        //   reads  K block ii+1 -> slot R, V block ii-1 -> slot (R+2)&3;  next iteration: (R+1)&3, (R+3)&3
        //   writes K block ii+4 -> slot (R+3)&3, V block ii+2 -> slot (R+1)&3.      ii odd <=> R even.
        // FL: 1 QK | 2 softmax | 4 PV | 8 masking possible | 16 / 32: the next iteration has QK / PV (fragment prefetch)
        constexpr int FL_FULL = 7 + 8 + 16 + 32, FL_FIRST = 1, FL_SECOND = 1 + 2 + 8 + 16 + 32,
                      FL_PENULT = 2 + 4 + 8 + 32, FL_LAST = 4;
        // Scalars that advance by one 32-key block per iteration (kept incremental: the loop is issue-bound and every
        // scalar instruction between two MFMAs costs a slot): byte offset of K block ii+4 and of V block ii+2 inside
        // their buffer resources (blocks past the keys only ever exceed num_records: zero fill), first key of block ii.
        // Block b -> b+1 advances 32 rows, or 96 when b is odd and the key halves interleave (KG = 2).
        const unsigned k_lo = 32u * k_ts2, v_lo = 32u * v_ts2;
        const unsigned k_hi = KG == 2 ? 96u * k_ts2 : k_lo, v_hi = KG == 2 ? 96u * v_ts2 : v_lo;
        unsigned kso = (unsigned)row0_of(3) * k_ts2, vso = (unsigned)row0_of(1) * v_ts2;
        int kwv = kwave + (KG == 2 ? -96 : -32);
#define HYD_IT(R, FLV, SW, SR, PW, PR)                                                                       \
    {                                                                                                        \
        iter(HYD_IC(slot_k((R) & 3)), HYD_IC(slot_v(((R) + 2) & 3)), HYD_IC(slot_k(((R) + 1) & 3)),          \
             HYD_IC(slot_v(((R) + 3) & 3)), HYD_IC(FLV), HYD_IC(1), SW, SR, PW, PR, kwv, true,               \
             __builtin_amdgcn_readfirstlane(kso), __builtin_amdgcn_readfirstlane(vso),                        \
             slot_k(((R) + 3) & 3), slot_v(((R) + 1) & 3));                                                  \
        /* ii = i0 + R is odd <=> R even: blocks ii, ii+2, ii+4 share its parity */                          \
        kso += ((R) & 1) ? k_lo : k_hi;                                                                      \
        vso += ((R) & 1) ? v_lo : v_hi;                                                                      \
        kwv += ((R) & 1) ? 32 : (KG == 2 ? 96 : 32);                                                         \
        dma_wait_w<2 * NLB>();                                                                               \
        if (!(ABL & 16)) __builtin_amdgcn_s_barrier();                                                       \
    }
        int i0 = -1;
        HYD_IT(0, FL_FIRST, S0, S1, P1, P0)
#pragma unroll
        for (int c = 0; c < PDK; ++c) kfr[c] = ldk_at(c, slot_k(1));  // K block 1 is visible only now (cold start, above)
        HYD_IT(1, FL_SECOND, S1, S0, P0, P1)
        for (;;) {
            if (i0 + 4 >= NB) {
                HYD_IT(2, FL_PENULT, S0, S1, P1, P0)
                HYD_IT(3, FL_LAST, S1, S0, P0, P1)
                break;
            }
            HYD_IT(2, FL_FULL, S0, S1, P1, P0)
            HYD_IT(3, FL_FULL, S1, S0, P0, P1)
            i0 += 4;
            if (i0 + 2 >= NB) {
                HYD_IT(0, FL_PENULT, S0, S1, P1, P0)
                HYD_IT(1, FL_LAST, S1, S0, P0, P1)
                break;
            }
            HYD_IT(0, FL_FULL, S0, S1, P1, P0)
            HYD_IT(1, FL_FULL, S1, S0, P0, P1)
        }
#undef HYD_IT
        dma_wait_w<0>();
        __syncthreads();  // nothing in flight, everyone done with the rings before the merge reuses them
    }
    regs.drain();
    if constexpr (NW == 8 && (ABL & (8192 | 16384)) != 0) __builtin_amdgcn_s_setprio(0);
    stampk(3);
#undef HYD_IC

    // ---- merge the two key halves through LDS (KG = 2), normalise, store ------------------------------------
    // Both waves of a pair (same rows, key half 0 / 1) take part: the wave of key half h finalises the d blocks
    // [h * NDB/2, (h+1) * NDB/2) of both query blocks and hands the other d blocks (+ its m, l) to its partner through
    // LDS.  A lane holds 4 consecutive d per (d block, q4); v_permlane32_swap pairs the two half-waves' groups so that
    // each lane stores 16 contiguous bytes (row-per-lane stores are issue-bound).
    constexpr int HDB = KG == 2 ? NDB / 2 : NDB;  // d blocks this wave finalises
    float l_tot[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) l_tot[qb] = pair_sum(l_run[qb]);
    f32x4* obuf = reinterpret_cast<f32x4*>(smem);  // [NW waves][QB][HDB * 4][64 lanes] of f32x4 (64 KB at D = 128)
    if constexpr (KG == 2) {
        static_for<QB * HDB>([&](auto I_) __attribute__((always_inline)) {
            constexpr int qb = decltype(I_)::value / HDB, i = decltype(I_)::value % HDB;
            float ob[16];  // the d block this wave hands to its partner
            if (kg) regs.template read<qb * NDB + i>(ob);
            else regs.template read<qb * NDB + HDB + i>(ob);
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                f32x4 x = {ob[4 * q4], ob[4 * q4 + 1], ob[4 * q4 + 2], ob[4 * q4 + 3]};
                obuf[((wave * QB + qb) * HDB * 4 + i * 4 + q4) * 64 + lane] = x;
            }
        });
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            mlbuf[(wave * QB + qb) * 128 + lane] = l_tot[qb] > 0.f ? m_raw[qb] * sc : -INFINITY;
            mlbuf[(wave * QB + qb) * 128 + 64 + lane] = l_tot[qb];
        }
        __syncthreads();
    }
    stampk(4);
    const int pw = wave ^ NRW;  // partner wave (KG = 2): same rows, other key half
    int lane_e = threadIdx.x & 31;
    asm volatile("" : "+v"(lane_e));  // opaque: the row decode below is recomputed, not kept live across the pipeline
    static_for<QB>([&](auto B_) __attribute__((always_inline)) {
        constexpr int qb = decltype(B_)::value;
        bool rvalid_q;
        int rtok_q, hq_q;
        int64_t row_off_q;
        row_of(qb, lane_e, rvalid_q, rtok_q, hq_q, row_off_q);
        const float m1 = KG == 2 ? mlbuf[(pw * QB + qb) * 128 + lane] : -INFINITY;
        const float l1 = KG == 2 ? mlbuf[(pw * QB + qb) * 128 + 64 + lane] : 0.f;
        // base-2 exponent units.  Opaque to the optimiser: with the product folded into `m_own - mfs` as an fma, a wave
        // that saw no key (m_raw = kMinit, e.g. the empty split of a short ragged group) would compute exp2 of the
        // PRODUCT'S ROUNDING ERROR (~1e21) instead of exp2(0) -- inf * 0 = NaN for scales whose error is positive
        // (found at D = 256).  A wave without keys also reports -inf, so that it drops out of the merge exactly.
        float m_own = m_raw[qb] * sc;
        asm volatile("" : "+v"(m_own));
        if (!(l_tot[qb] > 0.f)) m_own = -INFINITY;
        const float mf = fmaxf(m_own, m1);
        const float mfs = (mf == -INFINITY) ? 0.f : mf;
        const float a0 = fast_exp2(m_own - mfs), a1 = fast_exp2(m1 - mfs);
        const float lf = l_tot[qb] * a0 + l1 * a1;
        const float inv = lf > 0.f ? 1.0f / lf : 0.f;
        const float w0 = a0 * inv, w1 = a1 * inv;
        const int64_t obase = (int64_t)sp * a.out_split_stride + row_off_q;
        static_for<HDB>([&](auto D_) __attribute__((always_inline)) {
            constexpr int i = decltype(D_)::value;
            float ob[16];  // a d block this wave finalises
            if (KG == 2 && kg) regs.template read<qb * NDB + NDB - HDB + i>(ob);
            else regs.template read<qb * NDB + i>(ob);
            const int db = KG == 2 ? kg * HDB + i : i;
            f32x4 x[4];
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                if constexpr (KG == 2) {
                    const f32x4 y = obuf[((pw * QB + qb) * HDB * 4 + i * 4 + q4) * 64 + lane];
#pragma unroll
                    for (int j = 0; j < 4; ++j) x[q4][j] = ob[4 * q4 + j] * w0 + y[j] * w1;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) x[q4][j] = ob[4 * q4 + j] * w0;
                }
            }
            if (a.out_f32) {
                if (rvalid_q) {
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4)
                        *reinterpret_cast<f32x4*>(static_cast<float*>(a.out) + obase + 32 * db + 8 * q4 + 4 * hi) = x[q4];
                }
            } else {
#pragma unroll
                for (int qp = 0; qp < 2; ++qp) {
                    // groups (2qp, 2qp+1): after the swaps the lower half-wave holds d [8*(2qp), +8), the upper [8*(2qp+1), +8)
                    u32x4 w;
#pragma unroll
                    for (int dw = 0; dw < 2; ++dw) {
                        const unsigned ea = TR::pack2(x[2 * qp][2 * dw], x[2 * qp][2 * dw + 1]);
                        const unsigned eb = TR::pack2(x[2 * qp + 1][2 * dw], x[2 * qp + 1][2 * dw + 1]);
                        auto r2 = __builtin_amdgcn_permlane32_swap(ea, eb, false, false);
                        w[dw] = r2[0];      // lower half: own group 2qp      | upper half: lower's group 2qp+1
                        w[2 + dw] = r2[1];  // lower half: upper's group 2qp  | upper half: own group 2qp+1
                    }
                    if (rvalid_q)
                        *reinterpret_cast<u32x4*>(static_cast<uint16_t*>(a.out) + obase + 32 * db + 8 * (2 * qp + hi)) = w;
                }
            }
        });
        if (a.lse && kg == 0 && hi == 0 && rvalid_q) {
            const float lse = lf > 0.f ? mf * kLn2 + __logf(lf) : -INFINITY;
            int64_t idx;
            if (a.lse_layout == HYD_LSE_BQH)
                idx = (int64_t)(q_tok0 + rtok_q) * a.Hq + hq_q;
            else
                idx = ((int64_t)gi * a.Hq + hq_q) * a.lse_q_stride + rtok_q;
            a.lse[(int64_t)sp * a.lse_split_stride + idx] = lse;
        }
    });
    if constexpr ((ABL & 2048) != 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stampk(5);
        if (wave == 0 && lane < 12) {  // record of this workgroup: 16 words behind the LSEs
            unsigned xcc, hwid;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)\n\ts_getreg_b32 %1, hwreg(HW_REG_HW_ID)" : "=s"(xcc), "=s"(hwid));
            unsigned tv = tst[0];
            for (int q_ = 1; q_ < 8; ++q_) tv = lane == q_ ? tst[q_] : tv;
            tv = lane == 8 ? xcc : lane == 9 ? hwid : lane == 10 ? (unsigned)blockIdx.x : lane == 11 ? (unsigned)lin : tv;
            const size_t tail = a.nsplit > 1 ? (size_t)a.nsplit * (size_t)a.lse_split_stride : (size_t)a.B * a.nq * a.Hq;  // behind the LSEs
            reinterpret_cast<unsigned*>(a.lse)[tail + (size_t)blockIdx.x * 16 + lane] = tv;
        }
    }
}

}  // namespace hyd
