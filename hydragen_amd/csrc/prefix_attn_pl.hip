// Prefix pass, software-pipelined kernel ("pl"): batched-query attention of all queries of a group
// against the group's single shared K/V, on the gfx950 matrix cores.
//
// Replaces flash-attn's _flash_attn_forward / _flash_attn_varlen_forward as called from
// /root/reference/hydragen/flash.py:284-351 and hydragen/attention.py:270,313,344.
//
// Same work decomposition, LDS image and epilogue as prefix_attn.hip (512 threads = 4 row blocks x 2 key
// halves, 128 query rows per workgroup, S^T = K.Q^T and O^T += V^T.P^T with 32x32x16 MFMAs), but the
// loop is built around what the MI355X probes showed (tests/probes/coexec*_probe.hip, instream_probe.hip):
// VALU work hides in the shadow of MFMAs only when it sits in the SAME wave's instruction stream (~16 of
// the 32 cycles of an MFMA), never across waves, and a global->LDS DMA that is waited for right after it
// was issued costs its full L2 latency.  So:
//   * every wave runs a three-stage pipeline over 32-key blocks b; iteration i interleaves, instruction by
//     instruction,   MFMA stream: QK(i+1) and PV(i-1), alternating   |   VALU stream: online softmax of
//     block i (max, exp2, row sum, pack to 16-bit P^T).  The order is pinned by sched_barriers (machine
//     scheduler) and input-only asm anchors (IR passes).
//   * the running maximum is only raised when a block exceeds it by more than kTau (log2 units): the
//     rescale of the O accumulator becomes a cold wave-uniform branch instead of 64 multiplies per block
//     (softmax is shift invariant; P <= 2^kTau stays well inside fp16).
//   * K and V live in rings of four 32-key block slots.  Iteration i DMAs K block i+4 and V block i+2
//     into the slots whose last readers finished before the barrier that ended iteration i-1, and the
//     barrier ending iteration i only waits for the DMAs issued during iteration i-1 (s_waitcnt vmcnt(N)):
//     no fetch latency is exposed, one barrier per iteration.  Everything iteration i+1 reads is therefore
//     already visible during iteration i, whose tail prefetches the first LDS fragments of iteration i+1.
#include <type_traits>

#include "hyd_kernels.h"

namespace hyd {

typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr_pl;

__device__ __forceinline__ u32x2 lds_tr16_pl(unsigned lds_byte_addr) {
    s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr_pl)(uintptr_t)lds_byte_addr);
    return __builtin_bit_cast(u32x2, t);
}

// One LDS-DMA instruction: 64 lanes x 16 B from global memory to LDS [lds_dst, lds_dst + 1 KiB) (lds_dst
// wave-uniform, the LDS image is lane-linear: lane l lands at lds_dst + 16 l, tests/probes/bufdma_probe.hip).
// Buffer form: the source is a raw buffer resource (wave-uniform base + num_records bytes) plus a per-lane
// unsigned byte offset; lanes whose offset is >= num_records write zeros instead of faulting, which is how
// rows past the end of the keys (and whole blocks that do not exist) are handled without address math.
// Issued from inline asm on purpose: hipcc then does not know an LDS write is in flight, so it neither
// drains it (vmcnt(0)) in front of the LDS reads of the current blocks nor counts it; the loop waits for
// it explicitly (dma_wait<N>) before its barrier.
__device__ __forceinline__ void dma16b(u32x4 rsrc, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(rsrc), "s"(lds_dst)
                 : "memory");
}
// raw buffer resource over [base, base + bytes) (gfx9 word 3: 32-bit data format, no swizzle)
__device__ __forceinline__ u32x4 make_rsrc(const char* base, unsigned bytes) {
    const uint64_t b = (uint64_t)(uintptr_t)base;
    u32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((unsigned)b);
    r[1] = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32) & 0xffffu);
    r[2] = __builtin_amdgcn_readfirstlane(bytes);
    r[3] = 0x00020000u;
    return r;
}
template <int N>
__device__ __forceinline__ void dma_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory"); }

// ABL: development-only timing ablations (bit0 no in-loop DMA, bit1 no exp groups, bit2 no softmax VALU at all,
// bit3 no LDS fragment reads, bit4 no barrier in the fast loop); only ABL = 0 is ever used for results.
// KG: key groups per 128-key tile.  KG = 2 (default): 128 rows per workgroup, wave = (row block, key half).  KG = 1: 256
// rows per workgroup, every wave walks all keys 32 at a time -- half the K/V staging per flop and no cross-half merge,
// for shapes with enough rows per kv head to fill the chip with 256-row workgroups (D = 128 only).
template <typename T, int D, bool CAUSAL, int ABL = 0, int KG = 2>
__global__ __launch_bounds__(512) void prefix_attn_pl_kernel(const PrefixArgs a) {
    using TR = Traits<T>;
    constexpr int RB = D * 2;            // bytes per K/V row
    constexpr int NC = D / 16;           // k-chunks of the QK^T contraction
    constexpr int NDB = D / 32;          // 32-wide d blocks of O^T
    constexpr int NLB = KG == 2 ? RB / 128 : 1;  // DMA instructions per wave per 32-key block per tensor
    static_assert(KG == 2 || D == 128, "KG = 1 is instantiated for D = 128 only (one DMA instruction per wave and block)");
    constexpr int RWG = KG == 2 ? 128 : 256;  // query rows per workgroup
    constexpr int RPI = 1024 / RB;       // rows per DMA instruction
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* mlbuf = reinterpret_cast<float*>(smem + (KG == 2 ? 512 : 256) * RB);  // [8 waves][2][64], after the K/V rings

    unsigned tst[6] = {0, 0, 0, 0, 0, 0};
    auto stampk = [&](int k) __attribute__((always_inline)) {
        if constexpr ((ABL & 64) != 0) {
            uint64_t t;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
            tst[k] = (unsigned)t;
        }
    };
    stampk(0);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kg = KG == 2 ? wave >> 2 : 0;  // key half of every 128-key tile this wave computes on
    const int rw = KG == 2 ? wave & 3 : wave;  // 32-row sub-block
    const int l31 = lane & 31, hi = lane >> 5;

    // ---- which (group, kv head, split, row block) ------------------------------------------
    const int lin = xcd_remap(blockIdx.x, gridDim.x);
    const int rb = lin % a.row_blocks;
    int t_ = lin / a.row_blocks;
    const int sp = t_ % a.nsplit;
    t_ /= a.nsplit;
    const int hk = t_ % a.Hkv;
    const int gi = t_ / a.Hkv;

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
    const int nkt = (nkeys + 127) >> 7;
    // 32-key blocks per wave: its half of every tile (KG = 2) or all keys (KG = 1); even, the loop runs in pairs
    const int NB = KG == 2 ? 2 * nkt : 2 * ((nkeys + 63) >> 6);

    // ---- this lane's query row ------------------------------------------------------------
    const int r = rb * RWG + rw * 32 + l31;
    const bool rvalid = r < Mrows;
    const int rtok = rvalid ? r / a.g : 0;  // query token inside the group
    const int hq = hk * a.g + (rvalid ? r % a.g : 0);
    const int64_t row_off = ((int64_t)(q_tok0 + rtok) * a.Hq + hq) * D;
    int row_lim = 0x3fffffff;  // last visible key (causal)
    if (CAUSAL) row_lim = (rtok % nq_eff) + L - nq_eff;

    u32x4 qf[NC];
    {
        const uint16_t* qrow = static_cast<const uint16_t*>(a.q) + row_off;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            u32x4 z = {0u, 0u, 0u, 0u};
            qf[c] = rvalid ? *reinterpret_cast<const u32x4*>(qrow + 16 * c + 8 * hi) : z;
        }
    }

    // ---- LDS map (bytes): KA[2 tiles] | KB[2 tiles] | V[2 tiles] | mlbuf ------------------------------
    // KA: rows 0..63 of a K tile (kg = 0 waves), KB: rows 64..127 (kg = 1 waves), V: 128 rows.
    // Ring slot s (block b, s = b & 3): tile buffer s >> 1, rows [(s & 1) * 32, +32) of each 64-row half.
    // KG = 1: K[4 slots of 32 rows] | V[4 slots of 32 rows] | mlbuf.
    constexpr int KH_BYTES = 64 * RB;
    constexpr int V_BYTES = KG == 2 ? 128 * RB : 64 * RB;  // two ring slots of V
    constexpr int KA_OFF = 0, KB_OFF = 2 * KH_BYTES, V_OFF = KG == 2 ? 4 * KH_BYTES : 2 * KH_BYTES;
    typedef const __attribute__((address_space(3))) char* lptr_c;
    auto slot_k = [](int s) { return KG == 2 ? (s >> 1) * KH_BYTES + (s & 1) * 32 * RB : s * 32 * RB; };
    auto slot_v = [](int s) { return KG == 2 ? (s >> 1) * V_BYTES + (s & 1) * 32 * RB : s * 32 * RB; };

    // ---- per-lane LDS byte addresses of the MFMA fragments (ring slot 0) -------------------------------
    const int ksw = D == 128 ? (l31 & 15) : ((l31 >> 1) & 7);
    const int kx = hi ^ ksw;
    unsigned kaddr[NC];  // K fragment c of row l31 in slot 0 of this wave's key half (KA or KB)
#pragma unroll
    for (int c = 0; c < NC; ++c)
        kaddr[c] = (unsigned)(uintptr_t)(lptr_c)(smem + (kg ? KB_OFF : KA_OFF) + l31 * RB + (((2 * c) ^ kx) << 4));
    const int i16 = lane & 15, g16 = lane >> 4;
    const int vsw = D == 128 ? (i16 >> 2) : ((i16 >> 3) & 1);
    unsigned vaddr[NDB];  // V^T fragment address in slot 0 for key slot 0 of this wave's half
#pragma unroll
    for (int db = 0; db < NDB; ++db)
        vaddr[db] = (unsigned)(uintptr_t)(lptr_c)(smem + V_OFF + (kg * 64 + 4 * hi + (i16 >> 2)) * RB +
                                                  ((db ^ vsw) << 6) + 32 * (g16 & 1) + 8 * (i16 & 3));

    // ---- staging: global -> LDS DMA, one 32-key block (both key halves: 64 rows) per tensor at a time ----
    // The LDS image of a wave instruction is lane-linear (base + lane * 16), so the XOR swizzles are applied
    // to the per-lane SOURCE chunk (they are involutions inside a row).  Instruction q = wave * NLB + i
    // covers rows rr = q * RPI + [0, RPI) of the block pair: half h = rr >> 5, row r32 = rr & 31.
    const int drow = (lane * 16) / RB;        // row inside the instruction
    const int dcp = ((lane * 16) % RB) >> 4;  // 16-byte slot inside the row (LDS side)
    unsigned koffb[NLB], voffb[NLB];  // source byte offsets relative to the block's first row (fit 32 bits)
    unsigned kdst[NLB], vdst[NLB];    // wave-uniform LDS byte address of the instruction in ring slot 0
    int drr[NLB];                     // row (h * 64 + r32) inside the tile for slot-0 blocks
    const unsigned lds0 = (unsigned)(uintptr_t)(lptr_c)smem;
#pragma unroll
    for (int i = 0; i < NLB; ++i) {
        const int q = wave * NLB + i;
        const int rr = q * RPI + drow, h = KG == 2 ? rr >> 5 : 0, r32 = rr & 31;
        const int kch = D == 128 ? (dcp ^ (r32 & 15)) : (dcp ^ ((r32 >> 1) & 7));
        const int vs_ = D == 128 ? (r32 & 3) : ((r32 >> 1) & 1);
        const int vch = (((dcp >> 2) ^ vs_) << 2) | (dcp & 3);
        drr[i] = h * 64 + r32;
        koffb[i] = (unsigned)(((int64_t)drr[i] * a.k_ts + kch * 8) * 2);
        voffb[i] = (unsigned)(((int64_t)drr[i] * a.v_ts + vch * 8) * 2);
        const int qh = KG == 2 ? (q * RPI) >> 5 : 0, qr = (q * RPI) & 31;  // wave-uniform
        kdst[i] = __builtin_amdgcn_readfirstlane(lds0 + (qh ? KB_OFF : KA_OFF) + qr * RB);
        vdst[i] = __builtin_amdgcn_readfirstlane(lds0 + V_OFF + (qh * 64 + qr) * RB);
    }
    const char* kbase = reinterpret_cast<const char*>(k16) + (int64_t)kbeg * a.k_ts * 2;
    const char* vbase = reinterpret_cast<const char*>(v16) + (int64_t)kbeg * a.v_ts * 2;
    // DMA of 32-key block b (any integer: blocks / rows outside [0, nkeys) become zeros in LDS) into ring
    // slot b & 3; piece = one of the 2 * NLB instructions of the (K block, V block) pair of an iteration.
    auto rsrc_of = [&](const char* base, int64_t ts, int b) -> u32x4 {
        const int row0 = KG == 2 ? (b >> 1) * 128 + (b & 1) * 32 : b * 32;
        const int rem = min(nkeys - row0, KG == 2 ? 128 : 32);  // rows the window of this block (pair) may touch
        const unsigned bytes = (b >= 0 && rem > 0) ? (unsigned)((int64_t)(rem - 1) * ts * 2 + RB) : 0u;
        return make_rsrc(base + (int64_t)(b >= 0 ? row0 : 0) * ts * 2, bytes);
    };
    auto dma_block = [&](int b, bool isv) __attribute__((always_inline)) {
        const u32x4 rs = isv ? rsrc_of(vbase, a.v_ts, b) : rsrc_of(kbase, a.k_ts, b);
#pragma unroll
        for (int i = 0; i < NLB; ++i) {
            if (isv) dma16b(rs, voffb[i], vdst[i] + slot_v(b & 3));
            else dma16b(rs, koffb[i], kdst[i] + slot_k(b & 3));
        }
    };

    f32x16 o[NDB];
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[db][i] = 0.f;
    float m_run = -INFINITY, l_run = 0.f, m_thr = -INFINITY;
    const float sc = a.scale_log2e;
    constexpr float kTau = 8.0f;
    f32x16 S0, S1;
    u32x4 P0[2], P1[2];
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 16; ++i) S0[i] = S1[i] = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) P0[i] = P1[i] = u32x4{0u, 0u, 0u, 0u};

    // ---- one pipeline iteration i: QK(i+1) | SM(i) | PV(i-1) -------------------------------------------
    // KOFF / VOFF: compile-time ring-slot byte offsets of K block i+1 and V block i-1;
    // NKOFF / NVOFF: the same for iteration i+1, whose first fragments are prefetched in the tail of this one.
    // FL: bit0 QK, bit1 SM, bit2 PV, bit3 the softmax may need masking, bit4 / bit5: iteration i+1 has QK / PV.
    // DM: 0 no DMA; 1: K block i+4 and V block i+2 through the buffer resources krs / vrs into ring slots
    //     kslot / vslot (byte offsets), spread between the MFMAs.
    // bvalid: block i exists (iterations -1 and NB run the same code on a fully masked block).
    // Sw: scores written by QK; Sr: scores consumed by the softmax, which writes Pw; PV reads Pr.
    // kw: first key of block i for this wave (masking).
    constexpr int PDK = 3, PDV = 2;  // LDS prefetch distance (in MFMAs of the own stream)
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
        return lds_tr16_pl(vaddr[p % NDB] + off + (16 * (p / NDB) + 8 * h) * RB);
    };
    auto iter = [&](auto KOFF_C, auto VOFF_C, auto NKOFF_C, auto NVOFF_C, auto FL_C, auto DM_C, f32x16& Sw, f32x16& Sr,
                    u32x4(&Pw)[2], u32x4(&Pr)[2], int kw, bool bvalid,
                    u32x4 krs, u32x4 vrs, int kslot, int vslot) __attribute__((always_inline)) {
        constexpr int KOFF = decltype(KOFF_C)::value;
        constexpr int VOFF = decltype(VOFF_C)::value;
        constexpr int NKOFF = decltype(NKOFF_C)::value;
        constexpr int NVOFF = decltype(NVOFF_C)::value;
        constexpr int FL = decltype(FL_C)::value;
        constexpr int DM = decltype(DM_C)::value;
        constexpr bool QK = FL & 1, SM = FL & 2, PV = FL & 4, MASK = (FL & 8) || CAUSAL;
        constexpr bool NQK = FL & 16, NPVF = FL & 32;
        constexpr int NPV = 2 * NDB;        // PV MFMAs (2 key slots x NDB d blocks)
        constexpr int NSLOT = NC + NPV;     // MFMA slots; QK and PV alternate
        constexpr int NG = 16;              // VALU groups of the softmax
        auto ldk = [&](int c) -> u32x4 { return ldk_at(c, KOFF); };
        auto ldv = [&](int p, int h) -> u32x2 { return ldv_at(p, h, VOFF); };
        // softmax state of this iteration
        float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f, tmax = 0.f, nms = 0.f, alpha = 1.f;
        f32x2 sum2 = {0.f, 0.f};
        bool any_up = false;
        if constexpr (SM && MASK) {
            int lim = kend - 1;
            if (CAUSAL) lim = min(lim, row_lim);
            const bool need_mask = !bvalid || (kw + 32 > kend) ||
                                   (CAUSAL && __builtin_amdgcn_ballot_w64(row_lim < kw + 31) != 0ull);
            if (need_mask) {
                asm volatile("" ::: "memory");  // keep this a (cold) branch
                const int lr = bvalid ? lim - kw - 4 * hi : -1;
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    if (8 * (i >> 2) + (i & 3) > lr) Sr[i] = -INFINITY;
            }
        }
        auto valu_group = [&](int g) __attribute__((always_inline)) {
            if ((ABL & 2) && g >= 4) return;
            if constexpr (SM && !(ABL & 4)) {
                if (g == 0) {
                    t0 = fmaxf(fmaxf(Sr[0], Sr[1]), Sr[2]);
                    t1 = fmaxf(fmaxf(Sr[3], Sr[4]), Sr[5]);
                    t2 = fmaxf(fmaxf(Sr[6], Sr[7]), Sr[8]);
                    t3 = fmaxf(fmaxf(Sr[9], Sr[10]), Sr[11]);
                    asm volatile("" ::"v"(t0), "v"(t1), "v"(t2), "v"(t3));
                } else if (g == 1) {
                    const float t4 = fmaxf(fmaxf(Sr[12], Sr[13]), Sr[14]);
                    t0 = fmaxf(fmaxf(t0, t1), Sr[15]);
                    t2 = fmaxf(fmaxf(t2, t3), t4);
                    tmax = fmaxf(t0, t2);
                    asm volatile("" ::"v"(tmax));
                } else if (g == 2) {
                    tmax = pair_max(tmax) * sc;
                    asm volatile("" ::"v"(tmax));
                } else if (g == 3) {
                    const bool up = tmax > m_thr;  // only raise the reference maximum by > kTau at a time
                    any_up = __builtin_amdgcn_ballot_w64(up) != 0ull;
                    const float m_new = up ? tmax : m_run;
                    const float msafe = (m_new == -INFINITY) ? 0.f : m_new;
                    alpha = fast_exp2(m_run - msafe);
                    m_run = m_new;
                    m_thr = m_new + kTau;
                    nms = -msafe;
                    asm volatile("" ::"v"(alpha), "v"(nms), "v"(m_thr));
                } else {
                    // groups 4..15: 4 batches of 4 elements, 3 groups per batch
                    const int bt = (g - 4) / 3, ph = (g - 4) % 3, e0 = 4 * bt;
                    if (ph < 2) {
                        const int e = e0 + 2 * ph;
                        f32x2 x = {Sr[e], Sr[e + 1]};
                        const f32x2 sc2 = {sc, sc}, nm2 = {nms, nms};
                        x = x * sc2 + nm2;
                        Sr[e] = fast_exp2(x[0]);
                        Sr[e + 1] = fast_exp2(x[1]);
                        asm volatile("" ::"v"(Sr[e]), "v"(Sr[e + 1]));
                    } else {
                        const f32x2 pa = {Sr[e0], Sr[e0 + 1]}, pb = {Sr[e0 + 2], Sr[e0 + 3]};
                        sum2 += pa;
                        sum2 += pb;
                        // element pair k = e/2 -> P^T slot (k >> 2), word (k & 3)
                        Pw[(e0 / 2) >> 2][(e0 / 2) & 3] = TR::pack2(pa[0], pa[1]);
                        Pw[(e0 / 2 + 1) >> 2][(e0 / 2 + 1) & 3] = TR::pack2(pb[0], pb[1]);
                        asm volatile("" ::"v"(Pw[(e0 / 2) >> 2][(e0 / 2) & 3]), "v"(Pw[(e0 / 2 + 1) >> 2][(e0 / 2 + 1) & 3]),
                                     "v"(sum2));
                    }
                }
            }
        };
#pragma unroll
        for (int j = 0; j < NSLOT; ++j) {
            // MFMA of this slot: even -> QK chunk j/2, odd -> PV MFMA j/2 (D = 128: 8 + 8, D = 64: 4 + 4)
            const int idx = j >> 1;
            if ((j & 1) == 0) {
                if constexpr (QK) {
                    Sw = TR::mfma32(kfr[idx % PDK], qf[idx], idx == 0 ? zero16 : Sw);
                    if (idx + PDK < NC && !(ABL & 8)) kfr[idx % PDK] = ldk(idx + PDK);
                }
            } else {
                if constexpr (PV) {
                    const u32x4 vf = {vfr[idx % PDV][0][0], vfr[idx % PDV][0][1], vfr[idx % PDV][1][0], vfr[idx % PDV][1][1]};
                    o[idx % NDB] = TR::mfma32(vf, Pr[idx / NDB], o[idx % NDB]);
                    if (idx + PDV < NPV && !(ABL & 8)) { vfr[idx % PDV][0] = ldv(idx + PDV, 0); vfr[idx % PDV][1] = ldv(idx + PDV, 1); }
                }
            }
#pragma unroll
            for (int g = (j * NG) / NSLOT; g < ((j + 1) * NG) / NSLOT; ++g) valu_group(g);
            if constexpr (DM == 1 && !(ABL & 1)) {
                // 2 * NLB DMA instructions spread over the iteration, issued early (slots 1, 3 [, 5, 7])
                constexpr int EVERY = 2;
                if ((j % EVERY) == EVERY - 1 && j / EVERY < 2 * NLB) {
                    const int i = j / EVERY;
                    if (i < NLB) dma16b(krs, koffb[i], kdst[i] + kslot);
                    else dma16b(vrs, voffb[i - NLB], vdst[i - NLB] + vslot);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // first fragments of iteration i+1 (its blocks are already visible, see the header)
        if constexpr (NQK) {
#pragma unroll
            for (int c = 0; c < PDK; ++c) kfr[c] = ldk_at(c, NKOFF);
        }
        if constexpr (NPVF) {
#pragma unroll
            for (int p = 0; p < PDV; ++p) { vfr[p][0] = ldv_at(p, 0, NVOFF); vfr[p][1] = ldv_at(p, 1, NVOFF); }
        }
        if constexpr (SM && !(ABL & 4)) {
            l_run = l_run * alpha + (sum2[0] + sum2[1]);
            if (any_up) {  // cold: at most a handful of times per row block
#pragma unroll
                for (int db = 0; db < NDB; ++db) o[db] *= alpha;
            }
        }
    };

    // ---- pipeline over 32-key blocks ---------------------------------------------------------------------
    using std::integral_constant;
#define HYD_IC(x) integral_constant<int, (x)>{}
    const int kwave = kbeg + kg * 64;  // first key of this wave's half of tile 0
    auto kw_of = [&](int b) { return KG == 2 ? kwave + (b >> 1) * 128 + (b & 1) * 32 : kwave + b * 32; };
    // Every iteration i = -1 .. NB runs the same code: stages whose block does not exist work on zeros /
    // fully masked scores (QK(NB) and SM(-1), SM(NB) are harmless, PV(-2), PV(-1) add P = 0 times finite V).
    // Cold start: every workgroup of the chip fetches at once, so only what the first iteration needs is waited
    // for (Q, K blocks 0 and 1); K block 2 and V block 0 are issued behind that wait and land during iterations
    // -1 / 0; V slots 2 and 3 are zero-filled in LDS instead of being fetched.
    if (NB > 0) {
        dma_block(0, false);
        dma_block(1, false);
        // slots 2 and 3 of the V ring (= tile buffer 1) are read by PV(-2) / PV(-1) with P = 0: must hold finite data
        for (int off = tid * 16; off < V_BYTES; off += 512 * 16)  // V_BYTES = two slots in either layout
            *reinterpret_cast<u32x4*>(smem + V_OFF + V_BYTES + off) = u32x4{0u, 0u, 0u, 0u};
    }
    // Make the compiler wait for the Q fragments here, not (conservatively, with vmcnt(0)) inside the loop.
#pragma unroll
    for (int c = 0; c < NC; ++c) asm volatile("" ::"v"(qf[c]));
    stampk(1);
    dma_wait<0>();
    if (NB > 0) {
        dma_block(2, false);  // covered by the counted wait that ends iteration -1
        dma_block(0, true);
    }
    __syncthreads();
    stampk(2);
    if (NB > 0) {
#pragma unroll
        for (int c = 0; c < PDK; ++c) kfr[c] = ldk_at(c, slot_k(0));
#pragma unroll
        for (int p = 0; p < PDV; ++p) { vfr[p][0] = ldv_at(p, 0, slot_v(2)); vfr[p][1] = ldv_at(p, 1, slot_v(2)); }
        // ii = i0 + R = -1 + R (mod 4), so every ring slot is a compile-time constant:
        //   reads  K block ii+1 -> slot R, V block ii-1 -> slot (R+2)&3;  next iteration: (R+1)&3, (R+3)&3
        //   writes K block ii+4 -> slot (R+3)&3, V block ii+2 -> slot (R+1)&3.      ii odd <=> R even.
#define HYD_IT(R, SW, SR, PW, PR)                                                                            \
    {                                                                                                        \
        const int ii = i0 + (R);                                                                             \
        iter(HYD_IC(slot_k((R) & 3)), HYD_IC(slot_v(((R) + 2) & 3)), HYD_IC(slot_k(((R) + 1) & 3)),          \
             HYD_IC(slot_v(((R) + 3) & 3)), HYD_IC(7 + 8 + 16 + 32), HYD_IC(1), SW, SR, PW, PR, kw_of(ii),   \
             ii >= 0 && ii < NB, rsrc_of(kbase, a.k_ts, ii + 4), rsrc_of(vbase, a.v_ts, ii + 2), \
             slot_k(((R) + 3) & 3), slot_v(((R) + 1) & 3));                                                  \
        dma_wait<2 * NLB>();                                                                                 \
        if (!(ABL & 16)) __syncthreads();                                                                    \
    }
        for (int i0 = -1;; i0 += 4) {
            HYD_IT(0, S0, S1, P1, P0)
            HYD_IT(1, S1, S0, P0, P1)
            if (i0 + 2 >= NB) break;
            HYD_IT(2, S0, S1, P1, P0)
            HYD_IT(3, S1, S0, P0, P1)
            if (i0 + 4 >= NB) break;
        }
#undef HYD_IT
        dma_wait<0>();
        __syncthreads();  // nothing in flight, everyone done with the rings before the merge reuses them
    }
#undef HYD_IC

    // ---- merge the two key halves through LDS, normalise, store --------------------------------
    // Both waves of a pair (same rows, key half 0 / 1) take part: the wave of key half h finalises the d blocks
    // [h * NDB/2, (h+1) * NDB/2) and hands the other blocks (+ its m, l) to its partner through LDS.  A lane holds
    // 4 consecutive d per (d block, q4); v_permlane32_swap pairs the two half-waves' groups so that each lane
    // stores 16 contiguous bytes (row-per-lane stores are issue-bound: 4 instead of 16 per wave, on all 8 waves).
    stampk(3);
    constexpr int HDB = KG == 2 ? NDB / 2 : NDB;  // d blocks this wave finalises
    const float l_tot = pair_sum(l_run);
    f32x4* obuf = reinterpret_cast<f32x4*>(smem);  // [8 waves][HDB * 4][64 lanes] of f32x4
    if constexpr (KG == 2) {
#pragma unroll
        for (int i = 0; i < HDB; ++i)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const f32x16& ob = kg ? o[i] : o[HDB + i];
                f32x4 x = {ob[4 * q4], ob[4 * q4 + 1], ob[4 * q4 + 2], ob[4 * q4 + 3]};
                obuf[(wave * HDB * 4 + i * 4 + q4) * 64 + lane] = x;
            }
        mlbuf[wave * 128 + lane] = m_run;
        mlbuf[wave * 128 + 64 + lane] = l_tot;
    }
    if constexpr (KG == 2) __syncthreads();
    stampk(4);
    const int pw = wave ^ 4;  // partner wave
    const float m1 = KG == 2 ? mlbuf[pw * 128 + lane] : -INFINITY;
    const float l1 = KG == 2 ? mlbuf[pw * 128 + 64 + lane] : 0.f;
    const float mf = fmaxf(m_run, m1);
    const float mfs = (mf == -INFINITY) ? 0.f : mf;
    const float a0 = fast_exp2(m_run - mfs), a1 = fast_exp2(m1 - mfs);
    const float lf = l_tot * a0 + l1 * a1;
    const float inv = lf > 0.f ? 1.0f / lf : 0.f;
    const float w0 = a0 * inv, w1 = a1 * inv;

    const int64_t obase = (int64_t)sp * a.out_split_stride + row_off;
#pragma unroll
    for (int i = 0; i < HDB; ++i) {
        const f32x16& ob = (KG == 2 && kg) ? o[NDB - HDB + i] : o[i];
        const int db = KG == 2 ? kg * HDB + i : i;
        f32x4 x[4];
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            if constexpr (KG == 2) {
                const f32x4 y = obuf[(pw * HDB * 4 + i * 4 + q4) * 64 + lane];
#pragma unroll
                for (int j = 0; j < 4; ++j) x[q4][j] = ob[4 * q4 + j] * w0 + y[j] * w1;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) x[q4][j] = ob[4 * q4 + j] * w0;
            }
        }
        if (a.out_f32) {
            if (rvalid) {
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
                if (rvalid)
                    *reinterpret_cast<u32x4*>(static_cast<uint16_t*>(a.out) + obase + 32 * db + 8 * (2 * qp + hi)) = w;
            }
        }
    }
    if (a.lse && kg == 0 && hi == 0 && rvalid) {
        const float lse = lf > 0.f ? mf * kLn2 + __logf(lf) : -INFINITY;
        int64_t idx;
        if (a.lse_layout == HYD_LSE_BQH)
            idx = (int64_t)(q_tok0 + rtok) * a.Hq + hq;
        else
            idx = ((int64_t)gi * a.Hq + hq) * a.lse_q_stride + rtok;
        a.lse[(int64_t)sp * a.lse_split_stride + idx] = lse;
    }
    if constexpr ((ABL & 64) != 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stampk(5);
        if (blockIdx.x == 0 && lane < 6) {
            unsigned tv = tst[0];
            for (int q_ = 1; q_ < 6; ++q_) tv = lane == q_ ? tst[q_] : tv;
            reinterpret_cast<unsigned*>(a.lse)[(size_t)a.B * a.nq * a.Hq + wave * 8 + lane] = tv;
        }
    }
}

template <typename T, int D, bool CAUSAL, int ABL = 0, int KG = 2>
static int launch_prefix_pl_t(const PrefixArgs& a, int grid, hipStream_t s) {
    constexpr size_t lds = (KG == 2 ? 2 : 1) * 256 * (D * 2) + 8 * 128 * sizeof(float);
    auto kern = prefix_attn_pl_kernel<T, D, CAUSAL, ABL, KG>;
    // once per instantiation, thread-safe (C++11 static initialisation); the value never changes afterwards
    static const hipError_t attr_rc = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (attr_rc != hipSuccess) return (int)attr_rc;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, s, a);
    return (int)hipGetLastError();
}

int launch_prefix_pl(const PrefixArgs& a, int dtype, int D, bool causal, int grid, hipStream_t s) {
#ifdef HYD_ABLATION_BUILD
    if (a.dbg && dtype == HYD_BF16 && D == 128 && !causal) {
        switch (a.dbg) {
#define HYD_ABL(N) case N: return launch_prefix_pl_t<BF16, 128, false, N>(a, grid, s);
            HYD_ABL(1) HYD_ABL(2) HYD_ABL(8) HYD_ABL(16) HYD_ABL(64)
#undef HYD_ABL
            default: break;
        }
    }
#endif
    if (a.wg_rows == 256) {
        if (D != 128) return (int)hipErrorInvalidValue;
        if (dtype == HYD_F16)
            return causal ? launch_prefix_pl_t<F16, 128, true, 0, 1>(a, grid, s) : launch_prefix_pl_t<F16, 128, false, 0, 1>(a, grid, s);
        return causal ? launch_prefix_pl_t<BF16, 128, true, 0, 1>(a, grid, s) : launch_prefix_pl_t<BF16, 128, false, 0, 1>(a, grid, s);
    }
#define HYD_DISPATCH(TT, DD)                                                    \
    return causal ? launch_prefix_pl_t<TT, DD, true>(a, grid, s) : launch_prefix_pl_t<TT, DD, false>(a, grid, s)
    if (dtype == HYD_F16) {
        if (D == 128) { HYD_DISPATCH(F16, 128); }
        if (D == 64) { HYD_DISPATCH(F16, 64); }
    } else {
        if (D == 128) { HYD_DISPATCH(BF16, 128); }
        if (D == 64) { HYD_DISPATCH(BF16, 64); }
    }
#undef HYD_DISPATCH
    return (int)hipErrorInvalidValue;
}

}  // namespace hyd
