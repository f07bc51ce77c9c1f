// nnr_mlp_bf16.h -- building blocks of the bf16-MFMA MLP kernels (nnr_mlp_fwd_bf16.hip, nnr_mlp_dgrad_bf16.hip), gfx950 only.
//
// Why these kernels are not the fp32 ones with another MFMA: one v_mfma_f32_32x32x16_bf16 (32 cycles) consumes a whole 1 KiB weight
// fragment, so a wave of 32 samples needs 1 KiB of LDS per 32 matrix-pipe cycles -- four waves = 128 B/clk, the LDS read peak of a CU:
// with 32-sample waves the bf16 layers run at the LDS roofline, not the MFMA one (measured: 8 000 cycles per D x D layer and workgroup
// against 4 096 of MFMA).  Here a wave owns TWO chunks of 32 samples (kTiles): every fragment read feeds two MFMAs, the LDS stream
// halves, and the per-row overhead (fragment reads, DMA pieces, panel switches) is spread over twice the matrix work.
//
// That needs the activations PACKED between layers -- 2 x 32 x 256 values per wave do not fit as fp32 beside 256 accumulator
// registers.  A lane keeps, per tile, the layer input as 8 * DT packed registers: packed register p = (bf16 of fragment register 2p,
// bf16 of 2p + 1) (nnr_layout.h), so the four registers 4g .. 4g+3 ARE the MFMA B operand of row g -- no conversion at the MFMA, and
// the same 16 bytes are what the training stash stores (one tile-major block row per store).  The epilogue of a half-output pass
// (bias is in the accumulator; ReLU + gate bits, or the ReLU' select) produces one packed register per unit (v_cvt_pk_bf16_f32).
// Everything else is the design of the fp32 kernels: weights through the DMA-fed LDS panel ring, two half-output passes per layer
// with the epilogue of one pass hidden in the MFMA stream of the next, heads on the VALU.
#pragma once
#include "nnr_device.h"

namespace nnr {

// Two shapes of the same kernels (template parameters T = 32-sample chunks per wave, W = waves per workgroup):
//   T = 2, W = 4  one wave per SIMD, 512 registers each, every weight fragment read feeds two MFMAs ("wide");
//   T = 1, W = 8  two waves per SIMD, 256 registers each: one wave's epilogue VALU work runs while the other's MFMAs occupy the
//                 matrix pipe -- within ONE wave the two do not overlap (measured: every VALU instruction between two MFMAs costs
//                 its issue time).
// Either way a workgroup covers 256 samples per pass.
constexpr int kMaxTiles = 2;
constexpr int kDmaBurst = 4;   // DMA pieces (1 KiB each) a wave issues per row until the 8 of a panel are out
constexpr int kWideSamples = 256;   // samples per workgroup and pass (T * 32 * W)

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

// (bf16(lo), bf16(hi)) in one register, round to nearest even.  Spelled as the instruction: written as two __bf16 conversions and a
// bit cast, the <2 x bfloat> stores into the packed arrays keep SROA from promoting parts of them (mixed-type slices) -- some rows
// of the MFMA operands then live in scratch memory.
__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi) {
    uint32_t r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}

// acc += lo(a) * lo(b) + hi(a) * hi(b), the pairs bf16, the sum fp32 (the per-lane dot products of the two heads)
__device__ __forceinline__ void dot2_bf16(float& acc, uint32_t a, uint32_t b) {
    asm("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(acc) : "v"(a), "v"(b));
}

// A value the compiler must treat as new at this program point.  The kernels derive every plane address from a 32-bit sample index
// THROUGH this, right where the address is needed: otherwise the dozens of lane-constant 64-bit addresses of a pass are computed at
// its start and spilled around the MFMA streams (a spill reload is a VMEM load whose wait drains the weight DMA queue).
__device__ __forceinline__ int opaque(int x) {
    asm volatile("" : "+v"(x));
    return x;
}
// The lane index, re-derived on the spot (two VALU instructions, volatile so that the copies are not merged): the kernels use it for
// every address and bounds test OUTSIDE the MFMA streams instead of a lane / half / column value kept in a register for the whole pass
// -- at 512 registers per lane even that one register is spilled, and its reload drains the store queue like any other load.
__device__ __forceinline__ int lane_id() {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}
// x of this lane + x of the lane 32 further (the other half-wave, which holds the other half of the features of the same sample):
// v_permlane32_swap on two copies leaves (lower, lower) in one and (upper, upper) in the other -- their sum is the same number in both
// halves, bitwise what x + __shfl_xor(x, 32) gives, without the ds_bpermute index register the shuffle keeps alive (and spills).
__device__ __forceinline__ float sum_halves(float x) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// The same for a WAVE-UNIFORM value, which stays in a scalar register: the chunk index of a tile is one (block and wave index, pass
// counter), so every plane address is (scalar base + scalar chunk offset) + a 32-bit lane offset and costs no 64-bit vector register --
// made opaque as a VGPR it turned each use into a 64-bit per-lane address the compiler then kept, and spilled, per pass.
__device__ __forceinline__ int opaque_uniform(int x) {
    asm volatile("" : "+s"(x));
    return x;
}

// The value of a dot2_bf16 chain, safe to use.  gfx950 needs 3 wait states between a DOT instruction's write and a read by any other
// kind of instruction (and 2 before another instruction overwrites the register); hipcc inserts them for instructions it knows, but an
// inline-asm DOT is opaque to its hazard recogniser: without this the first consumer saw the accumulator before the last DOT's
// write (rgb off by 1e-2).  The s_nop sits between the chain and the only instruction allowed to touch the register.
__device__ __forceinline__ float dot2_result(float acc) {
    float r;
    asm volatile("s_nop 3\n\tv_mov_b32 %0, %1" : "=v"(r) : "v"(acc));
    return r;
}

template <int NIN>
__device__ __forceinline__ bf16x8 row_operand(const uint32_t (&in)[NIN], int g) {
    const u32x4 q = {in[4 * g], in[4 * g + 1], in[4 * g + 2], in[4 * g + 3]};
    return __builtin_bit_cast(bf16x8, q);
}

// Stash stores are NON-TEMPORAL: the 2.4 GB a launch writes are read next by another kernel, not by this one, and as ordinary
// stores they stream through the L2 and push the packed weights out of it -- every workgroup re-reads those ~1.2 MB per pass, and a
// weight panel that misses L2 arrives late at its panel switch.  Measured at 4096 x 128: forward 0.94 -> 0.78 ms, input-gradient
// 0.88 -> 0.75 ms (with the row-major planes of round 1, 16 bytes per line and instruction, the same hint made things worse).
template <class V>
__device__ __forceinline__ void stash_store(void* dst, V v) {
#if NNR_ABLATE & 8
    (void)dst; (void)v;      // profiling builds only
#else
    __builtin_nontemporal_store(v, reinterpret_cast<V*>(dst));
#endif
}

// relu of a packed bf16 pair in ONE instruction: as 16-bit integers the non-negative bf16 values are the non-negative integers, and
// everything with the sign bit set (negative numbers, -0) is a negative integer -- v_pk_max_i16 with 0 is exactly relu
__device__ __forceinline__ uint32_t relu_bf16x2(uint32_t p) {
    uint32_t r;
    asm("v_pk_max_i16 %0, %1, 0" : "=v"(r) : "v"(p));
    return r;
}

constexpr int kPh = 1;   // epilogue units issued whole (1) or as two half-units in different MFMA gaps (2)

// ---- ReLU gates of the training mode, one bit per value, in the layout the input-gradient kernel's select wants ---------------------
// A mask word covers 16 consecutive packed registers (32 values) of a lane: the gate of the LOW value of pair j sits at bit 15 - j, of
// the HIGH value at bit 31 - j.  The forward appends a pair with two instructions -- (v_pk_min_u16 relu'd pair, (1, 1)) turns each
// non-zero half into 1, v_lshl_or_b32 shifts it in -- so the gate is (stored activation != 0): exactly torch's relu backward, also for
// an accumulator that cancelled to +0.0 (the sign-bit gate of the first version let those pass).  The input-gradient kernel shifts
// the pair's two bits to the sign positions of the halves and smears them (v_pk_ashrrev_i16 by 15): the AND mask of the packed pair.
__device__ __forceinline__ uint32_t gate_append(uint32_t word, uint32_t relu_pair) {
    uint32_t t;
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(t) : "v"(relu_pair), "s"(0x00010001u));
    return (word << 1) | t;
}
__device__ __forceinline__ uint32_t gate_mask(uint32_t word, int j /* pair index inside the word, compile-time */) {
    uint32_t m;
    asm("v_pk_ashrrev_i16 %0, %1, %2" : "=v"(m) : "s"(0x000f000fu), "v"(word << j));
    return m;
}

// The epilogue of a value pair as ONE asm statement each.  Written as separate statements (pack_bf16, relu_bf16x2, gate_append /
// gate_mask) hipcc puts an s_nop between two of them whenever the second reads what the first wrote -- it cannot know the first is not
// a DOT or a transcendental -- and with one wave per SIMD an s_nop costs the wave the same ~4 cycles of issue as a real instruction
// (tools/ubench/mfma_valu_overlap.hip): 1.7 of the forward's 10.9 non-MFMA instructions per MFMA were s_nops.
// forward, training: packed relu'd pair; `word` gets the pair's two gates appended (gate_append's layout)
__device__ __forceinline__ uint32_t relu_pack_gate(float x0, float x1, uint32_t& word) {
    uint32_t h, t;
    asm("v_cvt_pk_bf16_f32 %0, %3, %4\n\tv_pk_max_i16 %0, %0, 0\n\tv_pk_min_u16 %1, %0, %5\n\tv_lshl_or_b32 %2, %2, 1, %1"
        : "=&v"(h), "=&v"(t), "+v"(word)
        : "v"(x0), "v"(x1), "s"(0x00010001u));
    return h;
}
// forward, inference: packed relu'd pair
__device__ __forceinline__ uint32_t relu_pack(float x0, float x1) {
    uint32_t h;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2\n\tv_pk_max_i16 %0, %0, 0" : "=&v"(h) : "v"(x0), "v"(x1));
    return h;
}
// input gradient: packed pair gated by pair j of `word` (gate_mask's shift / smear / and); J is a compile-time constant
template <int J>
__device__ __forceinline__ uint32_t pack_gated(float x0, float x1, uint32_t word) {
    uint32_t d, m;
    if constexpr (J == 0)
        asm("v_pk_ashrrev_i16 %1, %5, %4\n\tv_cvt_pk_bf16_f32 %0, %2, %3\n\tv_and_b32 %0, %0, %1"
            : "=&v"(d), "=&v"(m)
            : "v"(x0), "v"(x1), "v"(word), "s"(0x000f000fu));
    else
        asm("v_lshlrev_b32 %1, %6, %4\n\tv_cvt_pk_bf16_f32 %0, %2, %3\n\tv_pk_ashrrev_i16 %1, %5, %1\n\tv_and_b32 %0, %0, %1"
            : "=&v"(d), "=&v"(m)
            : "v"(x0), "v"(x1), "v"(word), "s"(0x000f000fu), "n"(J));
    return d;
}

// ---- PAIRED epilogue units ---------------------------------------------------------------------------------------------------------
// HYPOTHESIS (round 3): the epilogue of a value pair is a DEPENDENT chain (convert -> relu -> gate bits -> append), and with one wave per
// SIMD nothing else issues between two links -- the side work of a pass (6 instructions per pair) costs 0.19 ms of the 0.76 ms
// training forward, ~8 cycles per instruction where an independent instruction issues in 4.  The two tiles of a wave are independent, so the units of
// the same packed register of tile 0 and tile 1 are issued TOGETHER, link by link -- every instruction then depends on the one two
// back, not on its predecessor -- and in two phases that go to different MFMA gaps (six instructions each): phase 0 = the four
// accumulator reads + the two converts, phase 1 = everything else.
__device__ __forceinline__ void pack2(uint32_t& h0, uint32_t& h1, float x00, float x01, float x10, float x11) {
    asm("v_cvt_pk_bf16_f32 %0, %2, %3\n\tv_cvt_pk_bf16_f32 %1, %4, %5" : "=&v"(h0), "=&v"(h1) : "v"(x00), "v"(x01), "v"(x10), "v"(x11));
}
__device__ __forceinline__ void relu_gate2(uint32_t& h0, uint32_t& h1, uint32_t& w0, uint32_t& w1) {      // forward, training
    uint32_t t0, t1;
    asm("v_pk_max_i16 %0, %0, 0\n\tv_pk_max_i16 %1, %1, 0\n\tv_pk_min_u16 %4, %0, %6\n\tv_pk_min_u16 %5, %1, %6\n\t"
        "v_lshl_or_b32 %2, %2, 1, %4\n\tv_lshl_or_b32 %3, %3, 1, %5"
        : "+v"(h0), "+v"(h1), "+v"(w0), "+v"(w1), "=&v"(t0), "=&v"(t1)
        : "s"(0x00010001u));
}
__device__ __forceinline__ void relu2(uint32_t& h0, uint32_t& h1) {                                         // forward, inference
    asm("v_pk_max_i16 %0, %0, 0\n\tv_pk_max_i16 %1, %1, 0" : "+v"(h0), "+v"(h1));
}
template <int J>
__device__ __forceinline__ void gate2(uint32_t& d0, uint32_t& d1, uint32_t w0, uint32_t w1) {             // input gradient: AND with pair J's gates
    uint32_t m0, m1;
    if constexpr (J == 0)
        asm("v_pk_ashrrev_i16 %2, %6, %4\n\tv_pk_ashrrev_i16 %3, %6, %5\n\tv_and_b32 %0, %0, %2\n\tv_and_b32 %1, %1, %3"
            : "+v"(d0), "+v"(d1), "=&v"(m0), "=&v"(m1)
            : "v"(w0), "v"(w1), "s"(0x000f000fu));
    else
        asm("v_lshlrev_b32 %2, %7, %4\n\tv_lshlrev_b32 %3, %7, %5\n\tv_pk_ashrrev_i16 %2, %6, %2\n\tv_pk_ashrrev_i16 %3, %6, %3\n\t"
            "v_and_b32 %0, %0, %2\n\tv_and_b32 %1, %1, %3"
            : "+v"(d0), "+v"(d1), "=&v"(m0), "=&v"(m1)
            : "v"(w0), "v"(w1), "s"(0x000f000fu), "n"(J));
}
__device__ __forceinline__ void gate2_at(uint32_t& d0, uint32_t& d1, uint32_t w0, uint32_t w1, int j) {   // j folds after unrolling
    switch (j) {
#define NNR_G2_CASE(J) case J: gate2<J>(d0, d1, w0, w1); break;
        NNR_G2_CASE(0) NNR_G2_CASE(1) NNR_G2_CASE(2) NNR_G2_CASE(3) NNR_G2_CASE(4) NNR_G2_CASE(5) NNR_G2_CASE(6) NNR_G2_CASE(7)
        NNR_G2_CASE(8) NNR_G2_CASE(9) NNR_G2_CASE(10) NNR_G2_CASE(11) NNR_G2_CASE(12) NNR_G2_CASE(13) NNR_G2_CASE(14)
#undef NNR_G2_CASE
        default: gate2<15>(d0, d1, w0, w1); break;
    }
}
// MEASURED (profiles/r03/i_pairs_ab.txt, 4096 x 128): forward 0.771 / 0.775 ms paired vs 0.745 / 0.776 unpaired, input gradient 0.601 /
// 0.611 vs 0.602 / 0.610 -- no difference beyond the box's run-to-run spread.  The dependent chain is NOT what makes the side work
// expensive; the product keeps the one-statement-per-pair units and this form stays behind kPairs as a recorded negative.
constexpr bool kPairs = false;

__device__ __forceinline__ uint32_t sel_pair(float x0, float x1, uint32_t word, int j) {   // j folds after unrolling
    switch (j) {
#define NNR_SEL_CASE(J) case J: return pack_gated<J>(x0, x1, word);
        NNR_SEL_CASE(0) NNR_SEL_CASE(1) NNR_SEL_CASE(2) NNR_SEL_CASE(3) NNR_SEL_CASE(4) NNR_SEL_CASE(5) NNR_SEL_CASE(6) NNR_SEL_CASE(7)
        NNR_SEL_CASE(8) NNR_SEL_CASE(9) NNR_SEL_CASE(10) NNR_SEL_CASE(11) NNR_SEL_CASE(12) NNR_SEL_CASE(13) NNR_SEL_CASE(14)
#undef NNR_SEL_CASE
        default: return pack_gated<15>(x0, x1, word);
    }
}
constexpr bool kSplitAsm = false;

// ---- weight fragments: LDS reads the compiler does not see ------------------------------------------------------------------
// hipcc waits lgkmcnt(0) before the first MFMA that uses a fragment it loaded itself -- i.e. for EVERY outstanding LDS read, also the
// refill issued one instruction earlier: a full LDS latency per row with the matrix pipe idle (a third of the layer loop).  LDS reads
// return in order, so the right wait is "all but the MT - 1 younger refills".  As in the weight-gradient kernel the reads are inline
// asm (invisible to the compiler's counter model) and the counted waits are placed by hand; the compiler's own LDS / scalar loads only
// make them more conservative.
__device__ __forceinline__ f32x4 frag_read(unsigned addr, int slot) {   // 16 bytes per lane at LDS byte address addr + 1 KiB * slot
    f32x4 v;
#define NNR_FR(k) case k: asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(1024 * (k))); break;
#define NNR_FR8(a) NNR_FR(a) NNR_FR(a + 1) NNR_FR(a + 2) NNR_FR(a + 3) NNR_FR(a + 4) NNR_FR(a + 5) NNR_FR(a + 6) NNR_FR(a + 7)
    switch (slot) {   // the offset is an instruction immediate; `slot` is a constant after unrolling
        NNR_FR8(0) NNR_FR8(8) NNR_FR8(16)
        NNR_FR(24) NNR_FR(25) NNR_FR(26) NNR_FR(27) NNR_FR(28) NNR_FR(29) NNR_FR(30)
        default: asm volatile("ds_read_b128 %0, %1 offset:31744" : "=v"(v) : "v"(addr)); break;
    }
#undef NNR_FR8
#undef NNR_FR
    return v;
}
// Wait until at most n LDS reads are outstanding; the fragment is an in/out operand so that the MFMA that consumes it cannot be
// scheduled above the wait (the MFMA builtin has no other tie to it).  (hipcc then puts an s_nop 0 between the two; an operand-free
// wait fenced by sched_barriers avoids it and measured 1 % SLOWER, one wait per pair of fragments 3 % slower.)
__device__ __forceinline__ void wait_frag(f32x4& frag, int n) {
    switch (n) {
        case 0: asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(frag)); break;
        case 1: asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(frag)); break;
        case 2: asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(frag)); break;
        default: asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(frag)); break;
    }
}
__device__ __forceinline__ unsigned lds_byte_address(const void* p) {
    return (unsigned)(size_t)(const __attribute__((address_space(3))) char*)p;
}

template <int MT, int T, int NACC>
__device__ __forceinline__ void pin_acc2(f32x16 (&acc)[T][NACC]) {   // see pin_acc
    static_assert(T == 1 || T == 2, "one or two tiles per wave");
    if constexpr (T == 1) {
        pin_acc<MT>(acc[0]);
    } else {
        if constexpr (MT == 1) asm volatile("" : "+a"(acc[0][0]), "+a"(acc[1][0]));
        else if constexpr (MT == 2) asm volatile("" : "+a"(acc[0][0]), "+a"(acc[0][1]), "+a"(acc[1][0]), "+a"(acc[1][1]));
        else if constexpr (MT == 4)
            asm volatile("" : "+a"(acc[0][0]), "+a"(acc[0][1]), "+a"(acc[0][2]), "+a"(acc[0][3]), "+a"(acc[1][0]), "+a"(acc[1][1]),
                         "+a"(acc[1][2]), "+a"(acc[1][3]));
        else static_assert(MT == 1 || MT == 2 || MT == 4, "unsupported tile count");
    }
}

template <int T, int N>
__device__ __forceinline__ void zero_acc2(f32x16 (&acc)[T][N]) {
#pragma unroll
    for (int n = 0; n < T; ++n) zero_acc(acc[n]);
}

// acc[n][mt] += A_part[32 mt .., :] * in[n]   for both tiles n, one layer part whose packed bf16 panels start at stream panel p0.
//   in    : per tile 8 * KT packed registers (see above); row g = registers 4g .. 4g+3 = 16 k-values
//   STASH : the row operands also go to a tile-major bf16 plane -- stash[n] = the lane's 16 bytes of block (chunk of tile n, group 0),
//           row g at + 512 g elements: one fully coalesced 1 KiB wave-store per row and tile
//   side  : VALU work hidden under this part's MFMAs, NSIDE units side(0 .. NSIDE-1), PPG per row, starting at row SHIFT (units that
//           do not fit run after the last row).  SHIFT = 0: the units write registers of `in` this part reads later; SHIFT = 1: they
//           overwrite `in` behind the read pointer (the caller's unit order guarantees it, see the kernels).
// A row is MT * kTiles MFMAs of 32 cycles; its non-MFMA work -- MT fragment reads of the next row (behind the panel switch, if any),
// kTiles stash stores, the DMA piece(s) of the panel two ahead, PPG side units -- is placed in different MFMA gaps and pinned with
// sched_barrier(0), as in gemm_part (nnr_device.h).
//   PRE   : VMEM operations (stash stores) this wave has certainly issued AFTER the last DMA piece of this part's first panel -- i.e.
//           while it consumed the panel(s) before: they may stay in flight at the first panel switch.  Without it that switch waits
//           for stores issued a few hundred cycles earlier to be acknowledged (a full HBM write latency with the matrix pipe idle,
//           once per stashing pass: the training kernels lost a third of their time there).  Use stash_tail<>() of the previous part.
template <int KT, int MT, bool STASH, int NSIDE_, int PPG, int SHIFT, int PRE, class Side, class Pipe, int T, int NACC, int NIN>
__device__ __forceinline__ void gemm_wide(f32x16 (&acc)[T][NACC], const uint32_t (&in)[T][NIN], const Pipe& pipe, int p0,
                                          __bf16* const (&stash)[T], const Side& side) {
    constexpr int kTiles = T, PW = Pipe::PW;
#if NNR_ABLATE & 1
    constexpr int NSIDE = 0;   // profiling build only
#else
    constexpr int NSIDE = NSIDE_;
#endif
    static_assert(MT <= NACC && 8 * KT <= NIN, "tile counts exceed the register arrays");
    constexpr int G = 2 * KT, GP = part_gp(MT), NM = MT * kTiles;
    constexpr int kGapDma = NM / 2;
    auto rows_in = [](int pi) { return (G - pi * GP) < GP ? (G - pi * GP) : GP; };
    auto gap_stash = [](int n) { return n * (NM / kTiles); };   // the even gaps are free of fragment reads
    // DMA pieces of the panel two ahead, issued per row of the current panel: kDmaBurst at a time right after the switch, so that the
    // last piece has almost two panels (~2 x 2048 matrix-pipe cycles) to land -- one piece per row left the last one barely one
    // panel, less than the L2 -> LDS latency under load: a quarter of the wave cycles sat in the panel switch's s_waitcnt
    constexpr int kBurst = kDmaBurst < PW ? kDmaBurst : PW;
    auto ppk_of = [&](int pi) { return (PW + rows_in(pi) - 1) / rows_in(pi) > kBurst ? (PW + rows_in(pi) - 1) / rows_in(pi) : kBurst; };
    pipe.template enter<PRE>(p0);
    pipe.pieces(p0 + 2, 0, ppk_of(0));
    Frags<MT> cur;
    const unsigned lane_base = lds_byte_address(pipe.lds) + 16u * pipe.lane;
    unsigned panel_addr = lane_base + pipe.buffer(p0) * (kPanelF4 * 16);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) cur.v[mt] = frag_read(panel_addr, mt);
#pragma unroll
    for (int g = 0; g < G; ++g) {
#pragma unroll
        for (int j = 0; j < NM; ++j) {
            const int mt = j / kTiles, n = j % kTiles;
            __builtin_amdgcn_sched_barrier(0);
            // fragment mt has landed when at most the reads issued after it are outstanding: the MT - 1 other refills -- fewer in
            // the last row, which issues none
            if (n == 0) wait_frag(cur.v[mt], g + 1 < G ? MT - 1 : MT - 1 - mt);
            acc[n][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, cur.v[mt]), row_operand(in[n], g), acc[n][mt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            // Fragment mt is refilled IN PLACE for the next row right after its last MFMA of this row (no second fragment set: 16
            // registers the kernels need elsewhere); the read has (MT - 1) * kTiles MFMAs = 192 cycles to land.  A panel switch
            // (counted wait + barrier) sits in front of the first read from the new panel.
            if (n == kTiles - 1 && g + 1 < G) {
                const int pn = p0 + (g + 1) / GP;
                if (mt == 0 && (g + 1) % GP == 0) {
                    // the stash stores of this panel's earlier rows are younger than the pieces waited for
                    pipe.template enter<STASH ? kTiles * (GP - 1) : 0>(pn);
                    panel_addr = lane_base + pipe.buffer(pn) * (kPanelF4 * 16);
                }
                cur.v[mt] = frag_read(panel_addr, ((g + 1) % GP) * MT + mt);
            }
            if constexpr (STASH) {
#pragma unroll
                for (int t = 0; t < kTiles; ++t)
                    if (j == gap_stash(t)) stash_store(stash[t] + kBlockBf16 * g, u32x4{in[t][4 * g], in[t][4 * g + 1], in[t][4 * g + 2], in[t][4 * g + 3]});
            }
            if (j == kGapDma) {   // DMA pieces of the panel two ahead, spread over the rows of the current panel
                const int pi = g / GP, gi = g % GP;
                const int n_in = rows_in(pi);
                if (gi == n_in - 1) {
                    if (g + 1 < G) pipe.pieces(p0 + pi + 3, 0, ppk_of(pi + 1));
                } else {
                    const int ppk = ppk_of(pi);
                    pipe.pieces(p0 + pi + 2, (gi + 1) * ppk, ppk);
                }
            }
            if constexpr (NSIDE > 0) {
#pragma unroll
                for (int i = 0; i < PPG; ++i) {
                    const int u = (g - SHIFT) * PPG + i;
                    if (j == (i * NM) / PPG && u >= 0 && u < NSIDE) side(u);
                }
            }
        }
        pin_acc2<MT, T>(acc);
        __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (NSIDE > 0) {
#pragma unroll
        for (int u = (G - SHIFT) * PPG; u < NSIDE; ++u)
            if (u >= 0) side(u);
    }
}

template <int KT, int MT, int PRE = 0, class Pipe, int T, int NACC, int NIN>
__device__ __forceinline__ void gemm_wide(f32x16 (&acc)[T][NACC], const uint32_t (&in)[T][NIN], const Pipe& pipe, int p0) {
    __bf16* const none[T] = {};
    gemm_wide<KT, MT, false, 0, 1, 0, PRE>(acc, in, pipe, p0, none, NoSide{});
}

// stash stores a STASH part (KT x MT tiles, T tiles per wave) issues while it consumes its LAST panel = the PRE of the part that follows it
template <int KT, int MT, int T>
__device__ __forceinline__ constexpr int stash_tail() {
    constexpr int G = 2 * KT, GP = part_gp(MT), last = G % GP == 0 ? GP : G % GP;
    return T * last;
}

// ---- chain rule through the encodings --------------------------------------------------------------------------------------
// d gamma_f / d coordinate = scale_f * partner_f: a sin feature's partner is the cos of the same argument (scale +2^l), a cos feature's
// the sin (scale -2^l), the identity block has factor 1, padding 0.  The input-gradient kernel needs, per lane and fragment register,
// exactly that product.  Round 2 had the forward kernel build these factors (48 cross-half shuffles per tile) and store them as fp32
// planes, 384 bytes per sample written and read -- and every one of the 20 loads of a pass cost the input-gradient kernel a full drain
// of its store queue (a compiler-visible load next to pending stores is waited for with vmcnt(0): the two kinds may retire out of
// order as far as hipcc knows).  Now the forward leaves the 12-byte position / view direction (planes P_DPTS / P_DVIEW, which this
// kernel overwrites with the gradients afterwards) and the factors are RECOMPUTED here: sin / cos of the coordinate by the accurate
// routine once, the octaves above it by angle doubling (sin 2a = 2 sin a cos a, cos 2a = 1 - 2 sin^2 a: three VALU operations per
// level instead of ~20).  Doubling multiplies the absolute error by two per level -- 3e-5 at the tenth octave, relative to a factor
// of 512 -- which only enters the fp32 chain rule d point = sum g_f * factor_f, far below the bf16 rounding of g itself; the
// FORWARD's encodings, which the bf16 oracle restates, are not touched.
// d/d(x,y,z) of sum_r g(r) * gamma_{f(r,half)}(x, y, z) for the NR fragment registers of an encoding with NL octaves (3 + 6 NL
// features), the factors recomputed from the coordinates octave by octave -- a handful of live values, no table.  Feature f sits in
// register rp(f) of the lanes of half hp(f) (frag_feature^-1); the other half contributes nothing to it, and the two halves' sums are
// combined by one cross-half shuffle per coordinate at the end.
template <int NR, int NL, class G>
__device__ __forceinline__ f32x4 enc_chain(const G& g, float x, float y, float z, int half) {
    float o[3] = {0.f, 0.f, 0.f};
    const float xyz[3] = {x, y, z};
    // which half a feature belongs to enters as a 0 / 1 WEIGHT of its factor, not as a select of the accumulator value: hipcc turns the
    // selects into whole-vector selects of the 16-register accumulator tuples (two masked copies of each: ~150 spilled registers)
    const float w1 = half ? 1.f : 0.f, w0 = 1.f - w1;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        auto take = [&](int f, float fac) __attribute__((always_inline)) {     // f is a compile-time constant after unrolling
            const int hp = (f % 8) / 4, rp = 16 * (f / 32) + (f % 4) + 4 * ((f % 32) / 8);
            if (rp < NR) {
                o[c] = fmaf(g(rp), fac * (hp ? w1 : w0), o[c]);
            }
        };
        take(c, 1.f);                                                           // the identity block
        float sn = sin_or_cos(xyz[c], false), cs = sin_or_cos(xyz[c], true), scale = 1.f;
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            take(3 + 6 * l + c, scale * cs);          // d sin(2^l a) / da =  2^l cos(2^l a)
            take(3 + 6 * l + 3 + c, -scale * sn);     // d cos(2^l a) / da = -2^l sin(2^l a)
            const float s2 = 2.f * sn * cs;           // the next octave by angle doubling
            cs = __builtin_fmaf(-2.f * sn, sn, 1.f);
            sn = s2;
            scale *= 2.f;
            // Tie the octave recurrence to the running sum.  Left alone, hipcc runs the recurrence ahead -- all octaves of all
            // coordinates of both tiles, packed two by two into v_pk_* operations -- and keeps every factor live until the sums
            // catch up: ~150 spilled registers.  An empty asm that "modifies" both makes octave l + 1 wait for octave l's terms.
            asm volatile("" : "+v"(sn), "+v"(cs), "+v"(o[c]));
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) o[c] = sum_halves(o[c]);
    return f32x4{o[0], o[1], o[2], 0.f};
}

}  // namespace nnr
