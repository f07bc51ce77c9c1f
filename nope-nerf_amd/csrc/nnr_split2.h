// nnr_split2.h -- the GEMM part of the fp32 MLP kernels with every fp32 product taken as THREE fp16 MFMA terms of two-term operands
// (NNR_F_SPLIT2; the kernels: nnr_mlp_fwd_f16.hip, nnr_mlp_dgrad_f16.hip), gfx950 only.
//
// Why (round 6): the six-term bf16 products of nnr_split.h cost SIX matrix-pipe passes per fp32 product and a three-stage split of every
// activation in front of every pass that consumes it (11 instructions per value pair, done twice per layer) -- those kernels are bound by the
// issue of these instructions beside the MFMAs, not by the matrix pipe.  fp16 has 11
// significand bits where bf16 has 8: TWO terms carry 22 bits,
//     x s = x_h + 2^-11 x_m',   x_h = fp16(x s),   x_m' = fp16((x s - x_h) 2^11)          (both differences exact in fp32)
// and  w x = [w_h x_h + w_m x_h + w_hs x_m'] / (s_w s)  with  w_h = fp16(w s_w), w_m = fp16(w s_w - w_h), w_hs = fp16(w s_w 2^-11):
// THREE v_mfma_f32_32x32x16_f16 per 16 k-values instead of six bf16 ones; the dropped w_m x_m term and the two-term representation leave
// a relative error of about 2^-22 per product -- unbiased, below what the fp32 accumulation of a 256-long dot product adds itself
// (tools/micro/split2_f16.hip: the bit budget against the fp32 instruction; tools/split2_emulation.py: the whole training step against
// fp64, indistinguishable from exact products).
//   * Range.  fp16 spans 2^-24 .. 65504, so the operands are SCALED by powers of two (exact): the weights per tensor by the pack kernel
//     (max |w| s_w in [2^13, 2^14); the biases arrive pre-multiplied by s_w), the forward's activations not at all (s = 1: hidden
//     activations of this network are O(1); one above 65504 overflows to inf and the trainer's NaN check stops the run -- the six-term
//     mode has no such bound and stays selectable), the input gradients per SAMPLE (nnr_mlp_dgrad_f16.hip).  Carrying the residual at 2^11
//     keeps it a NORMAL fp16 number down to x_h = 2^-14: 22 bits for 6e-5 <= |x s| < 65504, an absolute floor of 2^-36 below.
//   * The split happens ONCE per value, in the epilogue that produces it (a pair of fp16 terms is 32 bits per value: the packed terms
//     replace the fp32 activation registers between the layers), not in front of every pass: a row of the GEMM carries no split at all.
//     What a training kernel stashes is written by the same epilogue unit while the fp32 value exists.
#pragma once
#include "nnr_split.h"

namespace nnr {

typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr float kResidualUp = 2048.f;      // 2^11: the scale the second term is carried at

// the two fp16 terms of a value pair, packed (x0 in the low halves): h = rn(x), m = rn((x - h) 2^11).  Infinite or NaN input gives NaN terms.
__device__ __forceinline__ void split2_pair(float x0, float x1, uint32_t& h, uint32_t& m) {
    const f32x2 v = {x0, x1};
    const f16x2 hh = __builtin_convertvector(v, f16x2);
    const f32x2 back = __builtin_convertvector(hh, f32x2);
    const f32x2 r = (v - back) * kResidualUp;
    h = __builtin_bit_cast(uint32_t, hh);
    m = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, f16x2));
}

// the packed B operands of a feature vector in fragment layout: pair p = registers 2 p, 2 p + 1 (nnr_layout.h); the four pairs 4 g .. 4 g + 3
// are the 8 k-values a lane contributes to row g of a 16-deep MFMA.  (Two plain arrays: a struct of arrays kept SROA from promoting slices.)
template <int NP, class F>
__device__ __forceinline__ void split2_all(uint32_t (&ph)[NP], uint32_t (&pm)[NP], const F& value) {
#pragma unroll
    for (int p = 0; p < NP; ++p) split2_pair(value(2 * p), value(2 * p + 1), ph[p], pm[p]);
}

// The ReLU gates of a finished pair appended to a mask word: word = (word << 2) | (x0 != 0) << 1 | (x1 != 0), x >= 0 (as integers: min(bits, 1)).
// Register r of a half (32 registers per word, appended in register order) ends up at bit 31 - (r & 31): the input-gradient kernel's select is
// (int32)(word << (r & 31)) >> 31.  One asm statement: written in C++ hipcc turns min(bits, 1) << r into a compare, an s_nop, a select between
// 0 and a per-register constant it keeps in a VGPR (32 registers of constants) and an or.
__device__ __forceinline__ void gate_append2(uint32_t& word, float x0, float x1) {
    uint32_t t;
    asm("v_min_u32_e32 %1, 1, %2\n\tv_lshl_or_b32 %0, %0, 1, %1\n\tv_min_u32_e32 %1, 1, %3\n\tv_lshl_or_b32 %0, %0, 1, %1"
        : "+v"(word), "=&v"(t)
        : "v"(x0), "v"(x1));
}

// ---- the epilogue units as single asm statements ---------------------------------------------------------------------------------------
// Written in C++ the unit of a pair cost 21 issue slots in the training forward (hipcc: the two accumulator reads hoisted out of the part
// as a 64-register copy that pushed the packed terms into AGPR spills; the gates as compare / select / or with s_nops; the residual through
// two conversions, a packed subtract and a packed multiply, each packed result padded with an s_nop before its consumer) -- and these kernels
// are bound by issue slots.  One statement per unit, 16 slots, no padding: the residual (x - h) 2^11 as ONE mixed-precision FMA per value
// (v_fma_mix_f32: -2048 x fp16 half of h + 2048 x, exact), independent instructions between a packed-encoding result and its consumer.
// Hazards the compiler cannot see inside asm, covered by construction: an accumulator read here is at least 25 instructions behind the last
// MFMA that wrote it (the accumulators of the OTHER half-pass; a part's first unit sits behind its panel switch, 12 fragment reads and
// 3 MFMAs); a VOP3P result (v_fma_mix_f32) is read one instruction later at the earliest.
// forward, training: pair of accumulators -> x = relu(acc / s_w) (returned: stash, density head), two gate bits appended, packed terms
// (mx: running maximum of the values stashed into the current activation plane: the weight-gradient kernel scales its fp16 terms by the plane's largest)
__device__ __forceinline__ void unit_fwd_train(float a0, float a1, float inv, uint32_t& word, float& x0, float& x1, uint32_t& h, uint32_t& m, float& mx) {
    float c0, c1;
    uint32_t t;
    asm volatile(
        "v_accvgpr_read_b32 %0, %9\n\t"
        "v_accvgpr_read_b32 %1, %10\n\t"
        "v_mul_f32_e64 %0, %0, %11\n\t"
        "v_mul_f32_e64 %1, %1, %11\n\t"
        "v_max_f32_e32 %0, 0, %0\n\t"
        "v_max_f32_e32 %1, 0, %1\n\t"
        "v_cvt_pk_f16_f32 %2, %0, %1\n\t"
        "v_mul_f32_e64 %4, %0, %12\n\t"
        "v_mul_f32_e64 %5, %1, %12\n\t"
        "v_min_u32_e32 %6, 1, %0\n\t"
        "v_lshl_or_b32 %7, %7, 1, %6\n\t"
        "v_fma_mix_f32 %4, %2, %13, %4 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %5, %2, %13, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_min_u32_e32 %6, 1, %1\n\t"
        "v_lshl_or_b32 %7, %7, 1, %6\n\t"
        "v_max3_f32 %8, %0, %1, %8\n\t"
        "v_cvt_pk_f16_f32 %3, %4, %5"
        : "=&v"(x0), "=&v"(x1), "=&v"(h), "=&v"(m), "=&v"(c0), "=&v"(c1), "=&v"(t), "+v"(word), "+v"(mx)
        : "a"(a0), "a"(a1), "s"(inv), "s"(kResidualUp), "s"(-kResidualUp));
}
// forward, inference: no gates.  mx: running maximum of the activations (the kernel turns a sample whose activations left fp16's range into NaN
// output: nnr_mlp_fwd_f16.hip) -- its v_max3 takes the slot of the s_nop that had to sit between the second mixed FMA and the conversion that reads it
__device__ __forceinline__ void unit_fwd_infer(float a0, float a1, float inv, float& x0, float& x1, uint32_t& h, uint32_t& m, float& mx) {
    float c0, c1;
    asm volatile(
        "v_accvgpr_read_b32 %0, %7\n\t"
        "v_accvgpr_read_b32 %1, %8\n\t"
        "v_mul_f32_e64 %0, %0, %9\n\t"
        "v_mul_f32_e64 %1, %1, %9\n\t"
        "v_max_f32_e32 %0, 0, %0\n\t"
        "v_max_f32_e32 %1, 0, %1\n\t"
        "v_cvt_pk_f16_f32 %2, %0, %1\n\t"
        "v_mul_f32_e64 %4, %0, %10\n\t"
        "v_mul_f32_e64 %5, %1, %10\n\t"
        "v_fma_mix_f32 %4, %2, %11, %4 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %5, %2, %11, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_max3_f32 %6, %0, %1, %6\n\t"
        "v_cvt_pk_f16_f32 %3, %4, %5"
        : "=&v"(x0), "=&v"(x1), "=&v"(h), "=&v"(m), "=&v"(c0), "=&v"(c1), "+v"(mx)
        : "a"(a0), "a"(a1), "s"(inv), "s"(kResidualUp), "s"(-kResidualUp));
}
// input gradient: pair of accumulators -> relu'(.) ? acc : 0 (gates of registers r, r + 1 at bits POS0, POS1 of `word`: gate_append2's order)
// -> x = . / s_w (the scaled gradient) -> packed terms; t = x / s (the true gradient, for the plane); mx: running maximum of |t| (the plane's scale
// in the weight-gradient kernel)

// the largest value of a wave, in every lane (non-negative floats): five cross-lane steps + the half-wave swap
__device__ __forceinline__ float wave_max_f32(float v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = fmaxf(v, __shfl_xor(v, d, 64));
    return v;
}
template <int POS0, int POS1>
__device__ __forceinline__ void unit_dgrad(float a0, float a1, uint32_t word, float invw, float sinv, float& t0, float& t1, uint32_t& h, uint32_t& m, float& mx) {
    float x0, x1, c0, c1;
    asm volatile(
        "v_accvgpr_read_b32 %4, %9\n\t"
        "v_accvgpr_read_b32 %5, %10\n\t"
        "v_bfe_i32 %6, %11, %16, 1\n\t"
        "v_bfe_i32 %7, %11, %17, 1\n\t"
        "v_and_b32_e32 %4, %4, %6\n\t"
        "v_and_b32_e32 %5, %5, %7\n\t"
        "v_mul_f32_e64 %4, %4, %12\n\t"
        "v_mul_f32_e64 %5, %5, %12\n\t"
        "v_cvt_pk_f16_f32 %2, %4, %5\n\t"
        "v_mul_f32_e64 %6, %4, %13\n\t"
        "v_mul_f32_e64 %7, %5, %13\n\t"
        "v_mul_f32_e32 %0, %4, %15\n\t"
        "v_fma_mix_f32 %6, %2, %14, %6 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %7, %2, %14, %7 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_mul_f32_e32 %1, %5, %15\n\t"
        "v_cvt_pk_f16_f32 %3, %6, %7\n\t"
        "v_max3_f32 %8, |%0|, |%1|, %8"
        : "=&v"(t0), "=&v"(t1), "=&v"(h), "=&v"(m), "=&v"(x0), "=&v"(x1), "=&v"(c0), "=&v"(c1), "+v"(mx)
        : "a"(a0), "a"(a1), "v"(word), "v"(invw), "s"(kResidualUp), "s"(-kResidualUp), "v"(sinv), "n"(POS0), "n"(POS1));
}

template <bool TILE>
using Split2PipeT = PanelPipeT<kWavesPerBlock, kPanelFrags, TILE>;     // 32-slot panels: two fragment classes per row and m-tile, 16 / MT rows (nnr_layout.h, MODE 3)

// w_h 2^-11 from a w_h fragment (8 fp16 per lane): the weight operand of the residual term, which is carried at 2^11.  An exponent shift -- exact
// unless the result is subnormal, i.e. for weights 2^17 below the tensor's largest -- made here, four packed multiplies per fragment, instead of
// being streamed as a third fragment class: these kernels read one 1 KiB fragment from LDS per MFMA as it is (every fragment feeds one
// 32-sample tile), and the stream through L2 and the LDS ring is a third shorter for it.
__device__ __forceinline__ f32x4 frag_down11(f32x4 f) {
    // (as ONE 8-wide multiply: written register by register -- u32x4 u = bit_cast(f); o[i] = bit_cast(bit_cast<f16x2>(u[i]) * k) -- hipcc 7.2
    // multiplied the FIRST register and copied it into the other three; the stage-by-stage bisect showed the residual term missing)
    const f16x8 v = __builtin_bit_cast(f16x8, f) * (_Float16)0.00048828125f;      // 2^-11
    return __builtin_bit_cast(f32x4, v);
}

// acc[mt] += A_part[32 mt .., :] * in, the products as three fp16 terms.  `ph` / `pm`: the packed terms of the part's input (8 KT pairs of
// registers; pairs 4 g .. 4 g + 3 belong to row g).  A row = 16 k-values = 3 MT MFMAs in the order (weights operand, activation term)
//     t0 (w_m, x_h)   t1 (w_h 2^-11, x_m')   t2 (w_h, x_h)        -- small products first
// Fragment class c (packed by the pack kernel as slot ((row % GP) * 2 + c) * MT + mt of the panel): 0 = w_m, used by t0; 1 = w_h, used by t2 and,
// shifted down in registers right before it (frag_down11), by t1.  Both classes of the NEXT row are requested behind this row's t0 MFMAs (class 0
// in place, class 1 into the second of two register sets), one wait at the head of a row.
// Side units: a unit finishes a pair of the PREVIOUS pass's accumulators and writes its packed terms.  Which row runs which units (SCHED):
//   0  UPR units per row from row 0 on (unit u in row u / UPR): the units write ANOTHER array than the part reads (no constraint);
//   1  "ahead": the units write the UPPER half of the part's own input (pair NSIDE + u, first read by row (NSIDE + u) / 4; NSIDE = 2 G pairs per
//      half): all of them in the rows [0, G - 1), 2 - 3 per row, so that every pair is complete a row before it is read;
//   2  "behind": the units overwrite the LOWER half in place (pair u, read by row u / 4: every MFMA of a row reads the row's pairs, so a pair can
//      be rewritten from the next row on): all of them in the rows [1, G).
// (first_unit(g) .. first_unit(g + 1) run in row g; tools: the two schedules brute-forced against their constraints for G = 16 and 8.)
// STORE_EVERY: every STORE_EVERY-th unit issues one VMEM store (the training kernels' stash; 0: none); PRE: stores the part BEFORE certainly
// issued after the last DMA piece of this part's first panel -- for the counted waits: a store younger than the DMA pieces waited for may
// stay in flight (PanelPipeT::enter<EXTRA>).
template <int SCHED, int NSIDE, int UPR, int G>
__device__ __forceinline__ constexpr int first_unit(int g) {
    if (NSIDE == 0 || g <= 0) return 0;
    if (SCHED == 0) return g * UPR < NSIDE ? g * UPR : NSIDE;
    if (SCHED == 1) return g >= G - 1 ? NSIDE : ((g * NSIDE + (G - 2)) / (G - 1) < NSIDE ? (g * NSIDE + (G - 2)) / (G - 1) : NSIDE);
    return g >= G ? NSIDE : ((g - 1) * NSIDE) / (G - 1);
}

template <int KT, int MT, int NSIDE_, int SCHED, int UPR, int STORE_EVERY, int PRE, class Side, class Pipe, int NACC, int NIN>
__device__ __forceinline__ void gemm_part2(f32x16 (&acc)[NACC], const uint32_t (&ph)[NIN], const uint32_t (&pm)[NIN], const Pipe& pipe, int p0,
                                           const Side& side, bool pre_valid = true) {
#if NNR_ABLATE & 1
    constexpr int NSIDE = 0;   // profiling build only
#else
    constexpr int NSIDE = NSIDE_;
#endif
    static_assert(MT <= NACC && 8 * KT <= NIN, "tile counts exceed the register arrays");
    static_assert(MT == 1 || MT == 2 || MT == 4, "m-tiles per part");
    constexpr int G = 2 * KT, GP = mode_gp(MT, 3), NM = 3 * MT, PW = Pipe::PW;
    static_assert(SCHED == 0 || G >= 2, "the constrained schedules need two rows");
    static_assert(SCHED == 0 || NSIDE_ <= 4 * (G - 1), "more units than the rows before the last one can finish in time");
    auto rows_in = [](int pi) { return (G - pi * GP) < GP ? (G - pi * GP) : GP; };
    auto ppk_of = [&](int pi) { return (PW + rows_in(pi) - 1) / rows_in(pi); };
    constexpr int ppk_full = (PW + GP - 1) / GP;
    auto fu = [](int g) { return first_unit<SCHED, NSIDE, UPR, G>(g); };
    // stash stores the units of row r issue
    auto stores_of_row = [&](int r) {
        if (STORE_EVERY == 0 || r < 0 || r >= G) return 0;
        int n = 0;
        for (int u = fu(r); u < fu(r + 1); ++u)
            if (u % (STORE_EVERY ? STORE_EVERY : 1) == STORE_EVERY - 1) ++n;
        return n;
    };

    if (PRE > 0 && pre_valid) pipe.template enter<PRE>(p0);      // (pre_valid is wave-uniform: the part before did issue those stores)
    else pipe.enter(p0);
    pipe.pieces(p0 + 2, 0, ppk_of(0));
    const unsigned lane_base = lds_byte_address(pipe.lds) + 16u * pipe.lane;
    unsigned panel_addr = lane_base + pipe.buffer(p0) * (Pipe::F4 * 16);
    // fragments: class 0 (w_m) refilled in place behind its MFMA; class 1 (w_h) in TWO sets alternating by row -- it is used by two of a row's three
    // terms, so a refill in place would be requested only MT MFMAs before its first use in the next row (measured: a quarter of the wave cycles parked
    // in s_waitcnt); with the second set the next row's class-1 fragments are requested a whole row ahead, and ONE wait at the head of a row covers both classes
    f32x4 fr0[MT], fr1[2][MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) fr0[mt] = frag_read(panel_addr, mt);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) fr1[0][mt] = frag_read(panel_addr, MT + mt);

#pragma unroll
    for (int g = 0; g < G; ++g) {
        const bool last = g + 1 == G;
        const int u0 = fu(g), nu = fu(g + 1) - fu(g);      // this row's units
#pragma unroll
        for (int j = 0; j < NM; ++j) {
            const int t = j / MT, mt = j % MT;
            __builtin_amdgcn_sched_barrier(0);
            // head of the row: every read issued so far is a row old (or the prologue's): all of this row's fragments have landed
            if (j == 0) {
                if constexpr (MT == 1) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fr0[0]), "+v"(fr1[g & 1][0]));
                else if constexpr (MT == 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fr0[0]), "+v"(fr0[1]), "+v"(fr1[g & 1][0]), "+v"(fr1[g & 1][1]));
                else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fr0[0]), "+v"(fr0[1]), "+v"(fr0[2]), "+v"(fr0[3]), "+v"(fr1[g & 1][0]), "+v"(fr1[g & 1][1]),
                                  "+v"(fr1[g & 1][2]), "+v"(fr1[g & 1][3]));
            }
            const u32x4 b = t == 1 ? u32x4{pm[4 * g], pm[4 * g + 1], pm[4 * g + 2], pm[4 * g + 3]} : u32x4{ph[4 * g], ph[4 * g + 1], ph[4 * g + 2], ph[4 * g + 3]};
            const f32x4 a = t == 0 ? fr0[mt] : (t == 1 ? frag_down11(fr1[g & 1][mt]) : fr1[g & 1][mt]);
            acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc[mt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (!last && t == 0) {      // behind a t0 MFMA: refill its class-0 fragment in place and request the next row's class-1 fragment into the other set;
                                        // a panel switch (counted wait + barrier) in front of the first read from a new panel
                const int pn = p0 + (g + 1) / GP;
                if (j == 0 && (g + 1) % GP == 0) {
                    // Stores that may stay in flight: those certainly issued after the last DMA piece of panel pn -- pn's pieces go out while the panel
                    // two back is consumed (first burst in the row before it, then ppk per row), so every store of the rows behind that panel's last
                    // burst row and of this panel's earlier rows is younger.  Rows of the part before this one are not counted (conservative).
                    constexpr int kLastBurstRow = (PW + ppk_full - 1) / ppk_full - 2;
                    int extra = 0;
                    for (int r = g - (GP - 1) - (GP - 1 - kLastBurstRow); r < g; ++r) extra += stores_of_row(r);
                    if ((g + 1) / GP < 2) {      // the part's second panel: its pieces went out before this part began: only this panel's earlier rows are counted
                        extra = 0;
                        for (int r = g - (GP - 1); r < g; ++r) extra += stores_of_row(r);
                    }
                    switch (extra) {
                        case 0: pipe.template enter<0>(pn); break;
                        case 1: pipe.template enter<1>(pn); break;
                        case 2: pipe.template enter<2>(pn); break;
                        case 3: pipe.template enter<3>(pn); break;
                        case 4: pipe.template enter<4>(pn); break;
                        case 5: pipe.template enter<5>(pn); break;
                        default: pipe.template enter<6>(pn); break;
                    }
                    panel_addr = lane_base + pipe.buffer(pn) * (Pipe::F4 * 16);
                }
                fr0[mt] = frag_read(panel_addr, (((g + 1) % GP) * 2) * MT + mt);
                fr1[(g + 1) & 1][mt] = frag_read(panel_addr, (((g + 1) % GP) * 2 + 1) * MT + mt);
            }
            // the DMA pieces of the panel two ahead, spread over the rows of the current panel: the row's share ONE piece per gap from gap kDmaGap on
            // (a piece costs the issuing wave 60 - 180 cycles of issue, more next to LDS reads: MI355X_MICROARCH.md; the fragment reads sit in the gaps
            // of the t0 phase)
            // (kDmaGap = 1: the row's pieces as ONE burst in gap 1; kUnitGap = 2: first gap the side units may use.  Other placements -- one piece
            // per gap, units from gap 0 / 1 / 3 -- measured the same within noise: profiles/r06/f2_f16_gap_placement_variants.txt)
            constexpr int kDmaGap = 1, kUnitGap = 2;
            {
                const int pi = g / GP, gi = g % GP;
                const bool into_next = gi == rows_in(pi) - 1;      // this row entered panel pi + 1 above: first share of the panel three ahead
                const int pnl = into_next ? p0 + pi + 3 : p0 + pi + 2, n = into_next ? ppk_of(pi + 1) : ppk_of(pi), first = into_next ? 0 : (gi + 1) * ppk_of(pi);
                const int g0 = kDmaGap < NM ? kDmaGap : NM - 1;
                if (!(into_next && last) && j == g0) pipe.pieces(pnl, first, n);
            }
            if constexpr (NSIDE > 0) {
#pragma unroll
                for (int k = 0; k < nu; ++k) {      // unit k of the row's nu: spread over the gaps from kUnitGap on
                    constexpr int U0 = kUnitGap < NM ? kUnitGap : 0;
                    const int gap = NM >= 4 ? U0 + (k * (NM - U0)) / nu : (k * NM) / nu;
                    if (gap == j) side(u0 + k);
                }
            }
        }
        pin_acc<MT>(acc);
        __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (NSIDE > 0) {
#pragma unroll
        for (int u = fu(G); u < NSIDE; ++u) side(u);
    }
}

template <int KT, int MT, class Pipe, int NACC, int NIN>
__device__ __forceinline__ void gemm_part2(f32x16 (&acc)[NACC], const uint32_t (&ph)[NIN], const uint32_t (&pm)[NIN], const Pipe& pipe, int p0) {
    gemm_part2<KT, MT, 0, 0, 1, 0, 0>(acc, ph, pm, pipe, p0, NoSide{});
}

// one whole 1 KiB block of a tile-major fp32 plane (nnr_layout.h: tile32_index), non-temporal: (scalar base of the plane's block 0 of this chunk, bumped
// by 4 KiB every fourth block) + (32-bit lane offset = 16 bytes per lane) + immediate -- the store instruction of nnr_split.h, written out for
// the same reasons (no FLAT store, no 64-bit vector address arithmetic; hipcc does not count it: the panel switches' counted waits do).
// The s_nop behind it is NOT optional: a VMEM store of more than 64 bits reads its data registers after it has issued, and a VALU write to one of
// them within the next two wait states corrupts what is stored (gfx940 hazard; hipcc pads it for stores it knows, it does not look into asm) --
// the units' temporaries are rewritten by the very next instruction (first build: every stashed plane wrong, the computation right).
__device__ __forceinline__ void tile_store(const char* base, int lane_off, int block, f32x4 val) {
    const uint64_t sb = reinterpret_cast<uint64_t>(base) + 4096u * (unsigned)(block >> 2);
    switch (block & 3) {
        case 0: asm volatile("global_store_dwordx4 %0, %1, %2 nt\n\ts_nop 1" : : "v"(lane_off), "v"(val), "s"(sb) : "memory"); break;
        case 1: asm volatile("global_store_dwordx4 %0, %1, %2 offset:1024 nt\n\ts_nop 1" : : "v"(lane_off), "v"(val), "s"(sb) : "memory"); break;
        case 2: asm volatile("global_store_dwordx4 %0, %1, %2 offset:2048 nt\n\ts_nop 1" : : "v"(lane_off), "v"(val), "s"(sb) : "memory"); break;
        default: asm volatile("global_store_dwordx4 %0, %1, %2 offset:3072 nt\n\ts_nop 1" : : "v"(lane_off), "v"(val), "s"(sb) : "memory"); break;
    }
}

}  // namespace nnr
