// nnr_split.h -- the GEMM part of the fp32 MLP kernels with every fp32 product taken as SIX bf16 MFMA terms (NNR_F_SPLIT3), gfx950 only.
//
// Why: v_mfma_f32_32x32x2_f32 is 1/16 of the matrix pipe's bf16 rate, and the fp32 kernels (nnr_mlp_fwd.hip, nnr_mlp_dgrad.hip) sit at
// 95 % of that fp32 peak -- nothing left to tune.  An fp32 value is the EXACT sum of three bf16 values (nnr_layout.h, MODE 2), so
//     w x = w_h x_h + (w_h x_m + w_m x_h) + (w_h x_l + w_m x_m + w_l x_h) + [w_m x_l + w_l x_m + w_l x_l  <  2^-24 |w x|, dropped]
// and six v_mfma_f32_32x32x16_bf16 (fp32 accumulate) do the work of eight fp32 MFMAs in 2.7 times fewer cycles.  The result differs from
// the fp32 instruction's only in rounding order: measured against an fp64 evaluation both are equally close (tests/test_gpu_split3.py).
//
// What changes for the kernels: nothing but this function and the packed weights.  Activations stay fp32 in the fragment layout of
// nnr_layout.h -- eight consecutive registers of a lane ARE the 8 k-values a lane contributes to a 16-deep bf16 MFMA -- stash stores,
// side units, masks, heads are the fp32 kernels' own code; the weights arrive pre-split by the pack kernel (three fragments per row and
// m-tile: l, m, h).  The activations of a row are split here, on the VALU, while the previous row's MFMAs run (9-11 instructions per value
// pair, 4 pairs per row against 6 * MT MFMAs).
#pragma once
#include "nnr_mlp_bf16.h"

namespace nnr {

template <bool TILE>
using SplitPipeT = PanelPipeT<kWavesPerBlock, kSplitPanelFrags, TILE>;
using SplitPipe = SplitPipeT<false>;

// the three bf16 terms of two fp32 values, packed (x0 in the low halves): h = rn(x), m = rn(x - h), l = rn(x - h - m); both differences
// are exact in fp32.  (An infinite or NaN input gives NaN terms -- as good as the inf the fp32 path would produce: the trainer stops.)
// (the conversion as the compiler's own v_cvt_pk_bf16_f32, round to nearest even: written as inline asm -- pack_bf16 -- hipcc puts an
// s_nop behind every use, and the two subtractions of a pair as one packed instruction)
__device__ __forceinline__ uint32_t pack_pair(f32x2 v) { return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2)); }
// What a pair leaves behind its packed rounding, r - float(bf16(r)): shift, mask, one packed subtract.
// Tried at the end of round 4 (NNR_SPLIT_DOT2 builds): one v_dot2c_f32_bf16 per value (D += A.x B.x + A.y B.y with B = {-1, 0} resp. {0, -1}
// from scalar registers) -- 7 instructions per split pair instead of 9, bit-identical for every finite input, denormals included, and the
// same issue rate in isolation (tools/micro/dot2_residual.hip; profiles/r04/s_dot2_residual.txt).  Inside these kernels it is SLOWER
// (forward 0.817 -> 0.862 ms in inference, input gradient 0.890 -> 0.912, weight gradient 1.126 -> 1.171; profiles/r04/t_*): the dot
// unit does not overlap with the matrix pipe the way the plain VALU does; and the D = 128 layer-local test failed with it.  Not used.
// (Beware of the selectors as compile-time constants: hipcc folds {-1, 0} into the inline constant -1.0, which the instruction reads as
// the 32-bit pattern 0xbf800000 = {0, -1}.)
__device__ __forceinline__ f32x2 pair_residual(f32x2 r, uint32_t packed) {
    return r - f32x2{__uint_as_float(packed << 16), __uint_as_float(packed & 0xffff0000u)};
}
__device__ __forceinline__ void split_pair(float x0, float x1, uint32_t& h, uint32_t& m, uint32_t& l) {
    f32x2 r = {x0, x1};
    h = pack_pair(r);
    r = pair_residual(r, h);
    m = pack_pair(r);
    r = pair_residual(r, m);
    l = pack_pair(r);
}

// bits 0, 1 = (low half != 0), (high half != 0) of a packed bf16 pair: the ReLU gates of the two activations it holds (an activation is
// >= 0 after the ReLU, and its h term is zero exactly when it is: bf16 has fp32's exponent range).  Two gates in four instructions where
// the epilogue's compare / select / or took three per value.
__device__ __forceinline__ uint32_t gate_pair(uint32_t hpk) {
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    const uint32_t u = __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(u16x2, hpk), u16x2{1, 1}));
    return (u & 1u) | (u >> 15);
}

// at most n LDS reads outstanding (n folds after unrolling; lgkmcnt is a 4-bit field); tied to the fragment like wait_frag
__device__ __forceinline__ void wait_lgkm_n(f32x4& frag, int n) {
    switch (n) {
#define NNR_WL(k) case k: asm volatile("s_waitcnt lgkmcnt(" #k ")" : "+v"(frag)); break;
        NNR_WL(0) NNR_WL(1) NNR_WL(2) NNR_WL(3) NNR_WL(4) NNR_WL(5) NNR_WL(6) NNR_WL(7) NNR_WL(8) NNR_WL(9) NNR_WL(10) NNR_WL(11)
        NNR_WL(12) NNR_WL(13) NNR_WL(14)
#undef NNR_WL
        default: asm volatile("s_waitcnt lgkmcnt(15)" : "+v"(frag)); break;
    }
}

// the same for all MT fragments of a class at once (one wait, one compiler-inserted s_nop)
template <int MT>
__device__ __forceinline__ void wait_class(f32x4 (&f)[MT], int n) {
    switch (n) {
#define NNR_WC(k)                                                                                                                 \
    case k:                                                                                                                       \
        if constexpr (MT == 1) asm volatile("s_waitcnt lgkmcnt(" #k ")" : "+v"(f[0]));                                            \
        else if constexpr (MT == 2) asm volatile("s_waitcnt lgkmcnt(" #k ")" : "+v"(f[0]), "+v"(f[1]));                           \
        else asm volatile("s_waitcnt lgkmcnt(" #k ")" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]));                          \
        break;
        NNR_WC(0) NNR_WC(1) NNR_WC(2) NNR_WC(4)
#undef NNR_WC
        default:
            if constexpr (MT == 1) asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(f[0]));
            else if constexpr (MT == 2) asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(f[0]), "+v"(f[1]));
            else asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]));
            break;
    }
}

// ---- where the work of a row goes ----------------------------------------------------------------------------------------------------
// One wave per SIMD hides about five 4-cycle instructions under each 32-cycle MFMA and pays for every further one
// (tools/ubench/mfma_valu_overlap.hip), so what matters is not how much a row carries besides its 6 MT MFMAs but how EVENLY it is spread
// over the 6 MT gaps.  The work is cut into small operations -- the three dependent stages of a pair's split (5, 5 and 1 instructions),
// the stash stores, the DMA burst, the side units -- listed in an order that interleaves the kinds, and a compile-time pass assigns each to
// the first gap whose instruction budget it still fits in (the fragment refills are fixed: one ds_read in each gap after a term-0 / 2 /
// 5 MFMA, a counted wait in front of the first MFMA of each fragment class).
constexpr int kStashCost = 2;      // instructions a stash store counts for in the balance of the gaps
struct RowOp { int kind, idx, cost; };      // kind 0: split stage A of pair idx, 1: stage B, 2: stage C, 3: stash store idx, 4: DMA, 5: side unit idx
template <int NOPS, int NM>
struct RowSched { RowOp op[NOPS]; int gap[NOPS]; };

template <int MT, bool STASH, int NUNITS, int UCOST>
constexpr auto make_row_sched() {
    constexpr int NM = 6 * MT, NOPS = 12 + (STASH ? 2 : 0) + 1 + NUNITS;
    RowSched<NOPS, NM> r{};
    // the split stages in order A0..A3, B0..B3, C0..C3 (stage k + 1 of a pair well after stage k), everything else merged in between
    RowOp x[12] = {}, y[3 + NUNITS + 1] = {};
    for (int i = 0; i < 12; ++i) x[i] = RowOp{i / 4, i % 4, i < 8 ? 5 : 1};
    int ny = 0;
    for (int u = 0; u < NUNITS; ++u) {
        y[ny++] = RowOp{5, u, UCOST};
        if (STASH && (u == NUNITS / 4 || u == (3 * NUNITS) / 4)) y[ny] = RowOp{3, u == NUNITS / 4 ? 0 : 1, kStashCost}, ++ny;
        if (u == NUNITS / 2) y[ny++] = RowOp{4, 0, 4};
    }
    if (NUNITS == 0) {
        if (STASH) y[ny++] = RowOp{3, 0, kStashCost};
        y[ny++] = RowOp{4, 0, 4};
        if (STASH) y[ny++] = RowOp{3, 1, kStashCost};
    }
    int n = 0, ix = 0, iy = 0;
    while (ix < 12 || iy < ny) {      // merge by fractional position
        const bool take_x = iy >= ny || (ix < 12 && (2 * ix + 1) * ny <= (2 * iy + 1) * 12);
        r.op[n++] = take_x ? x[ix++] : y[iy++];
    }
    int total = 0, fixed[NM] = {};
    for (int j = 0; j < NM; ++j) {
        const int t = j / MT;
        fixed[j] = (t == 0 || t == 2 || t == 5) ? 1 : 0;      // the refill read
        if (j % MT == 0 && (t == 0 || t == 1 || t == 3)) fixed[j > 0 ? j - 1 : 0] += 2;   // the counted wait (+ s_nop) in front of MFMA j
        total += fixed[j];
    }
    for (int i = 0; i < NOPS; ++i) total += r.op[i].cost;
    const int cap = (total + NM - 1) / NM;
    // the DMA burst must come after the panel switch in gap 0: start filling at gap 1
    int j = 1, load = fixed[1 < NM ? 1 : 0];
    for (int i = 0; i < NOPS; ++i) {
        while (j < NM - 1 && load > 0 && load + r.op[i].cost > cap + 1) { ++j; load = fixed[j]; }
        r.gap[i] = j;
        load += r.op[i].cost;
    }
    return r;
}

// acc[mt] += A_part[32 mt .., :] * in, the products as six bf16 terms.  Same contract as gemm_part (nnr_device.h) -- `in` in fragment
// layout, the optional row-major fp32 stash of `in`, NSIDE side units at PPG per k-group of 8 (so 2 PPG per row here) starting at
// k-group SHIFT -- with one difference the callers already respect: a row's eight registers are READ (split) while the row before it
// runs, so side units may overwrite them from the row itself on (SHIFT = 1 rewrites registers 8g - 4 .. 8g + 3 in row g), and a
// stashing part must not rewrite its input at all.
// A row = 16 k-values = 6 MT MFMAs, in this order (weights term, activation term), small products first:
//     t0 (l, h)   t1 (m, m)   t2 (m, h)   t3 (h, l)   t4 (h, m)   t5 (h, h)
// so the l fragments are free after t0, the m fragments after t2, the h fragments after t5: each is refilled IN PLACE for the next row
// right after its last MFMA (reads in the order l, m, h -- the order of first use), which gives every read at least 3 MT MFMAs to land.
// UCOST: instructions of one side unit, for the balance of the gaps (see above).
template <int KT, int MT, bool STASH, int NSIDE_, int PPG, int SHIFT, class Side, int NACC, int NIN, bool TILE>
__device__ __forceinline__ void gemm_part(f32x16 (&acc)[NACC], const float (&in)[NIN], const SplitPipeT<TILE>& pipe, int p0, float* stash,
                                          const Side& side) {
#if NNR_ABLATE & 1
    constexpr int NSIDE = 0;   // profiling build only
#else
    constexpr int NSIDE = NSIDE_;
#endif
    static_assert(MT <= NACC && 16 * KT <= NIN, "tile counts exceed the register arrays");
    static_assert(!(STASH && SHIFT != 0), "a part that stashes its input must not rewrite it");
    static_assert(MT == 1 || MT == 2 || MT == 4, "m-tiles per part");
    constexpr int G = 2 * KT, GP = mode_gp(MT, 2), NM = 6 * MT, PW = SplitPipeT<TILE>::PW;
    auto wcls = [](int t) { return t == 0 ? 0 : (t < 3 ? 1 : 2); };             // term -> class (0 = l, 1 = m, 2 = h) of the weights ...
    auto xcls = [](int t) { return t == 0 ? 2 : (t == 1 ? 1 : (t == 2 ? 2 : t - 3)); };   // ... and of the activations
    auto rows_in = [](int pi) { return (G - pi * GP) < GP ? (G - pi * GP) : GP; };
    auto ppk_of = [&](int pi) { return (PW + rows_in(pi) - 1) / rows_in(pi); };
    constexpr int ppk_full = (PW + GP - 1) / GP;      // pieces per row of a full panel
    constexpr int NUNITS = NSIDE > 0 ? 2 * PPG : 0;
    constexpr auto sched = make_row_sched<MT, STASH, NUNITS, 7>();      // (7: the instructions a side unit counts for)
    constexpr int NOPS = 12 + (STASH ? 2 : 0) + 1 + NUNITS;

    // the stash address as (wave-uniform base in scalar registers) + (32-bit lane offset): a 64-bit per-lane pointer kept across the
    // part is spilled at this register pressure, and its reload is a VMEM load that drains the store / DMA queues
    const char* stash_base = nullptr;
    int stash_off = 0;
    if constexpr (STASH) {
        const uint64_t p = reinterpret_cast<uint64_t>(stash);
        // (readfirstlane returns a SIGNED int: without the casts the low word is sign-extended over the high one)
        const uint64_t b = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(p >> 32)) << 32) |
                           (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)p);
        stash_base = reinterpret_cast<const char*>(b);
        stash_off = (int)(p - b);
    }
    // Entering the part's first panel.  Its pieces were issued while the previous part consumed its second-to-last panel; if that part
    // stashed, the stores of its last rows (pipe.part_pre of them, told by the kernel) are younger and may stay in flight -- without
    // this, every pass that follows a stashing pass starts with a full drain of the store queue (an HBM write latency, matrix pipe idle).
    if (pipe.part_pre == 6) pipe.template enter<6>(p0);
    else
        pipe.enter(p0);
    pipe.pieces(p0 + 2, 0, ppk_of(0));
    const unsigned lane_base = lds_byte_address(pipe.lds) + 16u * pipe.lane;
    unsigned panel_addr = lane_base + pipe.buffer(p0) * (SplitPipeT<TILE>::F4 * 16);
    f32x4 fr[3][MT];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) fr[c][mt] = frag_read(panel_addr, c * MT + mt);
    uint32_t xs[3][4], xn[3][4];   // [class][pair]: the packed B operands of the current / next row
    f32x2 rr[4];                   // the residuals of the next row's pairs between the split stages (adjacent registers: packed subtracts)
    const bool gates = STASH && pipe.gates_on;      // (set by the kernel right before the call: folded)
    if (gates) {
#pragma unroll
        for (int w = 0; w < 4; ++w) pipe.gw[w] = 0;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        split_pair(in[2 * q], in[2 * q + 1], xs[2][q], xs[1][q], xs[0][q]);
        if (gates) pipe.gw[0] |= gate_pair(xs[2][q]) << (2 * q);
    }

#pragma unroll
    for (int g = 0; g < G; ++g) {
        const bool last = g + 1 == G;
#pragma unroll
        for (int j = 0; j < NM; ++j) {
            const int t = j / MT, mt = j % MT, wc = wcls(t), xc = xcls(t);
            __builtin_amdgcn_sched_barrier(0);
            // first MFMA of a fragment class in this row: ALL its fragments have landed when at most the reads issued after the class's
            // last one are outstanding -- the later classes of the previous row's refills (order l, m, h) plus this row's refills so far
            // (none in the last row).  One wait per class, not per fragment: 3 instead of 3 MT (each drags an s_nop along).
            if (mt == 0 && t == 0) wait_class<MT>(fr[0], 2 * MT);
            else if (mt == 0 && t == 1) wait_class<MT>(fr[1], last ? MT : 2 * MT);
            else if (mt == 0 && t == 3) wait_class<MT>(fr[2], last ? 0 : 2 * MT);
            acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fr[wc][mt]),
                                                              __builtin_bit_cast(bf16x8, u32x4{xs[xc][0], xs[xc][1], xs[xc][2], xs[xc][3]}),
                                                              acc[mt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            // the fragment's last MFMA of this row: refill it in place for the next row.  A panel switch (counted wait + barrier) sits in
            // front of the first read from a new panel -- by then every read of the old panel has been issued (a row earlier).
            if (!last && (t == 0 || t == 2 || t == 5)) {
                const int pn = p0 + (g + 1) / GP;
                if (t == 0 && mt == 0 && (g + 1) % GP == 0) {
                    // Stash stores younger than the pieces waited for may stay in flight: those of this panel's earlier rows -- and, from
                    // the part's third panel on, those the panel BEFORE issued after the last burst of pn's pieces (pn's pieces go out
                    // while the panel two back is consumed: first burst in the row before it, then ppk per row).  Every store that must
                    // be waited for is then at least GP + 1 rows old instead of 1: the switch no longer waits out HBM write latency.
                    constexpr int kLastBurstRow = (PW + ppk_full - 1) / ppk_full - 2;
                    constexpr int kExtraNear = 2 * (GP - 1), kExtraFar = kExtraNear + 2 * (GP - 1 - kLastBurstRow);
                    if ((g + 1) / GP >= 2) pipe.template enter<STASH ? kExtraFar : 0>(pn);
                    else pipe.template enter<STASH ? kExtraNear : 0>(pn);
                    panel_addr = lane_base + pipe.buffer(pn) * (SplitPipeT<TILE>::F4 * 16);
                }
                fr[wc][mt] = frag_read(panel_addr, (((g + 1) % GP) * 3 + wc) * MT + mt);
            }
            // everything else, where the row's schedule puts it
#pragma unroll
            for (int i = 0; i < NOPS; ++i) {
                if (sched.gap[i] != j) continue;
                const int kind = sched.op[i].kind, k = sched.op[i].idx;
                if (kind == 0) {            // split of pair k of the next row, stage A: h, and what it leaves
                    if (!last) {
                        rr[k] = f32x2{in[8 * (g + 1) + 2 * k], in[8 * (g + 1) + 2 * k + 1]};
                        xn[2][k] = pack_pair(rr[k]);
                        rr[k] = pair_residual(rr[k], xn[2][k]);
                        if (gates) pipe.gw[(8 * (g + 1) + 2 * k) >> 5] |= gate_pair(xn[2][k]) << ((8 * (g + 1) + 2 * k) & 31);
                    }
                } else if (kind == 1) {     // stage B: m, and what it leaves
                    if (!last) {
                        xn[1][k] = pack_pair(rr[k]);
                        rr[k] = pair_residual(rr[k], xn[1][k]);
                    }
                } else if (kind == 2) {     // stage C: l
                    if (!last) xn[0][k] = pack_pair(rr[k]);
                } else if (kind == 3) {
                    if constexpr (STASH) {
#if !(NNR_ABLATE & 8)
                        const f32x4 val = f32x4{in[8 * g + 4 * k], in[8 * g + 4 * k + 1], in[8 * g + 4 * k + 2], in[8 * g + 4 * k + 3]};
                        if constexpr (TILE) {
                            // tile-major plane (nnr_layout.h): store 2 g + k of the part is octet 2 g + k of its input -- one contiguous 1 KiB block
                            // per wave, written past the L2 (whole lines: nothing for the L2 to merge, and the weight stream stays resident).
                            // Written as the instruction itself: (scalar base, bumped by 4 KiB every fourth block) + (32-bit lane offset) +
                            // immediate.  Through a pointer hipcc made it a FLAT store without the hint at first (the two arms of a run-time
                            // branch merged), then a global store whose 64-bit VECTOR address it re-based with two VALU adds per store.
                            // hipcc does not count asm memory operations: its own waits only get more conservative, and the counted
                            // vmcnt waits of the panel switches (PanelPipeT::enter<EXTRA>) are the ones written for these stores.
                            const uint64_t sb = reinterpret_cast<uint64_t>(stash_base) + 4096u * ((2 * g + k) >> 2);
                            asm volatile("global_store_dwordx4 %0, %1, %2 offset:%3 nt" : : "v"(stash_off), "v"(val), "s"(sb), "n"(1024 * ((2 * g + k) & 3)) : "memory");
                        } else {
                            f32x4* const dst = reinterpret_cast<f32x4*>(const_cast<char*>(stash_base) + stash_off + 32 * (2 * g + k));
                            *dst = val;
                        }
#endif
                    }
                } else if (kind == 4) {     // DMA pieces of the panel two ahead, spread over the rows of the current panel
                    const int pi = g / GP, gi = g % GP;
                    if (gi == rows_in(pi) - 1) {
                        if (!last) pipe.pieces(p0 + pi + 3, 0, ppk_of(pi + 1));   // this row entered panel pi + 1 above
                    } else {
                        pipe.pieces(p0 + pi + 2, (gi + 1) * ppk_of(pi), ppk_of(pi));
                    }
                } else {
                    if constexpr (NSIDE > 0) {
                        const int u = (2 * g - SHIFT) * PPG + k;
                        if (u >= 0 && u < NSIDE) side(u);
                    }
                }
            }
        }
        pin_acc<MT>(acc);
        __builtin_amdgcn_sched_barrier(0);
        if (!last) {
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int q = 0; q < 4; ++q) xs[c][q] = xn[c][q];
        }
    }
    if constexpr (NSIDE > 0) {
#pragma unroll
        for (int u = (2 * G - SHIFT) * PPG; u < NSIDE; ++u)
            if (u >= 0) side(u);
    }
}

template <int KT, int MT, bool STASH = false, int NACC, int NIN, bool TILE>
__device__ __forceinline__ void gemm_part(f32x16 (&acc)[NACC], const float (&in)[NIN], const SplitPipeT<TILE>& pipe, int p0,
                                          float* stash = nullptr) {
    gemm_part<KT, MT, STASH, 0, 1, 0>(acc, in, pipe, p0, stash, NoSide{});
}

}  // namespace nnr
