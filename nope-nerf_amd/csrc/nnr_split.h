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

using SplitPipe = PanelPipeT<kWavesPerBlock, kSplitPanelFrags>;

// the three bf16 terms of two fp32 values, packed (x0 in the low halves): h = rn(x), m = rn(x - h), l = rn(x - h - m); both differences
// are exact in fp32.  (An infinite or NaN input gives NaN terms -- as good as the inf the fp32 path would produce: the trainer stops.)
__device__ __forceinline__ void split_pair(float x0, float x1, uint32_t& h, uint32_t& m, uint32_t& l) {
    h = pack_bf16(x0, x1);
    const float r0 = x0 - __uint_as_float(h << 16), r1 = x1 - __uint_as_float(h & 0xffff0000u);
    m = pack_bf16(r0, r1);
    const float q0 = r0 - __uint_as_float(m << 16), q1 = r1 - __uint_as_float(m & 0xffff0000u);
    l = pack_bf16(q0, q1);
}

// experiments only (tools/gpu_r03_*.sh): NNR_SPLIT_SAFE_SYNC = every counted wait as a full one; NNR_SPLIT_TERMS = 1 / 3 / 6 of the terms
#ifndef NNR_SPLIT_TERMS
#define NNR_SPLIT_TERMS 6
#endif
// at most n LDS reads outstanding (n folds after unrolling; lgkmcnt is a 4-bit field); tied to the fragment like wait_frag
__device__ __forceinline__ void wait_lgkm_n(f32x4& frag, int n) {
#ifdef NNR_SPLIT_SAFE_SYNC
    n = 0;
#endif
    switch (n) {
#define NNR_WL(k) case k: asm volatile("s_waitcnt lgkmcnt(" #k ")" : "+v"(frag)); break;
        NNR_WL(0) NNR_WL(1) NNR_WL(2) NNR_WL(3) NNR_WL(4) NNR_WL(5) NNR_WL(6) NNR_WL(7) NNR_WL(8) NNR_WL(9) NNR_WL(10) NNR_WL(11)
        NNR_WL(12) NNR_WL(13) NNR_WL(14)
#undef NNR_WL
        default: asm volatile("s_waitcnt lgkmcnt(15)" : "+v"(frag)); break;
    }
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
// The other work of a row -- the split of the next row (4 pairs), two stash stores, the DMA pieces, 2 PPG side units -- is spread over
// the 3 MT gaps that hold no refill.
template <int KT, int MT, bool STASH, int NSIDE_, int PPG, int SHIFT, class Side, int NACC, int NIN>
__device__ __forceinline__ void gemm_part(f32x16 (&acc)[NACC], const float (&in)[NIN], const SplitPipe& pipe, int p0, float* stash,
                                          const Side& side) {
#ifdef NNR_ABLATE_NO_SIDE
    constexpr int NSIDE = 0;   // profiling build only
#else
    constexpr int NSIDE = NSIDE_;
#endif
    static_assert(MT <= NACC && 16 * KT <= NIN, "tile counts exceed the register arrays");
    static_assert(!(STASH && SHIFT != 0), "a part that stashes its input must not rewrite it");
    constexpr int G = 2 * KT, GP = mode_gp(MT, 2), NM = 6 * MT, PW = SplitPipe::PW;
    auto wcls = [](int t) { return t == 0 ? 0 : (t < 3 ? 1 : 2); };             // term -> class (0 = l, 1 = m, 2 = h) of the weights ...
    auto xcls = [](int t) { return t == 0 ? 2 : (t == 1 ? 1 : (t == 2 ? 2 : t - 3)); };   // ... and of the activations
    auto rows_in = [](int pi) { return (G - pi * GP) < GP ? (G - pi * GP) : GP; };
    auto ppk_of = [&](int pi) { return (PW + rows_in(pi) - 1) / rows_in(pi); };
    constexpr int NFREE = 3 * MT;                                              // gaps without a refill: terms 1, 3, 4
    constexpr int NWORK = 4 + (STASH ? 2 : 0) + 1 + (NSIDE > 0 ? 2 * PPG : 0);  // split pairs, stash stores, DMA, side units

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
    pipe.enter(p0);
    pipe.pieces(p0 + 2, 0, ppk_of(0));
    const unsigned lane_base = lds_byte_address(pipe.lds) + 16u * pipe.lane;
    unsigned panel_addr = lane_base + pipe.buffer(p0) * (SplitPipe::F4 * 16);
    f32x4 fr[3][MT];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) fr[c][mt] = frag_read(panel_addr, c * MT + mt);
    uint32_t xs[3][4], xn[3][4];   // [class][pair]: the packed B operands of the current / next row
#pragma unroll
    for (int q = 0; q < 4; ++q) split_pair(in[2 * q], in[2 * q + 1], xs[2][q], xs[1][q], xs[0][q]);

#pragma unroll
    for (int g = 0; g < G; ++g) {
        const bool last = g + 1 == G;
#pragma unroll
        for (int j = 0; j < NM; ++j) {
            const int t = j / MT, mt = j % MT, wc = wcls(t), xc = xcls(t);
            __builtin_amdgcn_sched_barrier(0);
            // first use of a fragment in this row: it has landed when at most the reads issued after it are outstanding -- the rest of
            // the previous row's refills (order l, m, h) plus this row's refills so far (none in the last row)
            if (t == 0) wait_lgkm_n(fr[0][mt], last ? 3 * MT - 1 - mt : 3 * MT - 1);
            else if (t == 1) wait_lgkm_n(fr[1][mt], (last ? 2 * MT : 3 * MT) - 1 - mt);
            else if (t == 3) wait_lgkm_n(fr[2][mt], (last ? MT : 3 * MT) - 1 - mt);
            if (NNR_SPLIT_TERMS == 6 || (NNR_SPLIT_TERMS == 3 && (t == 2 || t >= 4)) || (NNR_SPLIT_TERMS == 1 && t == 5))
            acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fr[wc][mt]),
                                                              __builtin_bit_cast(bf16x8, u32x4{xs[xc][0], xs[xc][1], xs[xc][2], xs[xc][3]}),
                                                              acc[mt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            // the fragment's last MFMA of this row: refill it in place for the next row.  A panel switch (counted wait + barrier) sits in
            // front of the first read from a new panel -- by then every read of the old panel has been issued (a row earlier).
            if (!last && (t == 0 || t == 2 || t == 5)) {
                const int pn = p0 + (g + 1) / GP;
                if (t == 0 && mt == 0 && (g + 1) % GP == 0) {
#ifdef NNR_SPLIT_SAFE_SYNC
                    pipe.enter<0>(pn);
#else
                    pipe.enter<STASH ? 2 * (GP - 1) : 0>(pn);
#endif   // this panel's earlier rows' stash stores are younger than the pieces waited for
                    panel_addr = lane_base + pipe.buffer(pn) * (SplitPipe::F4 * 16);
                }
                fr[wc][mt] = frag_read(panel_addr, (((g + 1) % GP) * 3 + wc) * MT + mt);
            }
            // everything else, in the gaps without a refill
            const int fj = t == 1 ? mt : (t == 3 ? MT + mt : (t == 4 ? 2 * MT + mt : -1));
            if (fj >= 0) {
#pragma unroll
                for (int i = 0; i < NWORK; ++i) {
                    if ((i * NFREE) / NWORK != fj) continue;
                    int k = i;
                    if (k < 4) {                       // split of pair k of the next row
                        if (!last) split_pair(in[8 * (g + 1) + 2 * k], in[8 * (g + 1) + 2 * k + 1], xn[2][k], xn[1][k], xn[0][k]);
                        continue;
                    }
                    k -= 4;
                    if (STASH) {
                        if (k < 2) {
#ifndef NNR_ABLATE_NO_STASH
                            *reinterpret_cast<f32x4*>(const_cast<char*>(stash_base) + stash_off + 32 * (2 * g + k)) =
                                f32x4{in[8 * g + 4 * k], in[8 * g + 4 * k + 1], in[8 * g + 4 * k + 2], in[8 * g + 4 * k + 3]};
#endif
                            continue;
                        }
                        k -= 2;
                    }
                    if (k == 0) {                      // DMA pieces of the panel two ahead, spread over the rows of the current panel
                        const int pi = g / GP, gi = g % GP;
                        if (gi == rows_in(pi) - 1) {
                            if (!last) pipe.pieces(p0 + pi + 3, 0, ppk_of(pi + 1));   // this row entered panel pi + 1 above
                        } else {
                            pipe.pieces(p0 + pi + 2, (gi + 1) * ppk_of(pi), ppk_of(pi));
                        }
                        continue;
                    }
                    k -= 1;
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

template <int KT, int MT, bool STASH = false, int NACC, int NIN>
__device__ __forceinline__ void gemm_part(f32x16 (&acc)[NACC], const float (&in)[NIN], const SplitPipe& pipe, int p0,
                                          float* stash = nullptr) {
    gemm_part<KT, MT, STASH, 0, 1, 0>(acc, in, pipe, p0, stash, NoSide{});
}

}  // namespace nnr
