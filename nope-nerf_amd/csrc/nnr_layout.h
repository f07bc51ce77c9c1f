// nnr_layout.h -- data layout shared by every kernel and by the host-side planners.
//
// Register ("fragment") layout of a feature vector, used for every activation / gradient that lives in
// VGPRs between MFMA layers.  A wave owns a chunk of 32 samples.  Lane l = (half h = l>>5, sample
// column c = l&31).  A vector of 32*T features of sample c is held in 16*T registers; register
// r = 16*t + rho (tile t, rho in [0,16)) holds feature
//        f(r,h) = 32*t + (rho&3) + 8*(rho>>2) + 4*h
// which is exactly the C/D layout of v_mfma_f32_32x32x2_f32 (row = (rho&3)+8*(rho>>2)+4*(lane>>5),
// col = lane&31) when the layer is computed transposed:  out^T[feature][sample] = W[feature][k] * in^T[k][sample].
// Because B[k][col] of that instruction is "lane l holds k = l>>5", register r of the *previous* layer's output is
// directly the B operand of two k-values (one per half) -- activations never leave registers between layers.
// Four consecutive registers r = 4g..4g+3 hold features 8g+4h+{0,1,2,3}: four consecutive k.
//
// Packed weights ("A fragments"): for a layer part with A[Mp=32*MT][Kp=32*KT] the fragment (g, mt), g in [0,4*KT),
// is 64 lanes x 4 floats = 1 KiB, lane l holding A[32*mt + (l&31)][8*g + 4*(l>>5) + i], i = 0..3: one 16-byte-per-lane
// read of a fragment feeds 4 MFMAs.  Fragments are grouped into PANELS of 32 KiB (32 fragment slots) in consumption
// order -- the unit the MLP kernels DMA into LDS (global_load_lds_dwordx4, lane-linear, so the LDS image equals the global
// image and a ds_read_b128 at +lane*16 is conflict-free).  A panel holds GP = 32/MT whole k-groups of one part (k-groups
// never straddle panels; slots beyond GP*MT are zero padding), fragment (g, mt) living in panel g/GP, slot (g%GP)*MT + mt.
#pragma once
#include <stdint.h>

#ifndef __HIPCC__
#define NNR_HD
#else
#define NNR_HD __host__ __device__
#endif

namespace nnr {

constexpr int kChunk = 32;         // samples per wave
constexpr int kWavesPerBlock = 4;  // one wave per SIMD
constexpr int kBlockSamples = kChunk * kWavesPerBlock;
// Shape of the bf16 MLP kernels (nnr_mlp_bf16.h): 32-sample chunks per wave x waves per workgroup = 256 samples.  2 x 4 is the
// product; the 1 x 8 shape (kBf16Tiles = 1: two waves per SIMD) measured slower, 0.82 / 0.78 ms against 0.77 / 0.68 ms (forward / input
// gradient, 4096 x 128) -- twice the LDS reads and DMA instructions per MFMA outweigh what the second wave hides of the epilogue.
constexpr int kBf16Tiles = 2, kBf16Waves = 8 / kBf16Tiles;
static_assert(kBf16Tiles == 1 || kBf16Tiles == 2, "one or two chunks per wave");
constexpr int kPosLevels = 10, kDirLevels = 4;  // hard-wired in the reference (model/official_nerf.py:61,87)
constexpr int kPosReal = 63, kDirReal = 27;     // (2L+1)*3
constexpr int kPosPad = 64, kDirPad = 32;

// ---- packed weight buffer -------------------------------------------------------------------------------------
// Forward parts in stream order, backward (transposed) parts in stream order, padded biases, head tables.
//
// Every D-wide layer is consumed as TWO passes over its input, one per half of the output tiles ("A" = output features
// [0,D/2), "B" = [D/2,D)): while pass B's MFMAs run, the wave finishes pass A's epilogue (bias is already in the
// accumulator; ReLU, sign bits, move to the activation registers), and pass B's epilogue runs under the first half of the
// NEXT layer's pass A -- whose first D/2 k-values only need the A half.  With one wave per SIMD this is the only way to
// keep the matrix pipe busy during epilogues.  The 1-row density head and the 3-row rgb head are not MFMA work at all
// (a 32-row tile would be 97 % padding): they are per-lane dot products against the head tables below.
//
// The feature layer (model/official_nerf.py:33,87: D -> D, NO activation) feeds only the colour-hidden layer, so the two
// are one linear map: g = relu(Wg [Wf h8 + bf ; dir] + bg) = relu((Wg[:, :D] Wf) h8 + Wg[:, D:] dir + (Wg[:, :D] bf + bg)).
// The pack kernel forms the MERGED matrix W' = Wg[:, :D] Wf (D/2 x D) and bias b' once per optimiser step; the MLP kernels
// never evaluate the feature layer (-10.8 % of the MACs of all three passes, -2 KB/sample of stash), and the weight-gradient
// kernel produces dW', db', from which dWf = Wg1^T dW', dWg1 = dW' Wf^T + db' bf^T, dbf = Wg1^T db', dbg = db' follow by
// two small products per step (exact chain rule; fp32 rounding differs from evaluating the layers one by one at the 1e-7 level).
constexpr int kMergedLayer = 12;   // pseudo-parameter index of W' / b' (after the 12 nn.Linear tensors)
enum FwdPart {
    F_L1A = 0, F_L1B, F_L2A, F_L2B, F_L3A, F_L3B, F_L4A, F_L4B, F_L5HA, F_L5EA, F_L5HB, F_L5EB,
    F_L6A, F_L6B, F_L7A, F_L7B, F_L8A, F_L8B, F_RGBH_F, F_RGBH_D, F_NPARTS
};
enum BwdPart {
    B_RGBH_FA = 0, B_RGBH_FB, B_RGBH_D, B_L8A, B_L8B, B_L7A, B_L7B, B_L6A, B_L6B,
    B_L5E, B_L5HA, B_L5HB, B_L4A, B_L4B, B_L3A, B_L3B, B_L2A, B_L2B, B_L1, B_NPARTS
};

struct PartDesc {
    int layer;      // index into nnr_params (state_dict order)
    int transpose;  // 0: A[m][k] = W[moff+m][koff+k]   1: A[m][k] = W[koff+k][moff+m]
    int KT, MT;     // tiles of 32 along k and m
    int m_real, k_real;  // valid rows / cols of A (rest is zero padding)
    int moff, koff; // offsets of the part inside W
    int ld;         // W row pitch (= in_features)
};

constexpr int kPanelFrags = 32;                 // fragment slots per panel
constexpr int kPanelFloats = kPanelFrags * 256;  // 32 KiB
NNR_HD constexpr int part_gp(int MT) { return kPanelFrags / MT; }  // fragment rows (k-groups) per panel
// fragment rows of a part: fp32 -- one per k-group of 8 (4*KT); bf16 -- one per DOUBLE k-group of 16 (2*KT), see below
NNR_HD constexpr int part_rows(int KT, bool bf16) { return bf16 ? 2 * KT : 4 * KT; }
NNR_HD constexpr int part_panels(int KT, int MT, bool bf16 = false) { return (part_rows(KT, bf16) + part_gp(MT) - 1) / part_gp(MT); }

// ---- MODE 2: fp32 operands as THREE bf16 terms (NNR_F_SPLIT3; the kernels: nnr_split.h) -------------------------------------------------
// CDNA4's matrix pipe multiplies bf16 sixteen times faster than fp32 (v_mfma_f32_32x32x16_bf16: 16 384 MACs in 32 cycles;
// v_mfma_f32_32x32x2_f32: 2 048 in 64).  An fp32 number is the exact sum of three bf16 numbers -- h = bf16(x), m = bf16(x - h),
// l = bf16(x - h - m), 8 + 8 + 8 significand bits, the same exponent range -- so a product w x = sum over the nine term pairs, of which
// the three smallest (w_m x_l, w_l x_m, w_l x_l: below 2^-24 |w x|, less than fp32's own rounding of the product) are dropped: SIX bf16
// MFMAs, accumulated in fp32, replace the eight fp32 MFMAs of the same 16 k-values -- 2.7 times fewer matrix-pipe cycles for a result
// that is as close to the exact product as the fp32 instruction's (measured against fp64: tests/test_gpu_split3.py).
// The packed weights of this mode: fragment rows as in the bf16 mode (one per 16 k-values), but each row holds 3 * MT fragments -- the
// l, m and h terms of the row, in that order (the order the kernels consume them in) -- and a panel is 24 fragment slots (24 KiB):
// GP = 8 / MT whole rows.  Fragment (b, term t, mt) lives in panel b / GP, slot ((b % GP) * 3 + t) * MT + mt.
constexpr int kSplitPanelFrags = 24;
// MODE 3 (NNR_F_SPLIT2; the kernels: nnr_split2.h): fp32 operands as TWO fp16 terms, three fp16 MFMAs per 16 k-values.  A fragment row holds
// TWO classes per m-tile -- class 0 = w_m = fp16(w s - w_h), class 1 = w_h = fp16(w s) (the third operand, w_h 2^-11 for the activations' residual
// term that is carried at 2^11, is made from w_h in registers: an exact exponent shift) -- in 32-slot panels: GP = 16 / MT whole rows, fragment
// (b, class c, mt) in panel b / GP, slot ((b % GP) * 2 + c) * MT + mt.  s = the power of two that puts the largest |w| of the weight tensor's
// SCALE SLOT into [2^13, 2^14) (scale_slot below; the table: Layout::scale_off).  The biases of the MFMA layers are stored multiplied by s (the
// accumulators start there); head tables and merge area are unscaled fp32.
NNR_HD constexpr int mode_panel_frags(int mode) { return mode == 2 ? kSplitPanelFrags : kPanelFrags; }
NNR_HD constexpr int mode_rows(int KT, int mode) { return mode ? 2 * KT : 4 * KT; }
NNR_HD constexpr int mode_gp(int MT, int mode) { return mode == 2 ? kSplitPanelFrags / (3 * MT) : (mode == 3 ? kPanelFrags / (2 * MT) : kPanelFrags / MT); }
// scale slot of a parameter index: hidden 1..8 their own; the merged colour matrix W' (kMergedLayer) and the direction columns of the colour-hidden
// layer (param 10) accumulate into ONE accumulator and share slot 8
NNR_HD constexpr int scale_slot(int layer) { return layer < 8 ? layer : 8; }
constexpr int kScaleSlots = 9;
NNR_HD constexpr int mode_panels(int KT, int MT, int mode) { return (mode_rows(KT, mode) + mode_gp(MT, mode) - 1) / mode_gp(MT, mode); }

// MODE = 1 (BF16): the packed weights of the bf16-MFMA mode (NNR_F_BF16).  A fragment is still 64 lanes x 16 bytes, but holds
// 8 bf16 per lane: lane l of fragment (b, mt) has A[32*mt + (l&31)][16b + 4h + i] (i = 0..3), then [16b + 8 + 4h + i] with
// h = l>>5 -- the two k-groups 2b, 2b+1 of the fp32 layout, which is exactly the order in which 8 consecutive activation
// registers of a lane hold them, so one v_mfma_f32_32x32x16_bf16 consumes 8 registers (packed to bf16) against one fragment
// (the k labelling inside an MFMA is arbitrary as long as A and B agree).  Biases, head tables and the merge area stay fp32.
template <int D, int MODE = 0>   // 0: fp32 MFMA, 1: bf16 MFMA, 2: fp32 as three bf16 terms, 3: fp32 as two fp16 terms
struct Layout {
    static constexpr int panel_floats = mode_panel_frags(MODE) * 256;
    static constexpr int DT = D / 32;
    static constexpr int HT = D / 64;  // tiles of half a layer's outputs == tiles of the colour-hidden layer (D/2 wide)
    static_assert(D == 128 || D == 256, "hidden width must be 128 or 256");
    static constexpr int P = kPosReal, Q = kDirReal, Dh = D / 2;

    NNR_HD static constexpr PartDesc fwd(int p) {
        if (p < F_L2A) return {0, 0, 2, HT, Dh, P, (p & 1) * Dh, 0, P};
        if (p < F_L5HA) return {1 + (p - F_L2A) / 2, 0, DT, HT, Dh, D, (p & 1) * Dh, 0, D};
        if (p < F_L6A) {
            const int q = p - F_L5HA, half = q >> 1;
            if (q & 1) return {4, 0, 2, HT, Dh, P, half * Dh, D, D + P};
            return {4, 0, DT, HT, Dh, D, half * Dh, 0, D + P};
        }
        if (p < F_RGBH_F) return {5 + (p - F_L6A) / 2, 0, DT, HT, Dh, D, (p & 1) * Dh, 0, D};
        if (p == F_RGBH_F) return {kMergedLayer, 0, DT, HT, Dh, D, 0, 0, D};   // W' (D/2 x D), input h8
        return {10, 0, 1, HT, Dh, Q, 0, D, D + Q};
    }
    NNR_HD static constexpr PartDesc bwd(int p) {
        if (p < B_RGBH_D) return {kMergedLayer, 1, HT, HT, Dh, Dh, p * Dh, 0, D};   // W'^T: d h8 halves from d g
        if (p == B_RGBH_D) return {10, 1, HT, 1, Q, Dh, D, 0, D + Q};
        if (p < B_L5E) return {7 - (p - B_L8A) / 2, 1, DT, HT, Dh, D, ((p - B_L8A) & 1) * Dh, 0, D};
        if (p == B_L5E) return {4, 1, DT, 2, P, D, D, 0, D + P};
        if (p < B_L4A) return {4, 1, DT, HT, Dh, D, (p - B_L5HA) * Dh, 0, D + P};
        if (p < B_L1) return {3 - (p - B_L4A) / 2, 1, DT, HT, Dh, D, ((p - B_L4A) & 1) * Dh, 0, D};
        return {0, 1, DT, 2, P, D, 0, 0, P};
    }
    // first panel of a part; the forward stream occupies panels [0, fwd_panels), the backward stream follows
    NNR_HD static constexpr int fwd_panel0(int p) {
        int o = 0;
        for (int i = 0; i < p; ++i) o += mode_panels(fwd(i).KT, fwd(i).MT, MODE);
        return o;
    }
    static constexpr int fwd_panels = fwd_panel0(F_NPARTS);
    NNR_HD static constexpr int bwd_panel0(int p) {  // relative to the start of the backward stream
        int o = 0;
        for (int i = 0; i < p; ++i) o += mode_panels(bwd(i).KT, bwd(i).MT, MODE);
        return o;
    }
    static constexpr int bwd_panels = bwd_panel0(B_NPARTS);
    static constexpr int bwd_base = fwd_panels * panel_floats;  // float offset of the backward stream
    // biases, each padded to a multiple of 32 floats: hidden 1..8, sigma, feature, colour hidden, rgb
    static constexpr int bias_base = (fwd_panels + bwd_panels) * panel_floats;
    NNR_HD static constexpr int bias_off(int layer) {  // layer in state_dict order
        int o = bias_base;
        for (int i = 0; i < layer; ++i) o += bias_pad(i);
        return o;
    }
    NNR_HD static constexpr int bias_pad(int layer) { return layer == 8 || layer == 11 ? 32 : (layer == 10 ? (D / 2 + 31) / 32 * 32 : D); }
    NNR_HD static constexpr int bias_real(int layer) { return layer == 8 ? 1 : layer == 11 ? 3 : layer == 10 ? D / 2 : D; }
    // head tables in REGISTER order (index [half][r], feature = frag_feature(r, half)): density row, then the 3 rgb rows
    static constexpr int bias_floats = bias_off(12) - bias_base;
    static constexpr int head_base = bias_off(12);
    static constexpr int wsig_off = head_base;                  // [2][16*DT]
    static constexpr int wrgb_off = wsig_off + 2 * 16 * DT;     // [3][2][16*HT]
    static constexpr int head_floats = 2 * 16 * DT + 3 * 2 * 16 * HT;
    // MODE 3: [16] the weight scales s by scale slot, [16] their inverses (powers of two; written by scale_kernel, nnr_pack.hip)
    static constexpr int scale_off = head_base + head_floats;
    static constexpr int scale_floats = MODE == 3 ? 32 : 0;
    static constexpr int table_floats = bias_floats + head_floats + scale_floats;  // what the MLP kernels copy into LDS
    // merge area (row-major): W' [Dh][D], b' [Dh], then copies of Wf [D][D], Wg[:, :D] [Dh][D] and bf [D] for the
    // un-merge step of the weight-gradient pass.  The bias slot of layer 10 above holds b' (not bg).
    static constexpr int merged_w_off = scale_off + scale_floats;
    static constexpr int merged_b_off = merged_w_off + Dh * D;
    static constexpr int copy_wf_off = merged_b_off + Dh;
    static constexpr int copy_wg_off = copy_wf_off + D * D;
    static constexpr int copy_bf_off = copy_wg_off + Dh * D;
    static constexpr int packed_floats = copy_bf_off + D;

    // ---- workspace planes (floats), S_pad = samples rounded up to a multiple of kBlockSamples ----
    static constexpr int x_width = kPosPad + 8 * D + kDirPad + D / 2;  // per-sample activation stash
    static constexpr int d_width = 8 * D + D / 2;                      // per-sample gradient stash
    static constexpr int mask_words = DT / 2;                                // uint32 per lane per masked layer
    static constexpr int n_mask_layers = 9;                                  // hidden 1..8 + colour hidden
};

// workspace plane ids (nnr_ws_plane)
enum Plane {
    P_OUT4 = 0, P_Z = 1, P_DOUT4 = 2, P_DPTS = 3, P_DVIEW = 4,
    P_XE = 10, P_XH1 = 11, /* .. P_XH8 = 18 */ P_XF = 19 /* direction encoding only */, P_XG = 20,
    P_XE16 = 21, P_XF16 = 22,   /* bf16 training only: tile-major bf16 copies of the two encodings for the weight-gradient kernel */
    P_MASK = 25,
    P_DH1 = 31, /* .. P_DH8 = 38 */ P_DG = 40,
};

// ---- tile-major bf16 planes (bf16 training mode) ------------------------------------------------------------------
// The operands of the weight-gradient products -- hidden activations P_XH1.., P_XG, the encodings' copies P_XE16 / P_XF16 and the
// pre-activation gradients P_DH1.., P_DG -- are stored as the very bf16 values the forward / input-gradient MFMAs consumed, in
// the order the producing wave holds them: a plane of G groups of 16 features is an array of 1 KiB BLOCKS [chunk][group], chunk =
// 32 consecutive samples (one wave of the MLP kernels), and a block is [lane = 32 h + c][8 bf16] with lane (h, c) holding, for
// sample 32 chunk + c, features 16 g + 4 h + {0..3} and 16 g + 8 + 4 h + {0..3} -- its packed MFMA B operand of that row-step.
// Element (sample s, feature f) sits at bf16 index
//     (((s / 32) * G + f / 16) * 64 + 32 * ((f % 8) / 4) + s % 32) * 8 + 4 * ((f % 16) / 8) + f % 4.
// Every stash store of the producers is then ONE fully coalesced 1 KiB wave-store (8 whole cache lines; the row-major planes of
// round 1 wrote 16 bytes into each of 32 lines per store), and the weight-gradient kernel streams whole blocks into LDS with
// global_load_lds_dwordx4 -- the transposition [sample][feature] -> [feature][8 samples] its MFMA operands need happens in the LDS
// read (ds_read_b64_tr_b16), not in HBM.
// P_DG carries one extra group in this mode: group D/32 = the per-sample output gradients as bf16 (d rgb_pre[0..2], d sigma_raw,
// 12 zeros), so that the two head layers are ordinary tiles of the same kernel.
// P_XE / P_XF stay fp32 in this mode but change meaning: they hold the chain-rule FACTORS of the encodings (scale * partner value) in
// fragment-register order, tile-major -- block (chunk, q) = [lane][factors of registers 4q .. 4q+3] (enc_factor, nnr_mlp_bf16.h).
NNR_HD constexpr int64_t tile_major_index(int64_t s, int f, int G) {
    return ((((s >> 5) * G + (f >> 4)) * 64 + 32 * ((f & 7) >> 2) + (s & 31)) << 3) + 4 * ((f & 15) >> 3) + (f & 3);
}

// ---- tile-major fp32 planes (three-term training mode, round 4) ---------------------------------------------------------------------
// The pre-activation gradients P_DH1.., P_DG of the NNR_F_SPLIT3 training workspace are stored in the order the producing wave holds
// them, like the bf16 planes above but 4 bytes per value: a plane of W floats per sample is an array of 1 KiB BLOCKS [chunk of 32
// samples][octet j of features], a block is [lane = 32 h + c][4 floats] with lane (h, c) holding, for sample 32 chunk + c, the features
// 8 j + 4 h + {0..3} -- the four registers one stash store of gemm_part (nnr_split.h) writes.  Every stash store of the input-gradient
// kernel is then ONE contiguous, non-temporal 1 KiB wave-store (8 whole cache lines that go past the L2 instead of evicting the weight
// stream from it) where the row-major plane took 16 bytes into each of 64 lines: 1.07 -> 0.88 ms for that kernel
// (profiles/r04/a_stash_variants.txt).  The weight-gradient kernel fetches 64-byte runs of such blocks by LDS-DMA (nnr_wgrad.hip).
constexpr bool kTileGradPlanes = true;
// The ACTIVATION planes (P_XE, P_XH1.., P_XF, P_XG) of the three-term training workspace are tile-major too (same blocks, written by the
// forward's stash stores).  The first measurement said the forward does not gain from it (1.185 against 1.108 ms): that experiment
// library had compiled its forward with 183 scratch reloads, every one a full drain of the store queue (an extra address register per
// store) -- found only at the end of round 4, when a build WITHOUT the weight DMA ran the training forward at the inference forward's
// time (0.85 ms): what the row-major stores cost is the 64 cache lines each of them touches in the address path the weight DMA shares.
constexpr bool kTileActPlanes = true;
NNR_HD constexpr int64_t tile32_index(int64_t s, int f, int W) {
    return (((s >> 5) * (W >> 3) + (f >> 3)) << 8) + ((((f >> 2) & 1) * 32 + (s & 31)) << 2) + (f & 3);
}

struct WsLayout {
    int64_t S, S_pad;
    int D;
    bool train;
    bool tile32 = false; // NNR_F_SPLIT3 training: the stash planes (activations and gradients) are tile-major fp32 (tiled(), above); offsets and sizes are unchanged
    NNR_HD bool tiled(int p) const {
        if (!(tile32 && train)) return false;
        if ((p >= P_DH1 && p < P_DH1 + 8) || p == P_DG) return kTileGradPlanes;
        if (p == P_XE || (p >= P_XH1 && p < P_XH1 + 8) || p == P_XF || p == P_XG) return kTileActPlanes;
        return false;
    }
    bool bf16 = false;   // NNR_F_BF16 training: the operands of the weight-gradient products -- hidden activations (P_XH1.., P_XG),
                         // the encodings' copies (P_XE16, P_XF16) and the pre-activation gradients (P_DH1.., P_DG) -- are tile-major
                         // bf16 planes (see above; pitch below = floats per sample = elements / 2); P_XE / P_XF hold the fp32
                         // chain-rule factors in register order; masks and the 4-wide planes are as in the fp32 mode
    NNR_HD int64_t plane(int p, int* pitch) const {
        int64_t o = 0;
        int w = 0;
        const int wD = bf16 ? D / 2 : D, wDh = bf16 ? D / 4 : D / 2;
        const int wDg = bf16 ? D / 4 + 8 : D / 2;   // bf16: D/32 groups of colour-hidden gradients + 1 group of output gradients
        auto step = [&](int id, int width) -> bool {
            if (id == p) { w = width; return true; }
            o += S_pad * (int64_t)width;
            return false;
        };
        if (step(P_OUT4, 4) || step(P_Z, 1)) { *pitch = w; return o; }
        if (!train) return -1;
        if (step(P_DOUT4, 4) || step(P_DPTS, 4) || step(P_DVIEW, 4) || step(P_XE, kPosPad)) { *pitch = w; return o; }
        for (int l = 0; l < 8; ++l)
            if (step(P_XH1 + l, wD)) { *pitch = w; return o; }
        if (step(P_XF, kDirPad) || step(P_XG, wDh)) { *pitch = w; return o; }
        if (bf16 && (step(P_XE16, kPosPad / 2) || step(P_XF16, kDirPad / 2))) { *pitch = w; return o; }
        // masks: [S_pad/32 chunks][9 layers][64 lanes][D/64 words]  == S_pad * 9 * 2 * (D/64) / ... words per sample: 9*2*(D/64)
        if (step(P_MASK, 9 * 2 * (D / 64))) { *pitch = w; return o; }
        for (int l = 0; l < 8; ++l)
            if (step(P_DH1 + l, wD)) { *pitch = w; return o; }
        if (step(P_DG, wDg)) { *pitch = w; return o; }
        if (p == -1) { *pitch = 0; return o; }  // total
        return -1;
    }
    NNR_HD int64_t total() const {
        int pitch;
        if (!train) {
            return S_pad * 5;
        }
        return plane(-1, &pitch);
    }
};

// ---- weight-gradient plan: one entry per wave job ---------------------------------------------------------------
// dW[layer][row0 + MI*m + i][wcol0 + NI*n + j] += sum_{s in [k0,k1)} Dlt[s][dcol0 + MI*m + i] * X[s][xcol0 + NI*n + j]
// (m, n in [0,32), i < MI, j < NI): a wave tile of 32*MI rows x 32*NI cols with interleaved sub-tiles, so one
// MI-wide and one NI-wide vector load per lane feed MI*NI MFMAs.  The sample axis is split over several jobs per tile;
// each job writes its partial tile to its own slot of the workspace (plain, coalesced stores) and a second small kernel
// adds the slots of a tile into dW -- ~27 float atomics per weight at the end of every wave cost 7 % of the kernel.
struct WgradJob {
    int32_t layer;           // parameter index (state_dict order); kMergedLayer = the merged feature/colour matrix W'
    int32_t MI, NI;          // 1, 2 or 4
    int32_t d_plane, d_col0, d_valid;  // gradient operand: workspace plane id, first column, valid columns from d_col0
    int32_t x_plane, x_col0, x_valid;  // activation operand
    int32_t row0, wcol0;     // destination offsets in W (rows = out features, cols = in features)
    int32_t rows_real, cols_real, ldw;  // bounds and pitch of W
    int32_t k0, k1;          // sample range (multiples of 16)
    int32_t bias;            // this tile also reduces d(bias)[row] = sum_s Dlt[s][row]: 1 = every sample pair, 2 / 3 = the
                             // even / odd pairs (two tiles of a row block share the work), 0 = not at all
    int32_t split, next_split;  // this job is split `split` (in sample order) of its tile; job index of the next one, -1 = last
    int32_t reserved;
};
// ---- weight-gradient plan of the bf16 mode: one entry per WORKGROUP job (nnr_wgrad_bf16.hip) --------------------------------
// A job streams the 32-sample chunks [c0, c1) of two or three tile-major planes -- d_groups consecutive blocks per chunk of the
// gradient operand, x_groups of the activation operand, then x2_groups of a second activation plane that continues the first one's
// feature axis (the skip layer's input = hidden | position encoding: one pass over the gradient for both) -- through LDS; wave w < WR * WC owns the MT x NT MFMA tiles (32 x 32)
// at tile row MT * (w / WC), tile column NT * (w % WC) of  dW_unit[d feature][x feature] = sum_s Dlt[s][d] X[s][x], keeps them in
// accumulators for the whole range and writes them to its slot (4 * job + w); waves of tile column 0 also sum the gradient
// operand over the samples (d bias).  Feature i of tile t is feature 32 t + i of the staged groups, in natural order.
struct WgradJobB {
    int64_t d_base, x_base;             // byte offset in the workspace of block (chunk 0, first staged group) of either operand
    int32_t d_stride, x_stride;         // bytes per chunk of either plane (1 KiB x its groups)
    int32_t d_groups, x_groups;         // blocks staged per chunk
    int32_t unit;                       // index into the unit table (which tile of which dW this is: BUnit in nnr_api.cpp)
    int32_t MT, NT, WR, WC;
    int32_t c0, c1;                     // chunk range
    int32_t bias;                       // 1: tile-column-0 waves reduce d(bias)
    int32_t split, next_split;          // position in the chain of the unit's jobs (sample order), next job or -1
    int64_t x2_base;                    // second activation plane (x2_groups = 0: none); x_groups is even when it is used
    int32_t x2_stride, x2_groups;
};
// one destination rectangle of a unit: rows [d_row, d_row + n_rows) x columns [x_col, x_col + n_cols) of the unit's product (feature
// offsets relative to the staged groups) go to W[layer][w_row + ..][w_col + ..]; bias rows likewise when `bias`
struct WgradOutB {
    int32_t unit, layer;
    int32_t d_row, n_rows, w_row;
    int32_t x_col, n_cols, w_col, ldw;
    int32_t bias;
    int32_t first_job;                  // head of the unit's job chain
    int32_t MT, NT, WR, WC;             // copy of the unit's tiling (to find an element's wave slot)
    int32_t reserved;
};
constexpr int kSlotBTile = 32 * 32;     // floats per MFMA tile in a slot, row-major
// slot of (job, wave): [MT * NT tiles][32][32] floats, then [MT][2 k-step halves][32] bias partials; fixed pitch for every shape
constexpr int kSlotBMaxTiles = 20;      // the widest wave tiling: 4 x 5
constexpr int kSlotBFloats = kSlotBMaxTiles * kSlotBTile + 5 * 2 * 32;

// partial slot of job i: floats [i*kSlotFloats, (i+1)*kSlotFloats) of the slot region = tile (32*MI rows x 32*NI cols,
// row-major, pitch 32*NI) followed by the two half-wave bias partials [2][32*MI]
constexpr int kSlotTile = 128 * 128;
constexpr int kSlotFloats = kSlotTile + 2 * 128;

}  // namespace nnr
