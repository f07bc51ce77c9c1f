// nnr_mlp_dgrad.hip -- fused input-gradient chain of the NeRF MLP for gfx950.
// Replaces autograd's AddmmBackward/ReluBackward/SigmoidBackward/... chain through model/official_nerf.py:60-96 for
// the *data* path: from d(rgb_pre), d(sigma_raw) per sample down to d(point), d(view dir), leaving every layer's
// pre-activation gradient in the workspace for the weight-gradient kernel.  Same structure as the forward: one wave =
// 32 samples, gradients stay in VGPRs between layers as MFMA B operands, A fragments are the transposed packed weights.
// ReLU masks come from the 1-bit-per-activation stash written by the forward (32x less traffic than re-reading h).
#include "nnr_device.h"
#include "nnr_kernels.h"

namespace nnr {

template <int D>
__global__ __launch_bounds__(256, 1) void mlp_dgrad_kernel(MlpDgradArgs a) {
    using L = Layout<D>;
    constexpr int DT = L::DT, HT = L::HT;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int half = lane >> 5;
    const int col = lane & 31;
    const int64_t s = (int64_t)blockIdx.x * kBlockSamples + wave * kChunk + col;
    const bool live = s < a.S;  // padded samples carry zero gradients so they add nothing to the weight gradients

    // weight panels of the transposed (backward) stream through the LDS ring -- see PanelPipe in nnr_device.h
    __shared__ __attribute__((aligned(16))) f32x4 smem[kNBuf * kPanelF4 + kWavesPerBlock * 8 * 64];
    f32x4* const de_lds = smem + kNBuf * kPanelF4 + wave * (8 * 64) + lane;   // parking spot of d(posenc) from the skip layer
    const PanelPipe pipe{reinterpret_cast<const f32x4*>(a.packed + L::bwd_base) + wave * (8 * 64) + lane, smem, wave, lane,
                         L::bwd_panels};
    pipe.start();
    auto p0 = [&](int part) { return L::bwd_panel0(part); };
    const int64_t chunk = (int64_t)blockIdx.x * kWavesPerBlock + wave;
    const uint32_t* mask_base = a.ws_mask + ((chunk * L::n_mask_layers) * 64 + lane) * L::mask_words;

    f32x4 dout = {0.f, 0.f, 0.f, 0.f};
    if (live) dout = *reinterpret_cast<const f32x4*>(a.ws_dout4 + 4 * s);
    else if (half == 0) *reinterpret_cast<f32x4*>(a.ws_dout4 + 4 * s) = dout;   // padded rows feed the weight-gradient kernel: zero them

    // ---- colour branch ----
    // d rgb_pre (3, padded to one 32-row tile): rows 0..2 = registers 0..2 of half 0
    float drgb[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) drgb[r] = 0.f;
    drgb[0] = half == 0 ? dout[0] : 0.f;
    drgb[1] = half == 0 ? dout[1] : 0.f;
    drgb[2] = half == 0 ? dout[2] : 0.f;
    float dg[16 * HT];
    {
        f32x16 acc[HT];
        zero_acc(acc);
        gemm_part<1, HT>(acc, drgb, pipe, p0(B_RGB), gemm_open<HT>(pipe, p0(B_RGB)));
        uint32_t mw[L::mask_words];
        const uint32_t* m = mask_base + (int64_t)8 * 64 * L::mask_words;
#pragma unroll
        for (int w = 0; w < L::mask_words; ++w) mw[w] = m[w];
#pragma unroll
        for (int r = 0; r < 16 * HT; ++r) dg[r] = ((mw[r >> 5] >> (r & 31)) & 1u) ? acc[r >> 4][r & 15] : 0.f;
    }
    Frags<DT> frh = gemm_open<DT>(pipe, p0(B_RGBH_F));   // opened before the mask epilogue above retires
    float d[16 * DT];  // current D-wide gradient (d feature, then d pre-activation of hidden 8..1)
    {
        f32x16 acc[DT];
        zero_acc(acc);
        // every gradient vector is stashed by the gemm that consumes it (one 16-byte store per k-group, inside the
        // MFMA stream) -- see gemm_part.  Wg^T is consumed as two parts: rows of the feature (D) and of the direction
        // encoding (27 -> one 32-row tile).
        gemm_part<HT, DT, true>(acc, dg, pipe, p0(B_RGBH_F), frh, a.ws_dg + s * (D / 2) + 4 * half);
        f32x16 accd[1];
        zero_acc(accd);
        gemm_part<HT, 1>(accd, dg, pipe, p0(B_RGBH_D), gemm_open<1>(pipe, p0(B_RGBH_D)));
#pragma unroll
        for (int r = 0; r < 16 * DT; ++r) d[r] = acc[r >> 4][r & 15];
        // direction-encoding backward: d v = sum_f d gamma_4(v)_f/dv * grad_f, using the stored encoding for the
        // sin<->cos partner values (model/official_nerf.py:112-118)
        const float* enc = a.ws_xf + (live ? s : 0) * (D + kDirPad) + D;
        float gv[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int f = frag_feature(r, half);
            int c, partner;
            float sc;
            enc_feature_meta(f, kDirReal, c, sc, partner);
            const float pv = partner >= 0 ? enc[partner] : 1.f;
            const float contrib = accd[0][r] * sc * pv;
            gv[0] += c == 0 ? contrib : 0.f;
            gv[1] += c == 1 ? contrib : 0.f;
            gv[2] += c == 2 ? contrib : 0.f;
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) gv[c] += __shfl_xor(gv[c], 32, 64);
        if (half == 0 && live) *reinterpret_cast<f32x4*>(a.ws_dview + 4 * s) = f32x4{gv[0], gv[1], gv[2], 0.f};
    }

    // ---- trunk ----
    f32x16 acc[DT];
    auto masked_layer = [&](int hidden_idx /*0..7*/) {  // d <- acc .* relu'(h_idx)
        uint32_t mw[L::mask_words];
        const uint32_t* m = mask_base + (int64_t)hidden_idx * 64 * L::mask_words;
#pragma unroll
        for (int w = 0; w < L::mask_words; ++w) mw[w] = m[w];
#pragma unroll
        for (int r = 0; r < 16 * DT; ++r) d[r] = ((mw[r >> 5] >> (r & 31)) & 1u) ? acc[r >> 4][r & 15] : 0.f;
    };
    auto dh = [&](int hidden_idx /*0..7*/) -> float* { return a.ws_dh + ((int64_t)hidden_idx * a.S_pad + s) * D + 4 * half; };
    // d h8 = Wf^T d feat + w_sigma^T d sigma_raw
    float dsig[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) dsig[r] = 0.f;
    dsig[0] = half == 0 ? dout[3] : 0.f;
    zero_acc(acc);
    gemm_part<DT, DT, true>(acc, d, pipe, p0(B_FEAT), gemm_open<DT>(pipe, p0(B_FEAT)), a.ws_df + s * D + 4 * half);
    gemm_part<1, DT>(acc, dsig, pipe, p0(B_SIG), gemm_open<DT>(pipe, p0(B_SIG)));
    Frags<DT> fr = gemm_open<DT>(pipe, p0(B_L8));
    masked_layer(7);
    // hidden 8,7,6 -> d pre-activation of 7,6,5
#pragma unroll 1
    for (int l = 0; l < 3; ++l) {
        zero_acc(acc);
        gemm_part<DT, DT, true>(acc, d, pipe, p0(B_L8) + l * part_panels(DT, DT), fr, dh(7 - l));
        if (l < 2) fr = gemm_open<DT>(pipe, p0(B_L8) + (l + 1) * part_panels(DT, DT));
        masked_layer(6 - l);
    }
    // hidden 5 (skip layer), two parts of W5^T: rows [D, D+63) -> d posenc (kept for the end), rows [0,D) -> d h4
    {
        f32x16 acce[2];
        zero_acc(acce);
        gemm_part<DT, 2>(acce, d, pipe, p0(B_L5E), gemm_open<2>(pipe, p0(B_L5E)));
        // 32 registers that are not needed again until the very end: park them in LDS (same array as the panels)
#pragma unroll
        for (int q = 0; q < 8; ++q)
            de_lds[q * 64] = f32x4{acce[q >> 2][4 * (q & 3)], acce[q >> 2][4 * (q & 3) + 1], acce[q >> 2][4 * (q & 3) + 2],
                                   acce[q >> 2][4 * (q & 3) + 3]};
    }
    zero_acc(acc);
    gemm_part<DT, DT, true>(acc, d, pipe, p0(B_L5H), gemm_open<DT>(pipe, p0(B_L5H)), dh(4));
    fr = gemm_open<DT>(pipe, p0(B_L4));
    masked_layer(3);
    // hidden 4,3,2 -> d pre-activation of 3,2,1
#pragma unroll 1
    for (int l = 0; l < 3; ++l) {
        zero_acc(acc);
        gemm_part<DT, DT, true>(acc, d, pipe, p0(B_L4) + l * part_panels(DT, DT), fr, dh(3 - l));
        if (l < 2) fr = gemm_open<DT>(pipe, p0(B_L4) + (l + 1) * part_panels(DT, DT));
        masked_layer(2 - l);
    }
    // hidden 1: d posenc = W1^T d1 + (skip-layer part parked in LDS)
    float de[32];
    {
        f32x16 acc2[2];
        zero_acc(acc2);
        gemm_part<DT, 2, true>(acc2, d, pipe, p0(B_L1), gemm_open<2>(pipe, p0(B_L1)), dh(0));
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const f32x4 v = de_lds[q * 64];
#pragma unroll
            for (int i = 0; i < 4; ++i) de[4 * q + i] = acc2[q >> 2][4 * (q & 3) + i] + v[i];
        }
    }
    // positional-encoding backward -> d point
    {
        const float* enc = a.ws_xe + (live ? s : 0) * kPosPad;
        float gp[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 32; ++r) {
            const int f = frag_feature(r, half);
            int c, partner;
            float sc;
            enc_feature_meta(f, kPosReal, c, sc, partner);
            const float pv = partner >= 0 ? enc[partner] : 1.f;
            const float contrib = de[r] * sc * pv;
            gp[0] += c == 0 ? contrib : 0.f;
            gp[1] += c == 1 ? contrib : 0.f;
            gp[2] += c == 2 ? contrib : 0.f;
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) gp[c] += __shfl_xor(gp[c], 32, 64);
        if (half == 0 && live) *reinterpret_cast<f32x4*>(a.ws_dpts + 4 * s) = f32x4{gp[0], gp[1], gp[2], 0.f};
    }
}

template <int D>
static hipError_t launch(const MlpDgradArgs& a, hipStream_t st) {
    dim3 grid((unsigned)(a.S_pad / kBlockSamples)), block(256);
    hipLaunchKernelGGL((mlp_dgrad_kernel<D>), grid, block, 0, st, a);
    return hipGetLastError();
}

hipError_t launch_mlp_dgrad(int D, const MlpDgradArgs& a, hipStream_t st) {
    return D == 256 ? launch<256>(a, st) : launch<128>(a, st);
}

}  // namespace nnr
