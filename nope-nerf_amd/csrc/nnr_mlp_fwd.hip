// nnr_mlp_fwd.hip -- fused NeRF MLP forward for gfx950: sampling + positional encoding + 12 layers, per-sample
// (rgb, sigma_raw) out.  Restates, per sample: model/rendering.py:184-195 (z, points, view dir) and
// model/official_nerf.py:60-96 (the MLP).  One wave = 32 samples; activations stay in VGPRs between layers as MFMA
// B-operands (see nnr_layout.h); weights arrive as pre-packed A fragments through a DMA-fed LDS ring shared by the 4 waves.
//
// Roofline: MFMA-bound.  593 408 MACs/sample at D=256 -> 9 472 v_mfma_f32_32x32x2_f32 per 32 samples (9 272 useful,
// 2 % padding of the 63/27/1/3-wide edges) = 606 k cycles per wave against ~10 k cycles of everything else.
// HBM per sample: 4 B jitter in, 20 B out (+ the 10 KB activation stash when training, written once, never re-read here).
#include "nnr_device.h"
#include "nnr_kernels.h"

namespace nnr {

#ifdef NNR_ABLATE_NO_MASK
constexpr bool kAblateNoMask = true;   // profiling build only
#else
constexpr bool kAblateNoMask = false;
#endif

template <int D, bool TRAIN>
__global__ __launch_bounds__(256, 1) void mlp_fwd_kernel(MlpFwdArgs a) {
    using L = Layout<D>;
    constexpr int DT = L::DT, HT = L::HT;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int half = lane >> 5;
    const int col = lane & 31;
    const int64_t s = (int64_t)blockIdx.x * kBlockSamples + wave * kChunk + col;  // this lane's sample
    const int64_t sc = s < a.S ? s : a.S - 1;                                      // clamp: padded samples recompute the last one
    const int ray = (int)(sc / a.N);
    const int j = (int)(sc - (int64_t)ray * a.N);

    // ---- sampling (model/rendering.py:184-195).  Unfused mul/add to round exactly like the reference. ----
    float zlo = a.z_lo[j], zhi = a.z_hi[j];
    float z = zlo;
    if (a.jitter) z = __fadd_rn(zlo, __fmul_rn(__fsub_rn(zhi, zlo), a.jitter[sc]));
    const float* ro = a.pts_o + 3 * (int64_t)ray;
    const float* rd = a.pts_d + 3 * (int64_t)ray;
    const float* rv = a.view_d + 3 * (int64_t)ray;
    const float px = __fadd_rn(ro[0], __fmul_rn(rd[0], z));
    const float py = __fadd_rn(ro[1], __fmul_rn(rd[1], z));
    const float pz = __fadd_rn(ro[2], __fmul_rn(rd[2], z));
    const float vx = rv[0], vy = rv[1], vz = rv[2];
    if (half == 0 && s < a.S) a.ws_z[s] = z;

    // ---- encodings, straight into fragment layout ----
    float e[32];  // gamma_10(p): 63 -> 64
#pragma unroll
    for (int r = 0; r < 32; ++r) e[r] = enc_feature(frag_feature(r, half), kPosReal, px, py, pz);
    float dirv[16];  // gamma_4(v): 27 -> 32
#pragma unroll
    for (int r = 0; r < 16; ++r) dirv[r] = enc_feature(frag_feature(r, half), kDirReal, vx, vy, vz);

    // ---- weight panels (LDS ring, DMA two panels ahead) and the bias table, all in ONE __shared__ array ----
    __shared__ __attribute__((aligned(16))) f32x4 smem[kNBuf * kPanelF4 + (L::bias_floats + 3) / 4];
    float* const lbias = reinterpret_cast<float*>(smem + kNBuf * kPanelF4);
    for (int i = threadIdx.x; i < L::bias_floats; i += 256) lbias[i] = a.packed[L::bias_base + i];
    __syncthreads();   // before any DMA is in flight: this is the only full barrier of the kernel
    const PanelPipe pipe{reinterpret_cast<const f32x4*>(a.packed) + wave * (8 * 64) + lane, smem, wave, lane, L::fwd_panels};
    pipe.start();
    auto p0 = [&](int part) { return L::fwd_panel0(part); };
    const float* bias = lbias - L::bias_base;   // index with L::bias_off(layer)

    uint32_t* mask_base = nullptr;
    if (TRAIN) {
        // masks: [chunk][layer][lane][mask_words]
        int64_t chunk = (int64_t)blockIdx.x * kWavesPerBlock + wave;
        mask_base = a.ws_mask + ((chunk * L::n_mask_layers) * 64 + lane) * L::mask_words;
    }

    float h[16 * DT];
    f32x16 acc[DT];

    // epilogue of a D-wide ReLU layer: h = relu(acc + b); keep the sign bits (the activations themselves are stashed
    // by the *next* layer's gemm, interleaved with its MFMAs)
    auto relu_layer = [&](int layer_idx /*0..7*/) {
        const float* b = bias + L::bias_off(layer_idx) + 4 * half;
        uint32_t mw[L::mask_words];
#pragma unroll
        for (int w = 0; w < L::mask_words; ++w) mw[w] = 0;
#pragma unroll
        for (int t = 0; t < DT; ++t) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 bb = *reinterpret_cast<const f32x4*>(b + 32 * t + 8 * q);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float x = acc[t][4 * q + i] + bb[i];
                    x = fmaxf(x, 0.f);
                    const int r = 16 * t + 4 * q + i;
                    h[r] = x;
                    if (TRAIN && !kAblateNoMask) mw[r >> 5] |= min(__float_as_uint(x), 1u) << (r & 31);   // x >= 0: bit = (x != 0), no VCC round trip
                }
            }
        }
        if (TRAIN) {
            uint32_t* m = mask_base + (int64_t)layer_idx * 64 * L::mask_words;
#pragma unroll
            for (int w = 0; w < L::mask_words; ++w) m[w] = mw[w];
        }
    };

    // hidden 1: 63 -> D
    zero_acc(acc);
    float* const xe = TRAIN ? a.ws_xe + s * kPosPad + 4 * half : nullptr;
    auto xh = [&](int hidden_idx /*0..7*/) -> float* {
        return TRAIN ? a.ws_xh + ((int64_t)hidden_idx * a.S_pad + s) * D + 4 * half : nullptr;
    };
    // Every part is opened (panel switch + first fragment reads) before the epilogue that precedes it.
    gemm_part<2, DT, TRAIN>(acc, e, pipe, p0(F_L1), gemm_open<DT>(pipe, p0(F_L1)), xe);
    Frags<DT> fr = gemm_open<DT>(pipe, p0(F_L2));
    relu_layer(0);
    // hidden 2..4
#pragma unroll 1
    for (int l = 0; l < 3; ++l) {
        zero_acc(acc);
        gemm_part<DT, DT, TRAIN>(acc, h, pipe, p0(F_L2) + l * part_panels(DT, DT), fr, xh(l));
        fr = gemm_open<DT>(pipe, p0(F_L2) + (l + 1) * part_panels(DT, DT));   // l == 2: that is the first panel of hidden 5
        relu_layer(1 + l);
    }
    // hidden 5: [h4 ; e] -> D   (skip connection, input order [h, posenc]: model/official_nerf.py:63)
    zero_acc(acc);
    static_assert(L::fwd_panel0(F_L5H) == L::fwd_panel0(F_L2) + 3 * part_panels(DT, DT), "stream order");
    gemm_part<DT, DT, TRAIN>(acc, h, pipe, p0(F_L5H), fr, xh(3));
    gemm_part<2, DT>(acc, e, pipe, p0(F_L5E), gemm_open<DT>(pipe, p0(F_L5E)));
    fr = gemm_open<DT>(pipe, p0(F_L6));
    relu_layer(4);
    // hidden 6..8
#pragma unroll 1
    for (int l = 0; l < 3; ++l) {
        zero_acc(acc);
        gemm_part<DT, DT, TRAIN>(acc, h, pipe, p0(F_L6) + l * part_panels(DT, DT), fr, xh(4 + l));
        if (l < 2) fr = gemm_open<DT>(pipe, p0(F_L6) + (l + 1) * part_panels(DT, DT));
        relu_layer(5 + l);
    }
    // density head: D -> 1 (row 0 of a 32-row tile)
    f32x16 acc1[1];
    zero_acc(acc1);
    gemm_part<DT, 1>(acc1, h, pipe, p0(F_SIG), gemm_open<1>(pipe, p0(F_SIG)));
    const float sigma_raw = acc1[0][0] + bias[L::bias_off(8)];
    // feature: D -> D, no activation
    zero_acc(acc);
    gemm_part<DT, DT, TRAIN>(acc, h, pipe, p0(F_FEAT), gemm_open<DT>(pipe, p0(F_FEAT)), xh(7));
    Frags<HT> frg = gemm_open<HT>(pipe, p0(F_RGBH_F));
    float* const xf = TRAIN ? a.ws_xf + s * (D + kDirPad) + 4 * half : nullptr;
    {
        const float* b = bias + L::bias_off(9) + 4 * half;
#pragma unroll
        for (int t = 0; t < DT; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 bb = *reinterpret_cast<const f32x4*>(b + 32 * t + 8 * q);
#pragma unroll
                for (int i = 0; i < 4; ++i) h[16 * t + 4 * q + i] = acc[t][4 * q + i] + bb[i];
            }
    }
    // colour hidden: [feat ; gamma_4(v)] -> D/2, ReLU   (input order [feat, dir_enc]: model/official_nerf.py:89)
    f32x16 accg[HT];
    zero_acc(accg);
    gemm_part<DT, HT, TRAIN>(accg, h, pipe, p0(F_RGBH_F), frg, xf);
    gemm_part<1, HT, TRAIN>(accg, dirv, pipe, p0(F_RGBH_D), gemm_open<HT>(pipe, p0(F_RGBH_D)), TRAIN ? xf + D : nullptr);
    Frags<1> fr1 = gemm_open<1>(pipe, p0(F_RGB));
    float g[16 * HT];
    {
        const float* b = bias + L::bias_off(10) + 4 * half;
        uint32_t mw[L::mask_words];
#pragma unroll
        for (int w = 0; w < L::mask_words; ++w) mw[w] = 0;
#pragma unroll
        for (int t = 0; t < HT; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 bb = *reinterpret_cast<const f32x4*>(b + 32 * t + 8 * q);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float x = fmaxf(accg[t][4 * q + i] + bb[i], 0.f);
                    const int r = 16 * t + 4 * q + i;
                    g[r] = x;
                    if (TRAIN && !kAblateNoMask) mw[r >> 5] |= min(__float_as_uint(x), 1u) << (r & 31);   // x >= 0: bit = (x != 0), no VCC round trip
                }
            }
        if (TRAIN) {
            uint32_t* m = mask_base + (int64_t)8 * 64 * L::mask_words;
#pragma unroll
            for (int w = 0; w < L::mask_words; ++w) m[w] = mw[w];
        }
    }
    // rgb: D/2 -> 3, sigmoid (rows 0..2 of a 32-row tile live in registers 0..2 of half 0)
    zero_acc(acc1);
    gemm_part<HT, 1, TRAIN>(acc1, g, pipe, p0(F_RGB), fr1, TRAIN ? a.ws_xg + s * (D / 2) + 4 * half : nullptr);
    if (half == 0 && s < a.S) {
        const float* b = bias + L::bias_off(11);
        f32x4 o;
        o[0] = sigmoid_ref(acc1[0][0] + b[0]);
        o[1] = sigmoid_ref(acc1[0][1] + b[1]);
        o[2] = sigmoid_ref(acc1[0][2] + b[2]);
        o[3] = sigma_raw;
        *reinterpret_cast<f32x4*>(a.ws_out4 + 4 * s) = o;
    }
}

template <int D>
static hipError_t launch(const MlpFwdArgs& a, bool train, hipStream_t st) {
    dim3 grid((unsigned)(a.S_pad / kBlockSamples)), block(256);
    if (train)
        hipLaunchKernelGGL((mlp_fwd_kernel<D, true>), grid, block, 0, st, a);
    else
        hipLaunchKernelGGL((mlp_fwd_kernel<D, false>), grid, block, 0, st, a);
    return hipGetLastError();
}

hipError_t launch_mlp_fwd(int D, const MlpFwdArgs& a, bool train, hipStream_t st) {
    return D == 256 ? launch<256>(a, train, st) : launch<128>(a, train, st);
}

}  // namespace nnr
