// nnr_mlp_dgrad.hip -- fused input-gradient chain of the NeRF MLP for gfx950.
// Replaces autograd's AddmmBackward/ReluBackward/SigmoidBackward/... chain through model/official_nerf.py:60-96 for
// the *data* path: from d(rgb_pre), d(sigma_raw) per sample down to d(point), d(view dir), leaving every layer's
// pre-activation gradient in the workspace for the weight-gradient kernel.  Same structure as the forward: one wave =
// 32 samples, gradients stay in VGPRs between layers as MFMA B operands, A fragments are the transposed packed weights.
// ReLU masks come from the 1-bit-per-activation stash written by the forward (32x less traffic than re-reading h).
// MODE 2 (NNR_F_SPLIT3): every GEMM part's products as six bf16 MFMA terms (nnr_split.h); the gradient planes it leaves for the weight-
// gradient kernel are tile-major fp32, one contiguous non-temporal 1 KiB wave-store per stash store (nnr_layout.h: tile32_index); everything else unchanged.
#include "nnr_device.h"
#include "nnr_kernels.h"
#include "nnr_split.h"

namespace nnr {

// Compile-time description of one encoding feature (same block order as enc_feature in nnr_device.h).
struct EncMeta { int coord; float scale; int partner; };
__device__ __forceinline__ constexpr EncMeta enc_meta(int f, int n_real) {
    if (f >= n_real) return {0, 0.f, 0};
    if (f < 3) return {f, 1.f, -1};
    const int t = f - 3, lvl = t / 6, rem = t - 6 * lvl;
    const bool is_cos = rem >= 3;
    const float a = (float)(1 << lvl);
    return {is_cos ? rem - 3 : rem, is_cos ? -a : a, is_cos ? f - 3 : f + 3};
}

// Chain rule through gamma_L for the NR registers of one lane: returns d/d(x,y,z) of sum_r g[r] * gamma(.)_{f(r,half)}.
// `pv[r]` = scale * partner value (cos for a sin feature, -sin for a cos feature, 1 for the identity block), prepared by
// enc_partners.  The coordinate of register r is compile-time for half 0 and rotates by one for half 1 (f -> f+4, blocks of 3).
// TILE: `enc` is this sample's place in a tile-major plane (nnr_layout.h: chunk base + 4 (s & 31)) and feature f sits at
// (f >> 3) 256 + ((f >> 2) & 1) 128 + (f & 3) from there; otherwise the sample's row.
template <int NR, bool TILE = false>
__device__ __forceinline__ void enc_partners(float (&pv)[NR], const float* enc, int n_real, int half) {
    auto at = [](int f) { return TILE ? (f >> 3) * 256 + ((f >> 2) & 1) * 128 + (f & 3) : f; };
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const EncMeta m0 = enc_meta(frag_feature(r, 0), n_real), m1 = enc_meta(frag_feature(r, 1), n_real);
        const int partner = half ? at(m1.partner >= 0 ? m1.partner : 0) : at(m0.partner >= 0 ? m0.partner : 0);
        const bool has = half ? m1.partner >= 0 : m0.partner >= 0;
        const float sc = half ? m1.scale : m0.scale;
        pv[r] = sc * (has ? enc[partner] : 1.f);
    }
}
template <int NR, class G>
__device__ __forceinline__ f32x4 enc_backward(const G& g, const float (&pv)[NR], int half) {
    float g3[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int f0 = frag_feature(r, 0);
        const int c0 = f0 < 3 ? f0 : (f0 - 3) % 3;
        g3[c0] = fmaf(g(r), pv[r], g3[c0]);
    }
    float o[3] = {half ? g3[2] : g3[0], half ? g3[0] : g3[1], half ? g3[1] : g3[2]};
#pragma unroll
    for (int c = 0; c < 3; ++c) o[c] += __shfl_xor(o[c], 32, 64);
    return f32x4{o[0], o[1], o[2], 0.f};
}

NNR_TL_DECL(tl_dgrad)

template <int D, int MODE>
__global__ __launch_bounds__(256, 1) void mlp_dgrad_kernel(MlpDgradArgs a) {
    NNR_STAMP(tl_dgrad, 0);
    using L = Layout<D, MODE>;
    constexpr bool kTile = MODE == 2 && kTileGradPlanes;      // tile-major gradient planes (nnr_layout.h)
    constexpr bool kTileXr = MODE == 2 && kTileActPlanes;    // the forward's encoding planes are tile-major too (read here for the chain rule)
    using Pipe = PanelPipeT<kWavesPerBlock, mode_panel_frags(MODE), kTile>;
    constexpr int kRingF4 = kNBuf * Pipe::F4;
    constexpr int DT = L::DT, HT = L::HT;
    constexpr int HR = 16 * HT;              // registers of half a layer's outputs
    constexpr int NP = HR / 2;               // register pairs per half (the unit of hidden epilogue work)
    constexpr int HW = (HR + 31) / 32;       // mask words per half
    constexpr int PP = mode_panels(DT, HT, MODE);  // panels of one D x D/2 pass
    const int lane0 = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;

    // LDS: the panel ring of the transposed (backward) weight stream, a parking area for d(posenc) of the skip layer and
    // the head tables (density row, rgb rows, register order) -- one array, see PanelPipe in nnr_device.h
    constexpr int kPark = kWavesPerBlock * 8 * 64;
    __shared__ __attribute__((aligned(16))) f32x4 smem[kRingF4 + kPark + (L::head_floats + 3) / 4];
    float* const ltab = reinterpret_cast<float*>(smem + kRingF4 + kPark);
    for (int i = threadIdx.x; i < L::head_floats; i += 256) ltab[i] = a.packed[L::head_base + i];
    __syncthreads();   // before any DMA is in flight: the only full barrier of the kernel
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    Pipe pipe{reinterpret_cast<const f32x4*>(a.packed + L::bwd_base) + wave_u * (Pipe::PW * 64), smem, wave_u, lane0, L::bwd_panels};
    // flat or ray-mode decomposition, exactly as in mlp_fwd_kernel: in ray mode a wave walks the chunks of one ray and the transposed
    // weight stream wraps around from pass to pass
    const int n_pass = a.chunks_per_ray > 0 ? a.chunks_per_ray : 1;
    pipe.more = n_pass > 1;
    pipe.start();
    NNR_STAMP(tl_dgrad, 1);
    auto p0 = [&](int part) { return L::bwd_panel0(part); };
#pragma unroll 1
    for (int pass = 0; pass < n_pass; ++pass) {
    int lane = lane0;                 // opaque per pass: keeps lane-constant addresses from being hoisted and spilled (mlp_fwd_kernel)
    asm volatile("" : "+v"(lane));
    pipe.lane = lane;
    const int half = lane >> 5;
    const int col = lane & 31;
    f32x4* const de_lds = smem + kRingF4 + wave * (8 * 64) + lane;
    const float* const wsig = ltab + half * (16 * DT);   // density row, this half's registers
    const float* const wrgb = ltab + 2 * 16 * DT;        // rgb rows: [(2c + half) * HR + r]
    const int64_t chunk = a.chunks_per_ray > 0 ? ((int64_t)blockIdx.x * kWavesPerBlock + wave) * n_pass + pass
                                                : (int64_t)blockIdx.x * kWavesPerBlock + wave;
    const int64_t s = chunk * kChunk + col;
    const int64_t ss = s;         // row of the stash planes
    const bool live = s < a.S;  // padded samples carry zero gradients so they add nothing to the weight gradients
    const uint32_t* mask_base = a.ws_mask + ((chunk * L::n_mask_layers) * 64 + lane) * L::mask_words;

    f32x4 dout = {0.f, 0.f, 0.f, 0.f};
    if (live) dout = *reinterpret_cast<const f32x4*>(a.ws_dout4 + 4 * s);
    else if (half == 0) *reinterpret_cast<f32x4*>(a.ws_dout4 + 4 * s) = dout;   // padded rows feed the weight-gradient kernel: zero them

    float d[16 * DT];    // current D-wide gradient (d feature, then d pre-activation of hidden 8..1), rewritten in place
    f32x16 accA[HT], accB[HT];   // halves A ([0,D/2)) and B ([D/2,D)) of the gradient being computed
    uint32_t mwA[HW], mwB[HW];   // ReLU sign bits of the layer whose gradient sits in accA / accB
    auto load_mask = [&](uint32_t(&mw)[HW], int layer_idx, int hb) __attribute__((always_inline)) {
        const uint32_t* m = mask_base + (int64_t)layer_idx * 64 * L::mask_words + hb * HW;
#pragma unroll
        for (int w = 0; w < HW; ++w) mw[w] = m[w];
    };
// one epilogue unit u (registers 2u, 2u+1 of the half): d[off + 2u + i] = relu'(.) ? acc : 0, or a plain move
#define NNR_SEL_PAIR(ACC, OFF, MW)                                                                           \
    [&](int u) __attribute__((always_inline)) {                                                              \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                      \
            const int r = 2 * u + i;                                                                         \
            /* the gate bit as an all-ones / zero AND mask (one signed bit-field extract), not compare + select */ \
            d[(OFF) + r] = __uint_as_float(__float_as_uint(ACC[r >> 4][r & 15]) &                             \
                                           (uint32_t)((int32_t)(MW[r >> 5] << (31 - (r & 31))) >> 31));       \
        }                                                                                                    \
    }
#define NNR_MOVE_PAIR(ACC, OFF)                                                                              \
    [&](int u) __attribute__((always_inline)) {                                                              \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) d[(OFF) + 2 * u + i] = ACC[(2 * u + i) >> 4][(2 * u + i) & 15]; \
    }

    // ---- colour branch ----
    // d g = relu'(g) .* (Wc^T d rgb_pre): three FMAs per value against the rgb rows in LDS (a 3-deep GEMM is not MFMA work)
    float dg[HR];
    load_mask(mwA, 8, 0);
#pragma unroll
    for (int q = 0; q < HR / 4; ++q) {
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(wrgb + (0 + half) * HR + 4 * q);
        const f32x4 w1 = *reinterpret_cast<const f32x4*>(wrgb + (2 + half) * HR + 4 * q);
        const f32x4 w2 = *reinterpret_cast<const f32x4*>(wrgb + (4 + half) * HR + 4 * q);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float v = fmaf(w2[i], dout[2], fmaf(w1[i], dout[1], w0[i] * dout[0]));
            dg[4 * q + i] = ((mwA[(4 * q + i) >> 5] >> ((4 * q + i) & 31)) & 1u) ? v : 0.f;
        }
    }
    // [d h8 ; d gamma(v)] from d g.  The feature layer is folded into the colour-hidden layer (nnr_layout.h): d h8 =
    // relu'(h8) .* (W'^T d g + w_sigma^T d sigma_raw), the rank-1 density term being the accumulator's initial value.
    // Every gradient vector is stashed by the first pass that consumes it (one 16-byte store per k-group inside the MFMA
    // stream, see gemm_part); the epilogue of each half-output pass runs as side work of the following pass.
    auto init_sigma = [&](f32x16(&acc)[HT], int hb) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < HT; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 w4 = *reinterpret_cast<const f32x4*>(wsig + hb * HR + 16 * t + 4 * q);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[t][4 * q + i] = w4[i] * dout[3];
            }
    };
    load_mask(mwA, 7, 0);
    init_sigma(accA, 0);
    // d g goes to P_DG.  Row-major: this lane's row + its half's four columns; tile-major (MODE 2): block (chunk, octet 0) + 16 bytes per lane
    float* const dg_stash = kTile ? a.ws_dg + chunk * (int64_t)((D / 16) * 256) + 4 * lane : a.ws_dg + ss * (D / 2) + 4 * half;
    gemm_part<HT, HT, true>(accA, dg, pipe, p0(B_RGBH_FA), dg_stash);
    load_mask(mwB, 7, 1);
    init_sigma(accB, 1);
    gemm_part<HT, HT, false, NP, 2, 0>(accB, dg, pipe, p0(B_RGBH_FB), nullptr, NNR_SEL_PAIR(accA, 0, mwA));
    {
        float pvd[16];   // stored direction encoding (sin<->cos partners): the loads land under this short pass
        {
            const int64_t sl = live ? s : 0;
            if constexpr (kTileXr) enc_partners<16, true>(pvd, a.ws_xf + (sl >> 5) * (int64_t)((kDirPad / 8) * 256) + (sl & 31) * 4, kDirReal, half);
            else enc_partners(pvd, a.ws_xf + sl * kDirPad, kDirReal, half);
        }
        f32x16 accd[1];
        zero_acc(accd);
        gemm_part<HT, 1>(accd, dg, pipe, p0(B_RGBH_D));
        const f32x4 gv = enc_backward<16>([&](int r) { return accd[0][r]; }, pvd, half);
        if (half == 0 && live) *reinterpret_cast<f32x4*>(a.ws_dview + 4 * s) = gv;
    }
    NNR_STAMP(tl_dgrad, 2);
    NNR_STAMP(tl_dgrad, 3);

    // ---- trunk ----
    auto dh = [&](int hidden_idx /*0..7*/) -> float* {
        if constexpr (kTile) return a.ws_dh + (int64_t)hidden_idx * a.S_pad * D + chunk * (int64_t)((D / 8) * 256) + 4 * lane;
        else return a.ws_dh + ((int64_t)hidden_idx * a.S_pad + ss) * D + 4 * half;
    };
    // Invariant from here on: d[0, HR) holds half A of the newest gradient, accB its half B still to be masked (mwB).

    // one transposed D x D layer at panel pa: consumes the gradient in d (stashing it to `stash`), produces the gradient of
    // the layer below, masked by the sign bits of hidden layer `mask_idx`
    auto bwd_layer = [&](int pa, float* stash, int mask_idx) __attribute__((always_inline)) {
        zero_acc(accA);
        load_mask(mwA, mask_idx, 0);
        // pass A: the first half of the k-groups only reads d[0,HR); the previous gradient's half B is finished meanwhile
        gemm_part<DT, HT, true, NP, 2, 0>(accA, d, pipe, pa, stash, NNR_SEL_PAIR(accB, HR, mwB));
        load_mask(mwB, mask_idx, 1);
        zero_acc(accB);
        // pass B: half A of the new gradient replaces d[0,HR) in place, one k-group behind the reads
        pipe.part_pre = MODE == 2 && PP >= 2 ? 6 : 0;   // pass A's last rows stashed (nnr_split.h)
        gemm_part<DT, HT, false, NP, 2, 1>(accB, d, pipe, pa + PP, nullptr, NNR_SEL_PAIR(accA, 0, mwA));
        pipe.part_pre = 0;
    };
    // hidden 8,7,6 -> d pre-activation of 7,6,5
#pragma unroll 1
    for (int l = 0; l < 3; ++l) bwd_layer(p0(B_L8A) + 2 * PP * l, dh(7 - l), 6 - l);
    NNR_STAMP(tl_dgrad, 4);
    // hidden 5 (skip layer), three passes over W5^T: rows [D, D+63) -> d posenc (parked in LDS until the end), rows [0,D) -> d h4
    {
        f32x16 acce[2];
        zero_acc(acce);
        load_mask(mwA, 3, 0);
        gemm_part<DT, 2, true, NP, 2, 0>(acce, d, pipe, p0(B_L5E), dh(4), NNR_SEL_PAIR(accB, HR, mwB));
        load_mask(mwB, 3, 1);
        zero_acc(accA);
        auto park = [&](int q) __attribute__((always_inline)) {
            de_lds[q * 64] = f32x4{acce[q >> 2][4 * (q & 3)], acce[q >> 2][4 * (q & 3) + 1], acce[q >> 2][4 * (q & 3) + 2],
                                   acce[q >> 2][4 * (q & 3) + 3]};
        };
        gemm_part<DT, HT, false, 8, 1, 0>(accA, d, pipe, p0(B_L5HA), nullptr, park);
    }
    zero_acc(accB);
    gemm_part<DT, HT, false, NP, 2, 1>(accB, d, pipe, p0(B_L5HB), nullptr, NNR_SEL_PAIR(accA, 0, mwA));
    NNR_STAMP(tl_dgrad, 5);
    // hidden 4,3,2 -> d pre-activation of 3,2,1
#pragma unroll 1
    for (int l = 0; l < 3; ++l) bwd_layer(p0(B_L4A) + 2 * PP * l, dh(3 - l), 2 - l);
    NNR_STAMP(tl_dgrad, 6);
    // hidden 1: d posenc = W1^T d1 + (skip-layer part parked in LDS), then the chain rule through gamma_10 -> d point
    {
        float pve[32];   // stored position encoding (sin<->cos partners): the loads land under the last pass
        {
            const int64_t sl = live ? s : 0;
            if constexpr (kTileXr) enc_partners<32, true>(pve, a.ws_xe + (sl >> 5) * (int64_t)((kPosPad / 8) * 256) + (sl & 31) * 4, kPosReal, half);
            else enc_partners(pve, a.ws_xe + sl * kPosPad, kPosReal, half);
        }
        f32x16 acc2[2];
        zero_acc(acc2);
        gemm_part<DT, 2, true, NP, 2, 0>(acc2, d, pipe, p0(B_L1), dh(0), NNR_SEL_PAIR(accB, HR, mwB));
        float de[32];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const f32x4 v = de_lds[q * 64];
#pragma unroll
            for (int i = 0; i < 4; ++i) de[4 * q + i] = acc2[q >> 2][4 * (q & 3) + i] + v[i];
        }
        const f32x4 gp = enc_backward<32>([&](int r) { return de[r]; }, pve, half);
        if (half == 0 && live) *reinterpret_cast<f32x4*>(a.ws_dpts + 4 * s) = gp;
    }
    NNR_STAMP(tl_dgrad, 7);
#undef NNR_SEL_PAIR
#undef NNR_MOVE_PAIR
    pipe.next_pass(pass + 2 < n_pass);
    }   // pass
}

#if defined(NNR_TIMELINE) && defined(NNR_DGRAD_D) && NNR_DGRAD_D == 256 && !defined(NNR_DGRAD_MODE)
extern "C" int nnr_timeline_dgrad(unsigned long long* host32) {
    return (int)hipMemcpyFromSymbol(host32, HIP_SYMBOL(tl_dgrad), 32 * sizeof(unsigned long long));
}
#endif

// one D per translation unit, see nnr_mlp_fwd.hip
#ifdef NNR_DGRAD_D
#ifndef NNR_DGRAD_MODE
#define NNR_DGRAD_MODE 0
#endif
template <>
hipError_t launch_mlp_dgrad_variant<NNR_DGRAD_D, NNR_DGRAD_MODE>(const MlpDgradArgs& a, hipStream_t st) {
    dim3 grid((unsigned)(a.chunks_per_ray > 0 ? a.S_pad / kBlockSamples / a.chunks_per_ray : a.S_pad / kBlockSamples)), block(256);
    prof_before(PROF_DGRAD, st);
    hipLaunchKernelGGL((mlp_dgrad_kernel<NNR_DGRAD_D, NNR_DGRAD_MODE>), grid, block, 0, st, a);
    prof_after(PROF_DGRAD, st);
    return hipGetLastError();
}
#else
hipError_t launch_mlp_dgrad(int D, const MlpDgradArgs& a, hipStream_t st, int mode) {
    if (mode == 3) return D == 256 ? launch_mlp_dgrad_variant<256, 3>(a, st) : launch_mlp_dgrad_variant<128, 3>(a, st);      // nnr_mlp_dgrad_f16.hip
    if (mode == 2) return D == 256 ? launch_mlp_dgrad_variant<256, 2>(a, st) : launch_mlp_dgrad_variant<128, 2>(a, st);
    return D == 256 ? launch_mlp_dgrad_variant<256>(a, st) : launch_mlp_dgrad_variant<128>(a, st);
}
#endif

}  // namespace nnr
