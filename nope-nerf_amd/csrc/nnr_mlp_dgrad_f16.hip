// nnr_mlp_dgrad_f16.hip -- the fused input-gradient chain of nnr_mlp_dgrad.hip with every fp32 product taken as three fp16 MFMA terms of
// two-term operands (NNR_F_SPLIT2, Layout<D, 3>; the arithmetic: nnr_split2.h).  Replaces autograd's AddmmBackward / ReluBackward / ... chain
// through model/official_nerf.py:60-96 for the data path, from d(rgb_pre), d(sigma_raw) per sample down to d(point), d(view dir), leaving every
// layer's pre-activation gradient in the (tile-major fp32) gradient planes for the weight-gradient kernel.  Same decomposition, planes and
// outputs as the six-term kernel; what differs:
//   * gradients span twenty orders of magnitude from sample to sample (the compositing weights) and fp16 does not: the chain of a sample runs
//     in a SCALED domain, d s with s = the power of two that puts the sample's largest |d g| (or |d sigma_raw| max |w_sigma|) into [2^7, 2^8) --
//     eight binades of head room for the growth of ONE layer, twenty-two below before the first term leaves the normal range (the residual term
//     is carried at 2^11, nnr_split2.h); what is stashed for the weight gradient and what leaves the kernel is multiplied by 1 / s again (exact).
//     The scale is RE-CENTRED after every layer (recentre(), below): a trained network's layers amplify the gradient -- a factor 1.9 per layer is
//     2^8 over the chain, the next conversion overflows to inf and the step's gradients are NaN behind a finite loss.  That happened: the
//     reference's train.py on a synthetic scene, per-image losses off, at iteration ~300 of the first build of this kernel, which scaled once
//     at the top (tests/test_gpu_loop_rate.py caught it; tests/test_gpu_split3.py::test_two_term_input_gradient_survives_amplifying_layers pins it);
//   * between the layers a lane holds the packed terms of the scaled gradient, made once by the epilogue unit that finishes a pair:
//     accumulator pair -> ReLU' (AND with the gate bits) -> 1 / s_w of the weights just applied -> [1 / s -> every second unit one whole-block
//     non-temporal store to the gradient plane] -> split.
#include "nnr_device.h"
#include "nnr_kernels.h"
#include "nnr_split2.h"

namespace nnr {

static_assert(kTileActPlanes && kTileGradPlanes, "the planes of a three-term / two-term training workspace are tile-major");

// Compile-time description of one encoding feature and the chain rule through gamma_L: as in nnr_mlp_dgrad.hip (tile-major planes only)
namespace f16dg {
// unit_dgrad for the pair of registers (r, r + 1), r even: their gates sit at bits 31 - (r & 31) and 30 - (r & 31) of the word (gate_append2's order); rr folds after unrolling
__device__ __forceinline__ void sel_unit(float a0, float a1, uint32_t word, float invw, float sinv, float& t0, float& t1, uint32_t& h, uint32_t& m, int rr, float& mx) {
    switch (rr) {
#define NNR_SU(R) case R: unit_dgrad<31 - (R), 30 - (R)>(a0, a1, word, invw, sinv, t0, t1, h, m, mx); break;
        NNR_SU(0) NNR_SU(2) NNR_SU(4) NNR_SU(6) NNR_SU(8) NNR_SU(10) NNR_SU(12) NNR_SU(14) NNR_SU(16) NNR_SU(18) NNR_SU(20) NNR_SU(22) NNR_SU(24) NNR_SU(26) NNR_SU(28)
#undef NNR_SU
        default: unit_dgrad<1, 0>(a0, a1, word, invw, sinv, t0, t1, h, m, mx); break;
    }
}
struct EncMeta { int coord; float scale; int partner; };
__device__ __forceinline__ constexpr EncMeta enc_meta(int f, int n_real) {
    if (f >= n_real) return {0, 0.f, 0};
    if (f < 3) return {f, 1.f, -1};
    const int t = f - 3, lvl = t / 6, rem = t - 6 * lvl;
    const bool is_cos = rem >= 3;
    const float a = (float)(1 << lvl);
    return {is_cos ? rem - 3 : rem, is_cos ? -a : a, is_cos ? f - 3 : f + 3};
}
// pv[r] = scale * partner value of register r's feature (cos for a sin feature, -sin for a cos feature, 1 for the identity block);
// `enc`: this sample's place in the tile-major plane (chunk base + 4 (s & 31)), feature f at (f >> 3) 256 + ((f >> 2) & 1) 128 + (f & 3)
template <int NR>
__device__ __forceinline__ void enc_partners(float (&pv)[NR], const float* enc, int n_real, int half) {
    auto at = [](int f) { return (f >> 3) * 256 + ((f >> 2) & 1) * 128 + (f & 3); };
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
}  // namespace f16dg

template <int D>
__global__ __launch_bounds__(256, 1) void mlp_dgrad_f16_kernel(MlpDgradArgs a) {
    using L = Layout<D, 3>;
    using Pipe = Split2PipeT<false>;
    constexpr int kRingF4 = kNBuf * Pipe::F4;
    constexpr int DT = L::DT, HT = L::HT;
    constexpr int HR = 16 * HT;              // registers of half a layer's outputs
    constexpr int NP = HR / 2;               // register pairs per half (the unit of hidden epilogue work)
    constexpr int HW = (HR + 31) / 32;       // mask words per half
    constexpr int PP = mode_panels(DT, HT, 3);  // panels of one D x D/2 pass
    // stash stores a dense pass certainly issues while it consumes its last panel (gemm_part2's PRE of the part behind it): an "ahead" pass one per row but the last, a "behind" pass one per row
    constexpr int kPreA = mode_gp(HT, 3) - 1, kPreB = mode_gp(HT, 3);
    const int lane0 = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;

    // LDS: the panel ring of the transposed (backward) weight stream, a parking area for d(posenc) of the skip layer, the head tables
    // (density row, rgb rows, register order) and the weight scales
    constexpr int kPark = kWavesPerBlock * 8 * 64;
    constexpr int kTab = L::head_floats + L::scale_floats;
    __shared__ __attribute__((aligned(16))) f32x4 smem[kRingF4 + kPark + (kTab + 3) / 4];
    float* const ltab = reinterpret_cast<float*>(smem + kRingF4 + kPark);
    for (int i = threadIdx.x; i < kTab; i += 256) ltab[i] = a.packed[L::head_base + i];
    __shared__ uint32_t wg_max[9];      // the workgroup's largest |gradient| per plane P_DH1..8 ([0..8)) and P_DG ([8]), as integers
    if (threadIdx.x < 9) wg_max[threadIdx.x] = 0;
    __syncthreads();   // before any DMA is in flight: the only full barrier in front of the passes
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    Pipe pipe{reinterpret_cast<const f32x4*>(a.packed + L::bwd_base) + wave_u * (Pipe::PW * 64), smem, wave_u, lane0, L::bwd_panels};
    const int n_pass = a.chunks_per_ray > 0 ? a.chunks_per_ray : 1;
    pipe.more = n_pass > 1;
    pipe.start();
    auto p0 = [&](int part) { return L::bwd_panel0(part); };
    const float* const lscale = ltab + L::head_floats;      // [16] s_w by scale slot ([9]: max |w_sigma|), [16] 1 / s_w
    auto uniform = [](float v) __attribute__((always_inline)) { return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(v))); };
#pragma unroll 1
    for (int pass = 0; pass < n_pass; ++pass) {
    int lane = lane0;                 // opaque per pass: keeps lane-constant addresses from being hoisted and spilled (nnr_mlp_fwd.hip)
    asm volatile("" : "+v"(lane));
    pipe.lane = lane;
    const int half = lane >> 5;
    const int col = lane & 31;
    const int lane_off = 16 * lane;
    f32x4* const de_lds = smem + kRingF4 + wave * (8 * 64) + lane;
    const float* const wsig = ltab + half * (16 * DT);   // density row, this half's registers
    const float* const wrgb = ltab + 2 * 16 * DT;        // rgb rows: [(2c + half) * HR + r]
    const int64_t chunk = a.chunks_per_ray > 0 ? ((int64_t)blockIdx.x * kWavesPerBlock + wave_u) * n_pass + pass
                                                : (int64_t)blockIdx.x * kWavesPerBlock + wave_u;
    const int64_t s = chunk * kChunk + col;
    const bool live = s < a.S;  // padded samples carry zero gradients so they add nothing to the weight gradients
    const uint32_t* mask_base = a.ws_mask + ((chunk * L::n_mask_layers) * 64 + lane) * L::mask_words;

    f32x4 dout = {0.f, 0.f, 0.f, 0.f};
    if (live) dout = *reinterpret_cast<const f32x4*>(a.ws_dout4 + 4 * s);
    else if (half == 0) *reinterpret_cast<f32x4*>(a.ws_dout4 + 4 * s) = dout;   // padded rows feed the weight-gradient kernel: zero them

    uint32_t ph[8 * DT], pm[8 * DT];      // packed terms of the current D-wide SCALED gradient (pairs [0, NP): half A, [NP, 2 NP): half B), rewritten in place
    f32x16 accA[HT], accB[HT];           // halves A ([0,D/2)) and B ([D/2,D)) of the gradient being computed
    uint32_t mwA[HW], mwB[HW];           // ReLU sign bits of the layer whose gradient sits in accA / accB
    f32x2 keep = {0.f, 0.f};
    float mxd = 0.f;                     // running maximum of the |true gradients| the units stash into the current plane
    float sS, sInv;      // the sample's scale (of the packed terms being made) and its inverse
    float cW = 1.f;      // the factor the units of the plane in progress apply on top of 1 / s_w to move from the previous plane's scale to sS
    int esS = 127;       // sS = 2^(esS - 127)
    // A plane is complete (both halves of every sample of this wave): its maximum to the workgroup's table, and the NEXT plane's scale from this
    // sample's own largest magnitude -- both lanes of a sample hold half of its features, mxd is the lane's running max of the TRUE values.
    // The packed terms of this plane stay as they are (scale sS_old); the accumulators made from them carry sS_old, and the units that finish the
    // next plane multiply by cW = sS_new / sS_old on the way (exact: a power of two) and stash with 1 / sS_new.
    auto flush_max = [&](int plane, bool recentre = true) __attribute__((always_inline)) {
        const float m = wave_max_f32(mxd);
        if (lane == 0) atomicMax(&wg_max[plane], __float_as_uint(m));
        if (recentre) {
            const float sm = fmaxf(mxd, __shfl_xor(mxd, 32, 64)) * sS;                      // the sample's largest scaled magnitude in this plane
            const int eb = (int)((__float_as_uint(sm) >> 23) & 255u);
            int es = esS + (134 - eb);                                                       // ... back to [2^7, 2^8)
            es = es > 227 ? 227 : (es < 27 ? 27 : es);
            es = (sm > 0.f && eb != 255) ? es : esS;                                         // an all-zero (or already non-finite) sample keeps its scale
            cW = __uint_as_float((uint32_t)(127 + es - esS) << 23);
            esS = es;
            sS = __uint_as_float((uint32_t)es << 23);
            sInv = __uint_as_float((uint32_t)(254 - es) << 23);
        }
        mxd = 0.f;
    };
    auto load_mask = [&](uint32_t(&mw)[HW], int layer_idx, int hb) __attribute__((always_inline)) {
        const uint32_t* m = mask_base + (int64_t)layer_idx * 64 * L::mask_words + hb * HW;
#pragma unroll
        for (int w = 0; w < HW; ++w) mw[w] = m[w];
    };

    // ---- colour branch ----
    // d g = relu'(g) .* (Wc^T d rgb_pre): three FMAs per value against the rgb rows in LDS (a 3-deep GEMM is not MFMA work)
    uint32_t gh[NP], gm[NP];
    {
        float dg[HR];
        load_mask(mwA, 8, 0);
        float mx = 0.f;
#pragma unroll
        for (int q = 0; q < HR / 4; ++q) {
            const f32x4 w0 = *reinterpret_cast<const f32x4*>(wrgb + (0 + half) * HR + 4 * q);
            const f32x4 w1 = *reinterpret_cast<const f32x4*>(wrgb + (2 + half) * HR + 4 * q);
            const f32x4 w2 = *reinterpret_cast<const f32x4*>(wrgb + (4 + half) * HR + 4 * q);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float v = fmaf(w2[i], dout[2], fmaf(w1[i], dout[1], w0[i] * dout[0]));
                dg[4 * q + i] = ((mwA[(4 * q + i) >> 5] >> (31 - ((4 * q + i) & 31))) & 1u) ? v : 0.f;      // (the forward's bit order: gate_append2)
                mx = fmaxf(mx, fabsf(dg[4 * q + i]));
            }
        }
        // d g goes to P_DG: block (chunk, octet q) + 16 bytes per lane
        const char* const dgp = reinterpret_cast<const char*>(a.ws_dg + chunk * (int64_t)((D / 16) * 256));
#pragma unroll
        for (int q = 0; q < HR / 4; ++q) tile_store(dgp, lane_off, q, f32x4{dg[4 * q], dg[4 * q + 1], dg[4 * q + 2], dg[4 * q + 3]});
        // the sample's scale: the largest magnitude that enters its chain -- its d g (both half-waves) and the density head's rank-1 term --
        // to [2^7, 2^8); a power of two from the exponent field, clamped to 2^+-100; 1 for an all-zero sample
        mxd = mx;
        flush_max(8, false);                          // P_DG (every sample's largest |d g|, unscaled); the first scale is set right here
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        mx = fmaxf(mx, fabsf(dout[3]) * lscale[9]);
        const int eb = (int)((__float_as_uint(mx) >> 23) & 255u);
        int es = 261 - eb;
        es = es > 227 ? 227 : (es < 27 ? 27 : es);
        esS = mx > 0.f ? es : 127;
        sS = __uint_as_float((uint32_t)esS << 23);
        sInv = __uint_as_float((uint32_t)(254 - esS) << 23);
        cW = 1.f;
        split2_all(gh, gm, [&](int r) { return dg[r] * sS; });
    }
    // [d h8 ; d gamma(v)] from d g.  The feature layer is folded into the colour-hidden layer (nnr_layout.h): d h8 =
    // relu'(h8) .* (W'^T d g + w_sigma^T d sigma_raw), the rank-1 density term being the accumulator's initial value (in the
    // accumulator's units: s s_w of the merged matrix' scale slot).
    const float dsig = dout[3] * sS * uniform(lscale[8]);
    auto init_sigma = [&](f32x16(&acc)[HT], int hb) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < HT; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 w4 = *reinterpret_cast<const f32x4*>(wsig + hb * HR + 16 * t + 4 * q);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[t][4 * q + i] = w4[i] * dsig;
            }
    };
    // One epilogue unit: pair u of a half (registers 2 u, 2 u + 1 of ACC) -> relu'(.) ? acc : 0 (the gate bit -- register r at bit 31 - (r & 31),
    // gate_append2 -- as an all-ones / zero AND mask: shift to the sign position, arithmetic shift back), times INV (1 / s_w of the weights just applied) = the scaled gradient; its true value (times 1 / s) goes to
    // the gradient plane PLANE (block (this chunk, octet BLK0 + u / 2): every second unit stores an octet), its packed terms to pair OFFP + u
#define NNR_SELECT(ACC, OFFP, MW, INV, PLANE, BLK0)                                                              \
    [&](int u) __attribute__((always_inline)) {                                                                  \
        const int r = 2 * u;                                                                                     \
        float t0, t1;                                                                                            \
        f16dg::sel_unit(ACC[r >> 4][r & 15], ACC[(r + 1) >> 4][(r + 1) & 15], MW[r >> 5], INV, sInv, t0, t1, ph[(OFFP) + u], pm[(OFFP) + u], r & 31, mxd); \
        if (u & 1) tile_store(PLANE, lane_off, (BLK0) + (u >> 1), f32x4{keep[0], keep[1], t0, t1});              \
        else keep = f32x2{t0, t1};                                                                               \
    }
    auto dh = [&](int hidden_idx /*0..7*/) -> const char* {      // block (this chunk, octet 0) of the gradient plane of hidden layer hidden_idx + 1
        return reinterpret_cast<const char*>(a.ws_dh + (int64_t)hidden_idx * a.S_pad * D + chunk * (int64_t)((D / 8) * 256));
    };
    auto inv_scale = [&](int slot) __attribute__((always_inline)) { return uniform(lscale[16 + slot]); };

    load_mask(mwA, 7, 0);
    init_sigma(accA, 0);
    gemm_part2<HT, HT>(accA, gh, gm, pipe, p0(B_RGBH_FA));
    load_mask(mwB, 7, 1);
    init_sigma(accB, 1);
    {
        const float inv = inv_scale(8) * cW;      // (per lane: the plane's re-centring factor rides on 1 / s_w)
        const char* const pl = dh(7);
        gemm_part2<HT, HT, NP, 0, 4, 2, 0>(accB, gh, gm, pipe, p0(B_RGBH_FB), NNR_SELECT(accA, 0, mwA, inv, pl, 0));
    }
    {
        float pvd[16];   // stored direction encoding (sin<->cos partners): the loads land under this short pass
        {
            const int64_t sl = live ? s : 0;
            f16dg::enc_partners<16>(pvd, a.ws_xf + (sl >> 5) * (int64_t)((kDirPad / 8) * 256) + (sl & 31) * 4, kDirReal, half);
        }
        f32x16 accd[1];
        zero_acc(accd);
        gemm_part2<HT, 1>(accd, gh, gm, pipe, p0(B_RGBH_D));
        const float c = inv_scale(8) * sInv;
        const f32x4 gv = f16dg::enc_backward<16>([&](int r) { return accd[0][r] * c; }, pvd, half);
        if (half == 0 && live) *reinterpret_cast<f32x4*>(a.ws_dview + 4 * s) = gv;
    }

    // ---- trunk ----
    // Invariant from here on: pairs [0, NP) hold half A of the newest gradient, accB its half B still to be masked (mwB).
    // One transposed D x D layer at panel pa: consumes the gradient of hidden k + 1 (finishing its half B, which the weights of scale slot
    // k + 1 produced), produces the gradient of hidden k (weights of slot k = parameter k), masked by the sign bits of hidden k
    // (pre: the part before this one stashed -- not so for the first layer, which follows the direction-encoding part)
    auto bwd_layer = [&](int pa, int k, bool pre) __attribute__((always_inline)) {
        zero_acc(accA);
        load_mask(mwA, k - 1, 0);
        {   // pass A: its first half of rows only reads pairs [0, NP); the previous gradient's half B is finished meanwhile
            const float inv = inv_scale(k + 1) * cW;
            const char* const pl = dh(k);
            gemm_part2<DT, HT, NP, 1, 0, 2, kPreB>(accA, ph, pm, pipe, pa, NNR_SELECT(accB, NP, mwB, inv, pl, HR / 4), pre);
        }
        flush_max(k);      // the gradient of hidden k + 1 (plane P_DH1 + k) is complete
        load_mask(mwB, k - 1, 1);
        zero_acc(accB);
        {   // pass B: half A of the new gradient replaces pairs [0, NP) in place, one row behind the reads
            const float inv = inv_scale(k) * cW;
            const char* const pl = dh(k - 1);
            gemm_part2<DT, HT, NP, 2, 0, 2, kPreA>(accB, ph, pm, pipe, pa + PP, NNR_SELECT(accA, 0, mwA, inv, pl, 0));
        }
    };
    // hidden 8,7,6 -> d pre-activation of 7,6,5
#pragma unroll 1
    for (int l = 0; l < 3; ++l) bwd_layer(p0(B_L8A) + 2 * PP * l, 7 - l, l > 0);
    float sInv4 = 1.f;
    // hidden 5 (skip layer), three passes over W5^T: rows [D, D+63) -> d posenc (parked in LDS until the end), rows [0,D) -> d h4
    {
        f32x16 acce[2];
        zero_acc(acce);
        load_mask(mwA, 3, 0);
        {
            const float inv = inv_scale(5) * cW;
            const char* const pl = dh(4);
            gemm_part2<DT, 2, NP, 1, 0, 2, kPreB>(acce, ph, pm, pipe, p0(B_L5E), NNR_SELECT(accB, NP, mwB, inv, pl, HR / 4));
        }
        sInv4 = sInv;      // (the encoding rows of the skip layer were multiplied into plane 4's terms: their scale, before it is re-centred)
        flush_max(4);
        load_mask(mwB, 3, 1);
        zero_acc(accA);
        auto park = [&](int q) __attribute__((always_inline)) {
            de_lds[q * 64] = f32x4{acce[q >> 2][4 * (q & 3)], acce[q >> 2][4 * (q & 3) + 1], acce[q >> 2][4 * (q & 3) + 2],
                                   acce[q >> 2][4 * (q & 3) + 3]};
        };
        gemm_part2<DT, HT, 8, 0, 1, 0, 0>(accA, ph, pm, pipe, p0(B_L5HA), park);
    }
    zero_acc(accB);
    {
        const float inv = inv_scale(4) * cW;
        const char* const pl = dh(3);
        gemm_part2<DT, HT, NP, 2, 0, 2, 0>(accB, ph, pm, pipe, p0(B_L5HB), NNR_SELECT(accA, 0, mwA, inv, pl, 0));
    }
    // hidden 4,3,2 -> d pre-activation of 3,2,1
#pragma unroll 1
    for (int l = 0; l < 3; ++l) bwd_layer(p0(B_L4A) + 2 * PP * l, 3 - l, true);
    // hidden 1: d posenc = W1^T d1 + (skip-layer part parked in LDS), then the chain rule through gamma_10 -> d point
    {
        float pve[32];   // stored position encoding (sin<->cos partners): the loads land under the last pass
        {
            const int64_t sl = live ? s : 0;
            f16dg::enc_partners<32>(pve, a.ws_xe + (sl >> 5) * (int64_t)((kPosPad / 8) * 256) + (sl & 31) * 4, kPosReal, half);
        }
        f32x16 acc2[2];
        zero_acc(acc2);
        {
            const float inv = inv_scale(1) * cW;
            const char* const pl = dh(0);
            gemm_part2<DT, 2, NP, 1, 0, 2, kPreB>(acc2, ph, pm, pipe, p0(B_L1), NNR_SELECT(accB, NP, mwB, inv, pl, HR / 4));
        }
        flush_max(0, false);      // (the last plane: the scale of its terms is what acc2 carries)
        const float c1 = inv_scale(0) * sInv, c4 = inv_scale(4) * sInv4;      // out of the accumulators' units: weights of slot 0 / slot 4, the scale of the terms each was made from
        float de[32];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const f32x4 v = de_lds[q * 64];
#pragma unroll
            for (int i = 0; i < 4; ++i) de[4 * q + i] = acc2[q >> 2][4 * (q & 3) + i] * c1 + v[i] * c4;
        }
        const f32x4 gp = f16dg::enc_backward<32>([&](int r) { return de[r]; }, pve, half);
        if (half == 0 && live) *reinterpret_cast<f32x4*>(a.ws_dpts + 4 * s) = gp;
    }
#undef NNR_SELECT
    pipe.next_pass(pass + 2 < n_pass);
    }   // pass
    __syncthreads();      // the workgroup's maxima to the launch's table (the weight-gradient kernel's scales): P_DH1..8 at [8, 16), P_DG at [16]
    if (a.plane_max != nullptr && threadIdx.x < 9) atomicMax(reinterpret_cast<uint32_t*>(a.plane_max) + 8 + threadIdx.x, wg_max[threadIdx.x]);
}

// one D per translation unit (csrc/build.py: -DNNR_DGRAD_D=..), as for nnr_mlp_dgrad.hip
#ifdef NNR_DGRAD_D
template <>
hipError_t launch_mlp_dgrad_variant<NNR_DGRAD_D, 3>(const MlpDgradArgs& a, hipStream_t st) {
    dim3 grid((unsigned)(a.chunks_per_ray > 0 ? a.S_pad / kBlockSamples / a.chunks_per_ray : a.S_pad / kBlockSamples)), block(256);
    prof_before(PROF_DGRAD, st);
    hipLaunchKernelGGL((mlp_dgrad_f16_kernel<NNR_DGRAD_D>), grid, block, 0, st, a);
    prof_after(PROF_DGRAD, st);
    return hipGetLastError();
}
#endif

}  // namespace nnr
