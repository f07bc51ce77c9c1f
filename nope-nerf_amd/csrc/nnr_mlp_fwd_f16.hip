// nnr_mlp_fwd_f16.hip -- the fused NeRF MLP forward of nnr_mlp_fwd.hip with every fp32 product taken as three fp16 MFMA terms of two-term
// operands (NNR_F_SPLIT2, Layout<D, 3>; the arithmetic: nnr_split2.h).  Restates, per sample: model/rendering.py:184-195 (z, points, view dir)
// and model/official_nerf.py:60-96 (the MLP).  Same decomposition (one wave = 32 samples, ray mode / flat mode, weights through the DMA-fed
// LDS ring, every D-wide layer as two half-output passes with the epilogue of one pass hidden in the MFMA stream of the next), same planes,
// same outputs as the six-term kernel -- what differs:
//   * between the layers a lane holds the PACKED TERMS of the activations (pair p -> ph[p], pm[p]: 32 bits per value, as many registers as the
//     fp32 values took), made once by the epilogue unit that finishes the pair; a GEMM row is 3 MT MFMAs + 3 MT fragment refills and nothing else;
//   * an epilogue unit = read the accumulator pair, undo the weight scale (one packed multiply by 1 / s_w: the accumulator started at s_w bias),
//     ReLU, [training: two gate bits, every second unit one whole-block non-temporal store of the four fp32 values to the activation plane],
//     [hidden 8: two FMAs of the density head], split;
//   * the density head is summed inside the units of hidden 8 (the fp32 values exist only there).
// Roofline: the matrix pipe in fp16 (2.5 PFLOP/s dense): 3 x 528 000 MACs per sample executed.
#include "nnr_device.h"
#include "nnr_kernels.h"
#include "nnr_split2.h"

#include <type_traits>

namespace nnr {

static_assert(kTileActPlanes, "the activation planes of a three-term / two-term training workspace are tile-major");

template <int D, bool TRAIN>
__global__ __launch_bounds__(256, 1) void mlp_fwd_f16_kernel(MlpFwdArgs a) {
    using L = Layout<D, 3>;
    using Pipe = Split2PipeT<false>;
    constexpr int kRingF4 = kNBuf * Pipe::F4;
    constexpr int DT = L::DT, HT = L::HT;
    const int lane0 = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;

    constexpr int kPark = kWavesPerBlock * 12 * 64;   // per wave 12 16-byte slots per lane: the packed terms of posenc (8) + direnc (4)
    __shared__ __attribute__((aligned(16))) f32x4 smem[kRingF4 + kPark + (L::table_floats + 3) / 4];
    float* const ltab = reinterpret_cast<float*>(smem + kRingF4 + kPark);
    for (int i = threadIdx.x; i < L::table_floats; i += 256) ltab[i] = a.packed[L::bias_base + i];
    __shared__ uint32_t wg_max[9];      // training: the workgroup's largest stashed value per activation plane P_XH1..8 (as integers: the values are >= 0); [8]: the position encoding's
    if (TRAIN && threadIdx.x < 9) wg_max[threadIdx.x] = 0;
    __syncthreads();   // before any DMA is in flight: the only full barrier in front of the passes
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    Pipe pipe{reinterpret_cast<const f32x4*>(a.packed) + wave_u * (Pipe::PW * 64), smem, wave_u, lane0, L::fwd_panels};
    const int n_pass = a.chunks_per_ray > 0 ? a.chunks_per_ray : 1;
    pipe.more = n_pass > 1;
    pipe.start();
    float cT = 1.f, cr = 0.f, cg = 0.f, cb = 0.f, cz = 0.f, cw = 0.f;      // fused compositing (inference, ray mode), carried across the chunks
#pragma unroll 1
    for (int pass = 0; pass < n_pass; ++pass) {
    int lane = lane0;      // opaque per pass: keeps lane-constant addresses from being hoisted out of the pass loop and spilled (nnr_mlp_fwd.hip)
    asm volatile("" : "+v"(lane));
    pipe.lane = lane;
    const int half = lane >> 5;
    const int col = lane & 31;
    f32x4* const park = smem + kRingF4 + wave * (12 * 64) + lane;
    // wave-uniform: every plane address below is (scalar block base of this chunk) + 16 bytes per lane
    const int64_t chunk_id = a.chunks_per_ray > 0 ? ((int64_t)blockIdx.x * kWavesPerBlock + wave_u) * n_pass + pass
                                                   : (int64_t)blockIdx.x * kWavesPerBlock + wave_u;
    const int64_t s = chunk_id * kChunk + col;                                     // this lane's sample
    const int64_t sc = s < a.S ? s : a.S - 1;                                      // clamp: padded samples recompute the last one
    const int ray = (int)(sc / a.N);
    const int j = (int)(sc - (int64_t)ray * a.N);
    const int lane_off = 16 * lane;

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
    const bool fuse = !TRAIN && a.fuse_rgb != nullptr;
    if (half == 0 && s < a.S && !fuse) a.ws_z[s] = z;

    constexpr int HR = 16 * HT;              // registers of half a layer's outputs
    constexpr int NP = HR / 2;               // register pairs per half (the unit of hidden epilogue work)
    constexpr int HW = (HR + 31) / 32;       // mask words per half
    constexpr int PP = mode_panels(DT, HT, 3);  // panels of one D x D/2 pass
    // stash stores a dense pass certainly issues while it consumes its last panel (gemm_part2's PRE of the part behind it): an "ahead" pass one per row but the last, a "behind" pass one per row
    constexpr int kPreA = TRAIN ? mode_gp(HT, 3) - 1 : 0, kPreB = TRAIN ? mode_gp(HT, 3) : 0;

    // ---- encodings, straight into fragment layout; split once, parked in LDS until their layers ----
    uint32_t eh[16], em[16];      // gamma_10(p): 63 -> 64 values = 16 pairs
    {
        float e[32];
#pragma unroll
        for (int r = 0; r < 32; ++r) e[r] = enc_register(r, half, kPosReal, px, py, pz);
        if constexpr (TRAIN) {
            const char* const xe = reinterpret_cast<const char*>(a.ws_xe + chunk_id * (int64_t)((kPosPad / 8) * 256));
#pragma unroll
            for (int q = 0; q < 8; ++q) tile_store(xe, lane_off, q, f32x4{e[4 * q], e[4 * q + 1], e[4 * q + 2], e[4 * q + 3]});
            // the plane's largest magnitude (the weight gradient's two-term job against this plane scales by it: nnr_wgrad.hip wgrad_job_enc2): the
            // identity block's coordinates, or 1 (the sines and cosines)
            const float me = wave_max_f32(fmaxf(1.f, fmaxf(fabsf(px), fmaxf(fabsf(py), fabsf(pz)))));
            if (lane == 0) atomicMax(&wg_max[8], __float_as_uint(me));
        }
        split2_all(eh, em, [&](int r) { return e[r]; });
        float dirv[16];  // gamma_4(v): 27 -> 32
#pragma unroll
        for (int r = 0; r < 16; ++r) dirv[r] = enc_register(r, half, kDirReal, vx, vy, vz);
        if constexpr (TRAIN) {
            const char* const xf = reinterpret_cast<const char*>(a.ws_xf + chunk_id * (int64_t)((kDirPad / 8) * 256));
#pragma unroll
            for (int q = 0; q < 4; ++q) tile_store(xf, lane_off, q, f32x4{dirv[4 * q], dirv[4 * q + 1], dirv[4 * q + 2], dirv[4 * q + 3]});
        }
        uint32_t dh[8], dm[8];
        split2_all(dh, dm, [&](int r) { return dirv[r]; });
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            park[(8 + q) * 64] = __builtin_bit_cast(f32x4, u32x4{dh[4 * q], dh[4 * q + 1], dh[4 * q + 2], dh[4 * q + 3]});
            park[(10 + q) * 64] = __builtin_bit_cast(f32x4, u32x4{dm[4 * q], dm[4 * q + 1], dm[4 * q + 2], dm[4 * q + 3]});
        }
    }
    const float* bias = ltab - L::bias_base;   // index with L::bias_off(layer), L::wsig_off, L::wrgb_off, L::scale_off
    // 1 / s_w of a weight tensor's scale slot, wave-uniform (nnr_layout.h: scale_slot; the accumulators start at s_w bias)
    auto inv_scale = [&](int slot) __attribute__((always_inline)) {
        return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(bias[L::scale_off + 16 + slot])));
    };

    uint32_t* mask_base = nullptr;
    // [chunk][layer][lane][words]; half A owns the low words; register r of a half at bit 31 - (r & 31) of word r >> 5 (gate_append2: NOT the bit order of the other modes' planes)
    if (TRAIN) mask_base = a.ws_mask + ((chunk_id * L::n_mask_layers) * 64 + lane) * L::mask_words;

    uint32_t ph[8 * DT], pm[8 * DT];      // the packed terms of the current layer input (pairs [0, NP): half A, [NP, 2 NP): half B), rewritten in place
    f32x16 accA[HT], accB[HT];           // halves A ([0,D/2)) and B ([D/2,D)) of the layer being computed
    uint32_t mwA[HW], mwB[HW];
    f32x2 keep = {0.f, 0.f};             // the first pair of an octet between its unit and the next one's store
    float mx = 0.f;                      // running maximum of the activations the units make (training: of the current plane, reset per plane)
    float mxa = 0.f;                     // training: ... of all planes of this pass so far
    // a plane is complete (half A in the pass B of its layer, half B in the pass A of the next): its maximum to the workgroup's table
    auto flush_max = [&](int plane) __attribute__((always_inline)) {
        if constexpr (TRAIN) {
            const float m = wave_max_f32(mx);
            if (lane == 0) atomicMax(&wg_max[plane], __float_as_uint(m));
            mxa = fmaxf(mxa, mx);
            mx = 0.f;
        }
    };
    float sg0 = 0.f, sg1 = 0.f;          // density head: this lane's share of w_sigma . h8

    auto init_acc = [&](f32x16(&acc)[HT], int bias_offset) __attribute__((always_inline)) {      // the pack kernel stored s_w bias
        const float* b = bias + bias_offset + 4 * half;
#pragma unroll
        for (int t = 0; t < HT; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 bb = *reinterpret_cast<const f32x4*>(b + 32 * t + 8 * q);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[t][4 * q + i] = bb[i];
            }
    };
    auto clear_mask = [&](uint32_t(&mw)[HW]) __attribute__((always_inline)) {
#pragma unroll
        for (int w = 0; w < HW; ++w) mw[w] = 0;
    };
    auto store_mask = [&](const uint32_t(&mw)[HW], int layer_idx, int hb) __attribute__((always_inline)) {
        if constexpr (TRAIN) {
            uint32_t* m = mask_base + (int64_t)layer_idx * 64 * L::mask_words + hb * HW;
#pragma unroll
            for (int w = 0; w < HW; ++w) m[w] = mw[w];
        }
    };
    // One epilogue unit: pair u of a half (registers 2 u, 2 u + 1 of ACC) -> pair OFFP + u of the packed input of the next layer.
    // INV: 1 / s_w of the layer; PLANE: block (this chunk, octet 0) of the layer's activation plane (training), BLK0: the half's first octet;
    // SIG: hidden 8 -- add the pair's share of the density head (its row sits in the head tables in register order)
#define NNR_FINISH(ACC, OFFP, MW, INV, PLANE, BLK0, SIG)                                                           \
    [&](int u) __attribute__((always_inline)) {                                                                  \
        const int r = 2 * u;                                                                                     \
        float x0, x1;                                                                                            \
        if constexpr (TRAIN) {      /* (x > 0) == (relu(x) != 0): two gate bits appended to the half's mask word (nnr_split2.h) */ \
            unit_fwd_train(ACC[r >> 4][r & 15], ACC[(r + 1) >> 4][(r + 1) & 15], INV, MW[r >> 5], x0, x1, ph[(OFFP) + u], pm[(OFFP) + u], mx); \
            if (u & 1) tile_store(PLANE, lane_off, (BLK0) + (u >> 1), f32x4{keep[0], keep[1], x0, x1});          \
            else keep = f32x2{x0, x1};                                                                           \
        } else {                                                                                                 \
            unit_fwd_infer(ACC[r >> 4][r & 15], ACC[(r + 1) >> 4][(r + 1) & 15], INV, x0, x1, ph[(OFFP) + u], pm[(OFFP) + u], mx); \
        }                                                                                                        \
        if constexpr (SIG) {                                                                                     \
            const f32x2 w2 = *reinterpret_cast<const f32x2*>(bias + L::wsig_off + half * (16 * DT) + 2 * (OFFP) + r); \
            sg0 = fmaf(w2[0], x0, sg0);                                                                          \
            sg1 = fmaf(w2[1], x1, sg1);                                                                          \
        }                                                                                                        \
    }
    auto p0 = [&](int part) { return L::fwd_panel0(part); };
    auto xh = [&](int hidden_idx /*0..7*/) -> const char* {      // block (this chunk, octet 0) of hidden layer hidden_idx + 1's activation plane
        if constexpr (!TRAIN) return nullptr;
        else return reinterpret_cast<const char*>(a.ws_xh + (int64_t)hidden_idx * a.S_pad * D + chunk_id * (int64_t)((D / 8) * 256));
    };
    constexpr int SE = TRAIN ? 2 : 0;      // every second unit stores an octet

    // ---- hidden 1: 63 -> D, input = posenc.  Pass A, then pass B with A's epilogue hidden under it. ----
    init_acc(accA, L::bias_off(0));
    gemm_part2<2, HT>(accA, eh, em, pipe, p0(F_L1A));
    init_acc(accB, L::bias_off(0) + L::Dh);
    clear_mask(mwA);
    {
        const float inv = inv_scale(0);
        const char* const pl = xh(0);
        gemm_part2<2, HT, NP, 0, NP / 4, SE, 0>(accB, eh, em, pipe, p0(F_L1B), NNR_FINISH(accA, 0, mwA, inv, pl, 0, false));
    }
    store_mask(mwA, 0, 0);
    // posenc is needed again only by the skip layer: its terms wait in LDS
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        park[q * 64] = __builtin_bit_cast(f32x4, u32x4{eh[4 * q], eh[4 * q + 1], eh[4 * q + 2], eh[4 * q + 3]});
        park[(4 + q) * 64] = __builtin_bit_cast(f32x4, u32x4{em[4 * q], em[4 * q + 1], em[4 * q + 2], em[4 * q + 3]});
    }
    // Invariant from here on: pairs [0, NP) hold half A of the newest layer, accB holds its half B still to be finished.

    // one D -> D ReLU layer (state_dict index `li`, previous layer index li - 1), packed at panel pa; SIG: this layer is hidden 8
    // (pre: the part before this one stashed -- not so behind the skip layer's encoding part)
    auto dense_layer = [&](int li, int pa, auto sig_tag, bool pre) __attribute__((always_inline)) {
        constexpr bool SIG = decltype(sig_tag)::value;
        init_acc(accA, L::bias_off(li));
        clear_mask(mwB);
        {   // pass A: its first half of rows only reads pairs [0, NP); the previous layer's half B is finished meanwhile
            const float inv = inv_scale(li - 1);
            const char* const pl = xh(li - 1);
            gemm_part2<DT, HT, NP, 1, 0, SE, kPreB>(accA, ph, pm, pipe, pa, NNR_FINISH(accB, NP, mwB, inv, pl, HR / 4, false), pre);
        }
        store_mask(mwB, li - 1, 1);
        flush_max(li - 1);
        init_acc(accB, L::bias_off(li) + L::Dh);
        clear_mask(mwA);
        {   // pass B: half A of the new layer replaces pairs [0, NP) in place, one row behind the reads
            const float inv = inv_scale(li);
            const char* const pl = xh(li);
            gemm_part2<DT, HT, NP, 2, 0, SE, kPreA>(accB, ph, pm, pipe, pa + PP, NNR_FINISH(accA, 0, mwA, inv, pl, 0, SIG));
        }
        store_mask(mwA, li, 0);
    };
    // hidden 2..4
#pragma unroll 1
    for (int l = 0; l < 3; ++l) dense_layer(1 + l, p0(F_L2A) + 2 * PP * l, std::false_type{}, true);
    // hidden 5: [h4 ; e] -> D   (skip connection, input order [h, posenc]: model/official_nerf.py:63)
    init_acc(accA, L::bias_off(4));
    clear_mask(mwB);
    {
        const float inv = inv_scale(3);
        const char* const pl = xh(3);
        gemm_part2<DT, HT, NP, 1, 0, SE, kPreB>(accA, ph, pm, pipe, p0(F_L5HA), NNR_FINISH(accB, NP, mwB, inv, pl, HR / 4, false));
    }
    store_mask(mwB, 3, 1);
    flush_max(3);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const u32x4 vh = __builtin_bit_cast(u32x4, park[q * 64]), vm = __builtin_bit_cast(u32x4, park[(4 + q) * 64]);
#pragma unroll
        for (int i = 0; i < 4; ++i) { eh[4 * q + i] = vh[i]; em[4 * q + i] = vm[i]; }
    }
    gemm_part2<2, HT>(accA, eh, em, pipe, p0(F_L5EA));
    init_acc(accB, L::bias_off(4) + L::Dh);
    clear_mask(mwA);
    {
        const float inv = inv_scale(4);
        const char* const pl = xh(4);
        gemm_part2<DT, HT, NP, 2, 0, SE, 0>(accB, ph, pm, pipe, p0(F_L5HB), NNR_FINISH(accA, 0, mwA, inv, pl, 0, false));
    }
    gemm_part2<2, HT>(accB, eh, em, pipe, p0(F_L5EB));
    store_mask(mwA, 4, 0);
    // hidden 6, 7, 8
#pragma unroll 1
    for (int l = 0; l < 2; ++l) dense_layer(5 + l, p0(F_L6A) + 2 * PP * l, std::false_type{}, l > 0);
    dense_layer(7, p0(F_L6A) + 2 * PP * 2, std::true_type{}, true);

    // colour hidden: g = relu(W' h8 + Wg[:, D:] gamma_4(v) + b'), W' = Wg[:, :D] Wf (the feature layer folded in by the pack kernel, nnr_layout.h);
    // both parts share scale slot 8.  Its side work finishes hidden 8 (half B) incl. the density head's other half.
    init_acc(accA, L::bias_off(10));
    clear_mask(mwB);
    {
        const float inv = inv_scale(7);
        const char* const pl = xh(7);
        gemm_part2<DT, HT, NP, 1, 0, SE, kPreB>(accA, ph, pm, pipe, p0(F_RGBH_F), NNR_FINISH(accB, NP, mwB, inv, pl, HR / 4, true));
    }
    store_mask(mwB, 7, 1);
    flush_max(7);
    const float sg = sg0 + sg1;
    const float sigma_raw = sg + __shfl_xor(sg, 32, 64) + bias[L::bias_off(8)];
    {
        uint32_t dh[8], dm[8];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const u32x4 vh = __builtin_bit_cast(u32x4, park[(8 + q) * 64]), vm = __builtin_bit_cast(u32x4, park[(10 + q) * 64]);
#pragma unroll
            for (int i = 0; i < 4; ++i) { dh[4 * q + i] = vh[i]; dm[4 * q + i] = vm[i]; }
        }
        gemm_part2<1, HT>(accA, dh, dm, pipe, p0(F_RGBH_D));
    }
    float g[HR];      // the colour-hidden activations stay fp32: the rgb head is a per-lane dot product
    clear_mask(mwA);
    {
        const float inv = inv_scale(8);
#pragma unroll
        for (int r = 0; r < HR; r += 2) {
            g[r] = relu1(accA[r >> 4][r & 15] * inv);
            g[r + 1] = relu1(accA[(r + 1) >> 4][(r + 1) & 15] * inv);
            if constexpr (TRAIN) gate_append2(mwA[r >> 5], g[r], g[r + 1]);
        }
    }
    store_mask(mwA, 8, 0);
    if constexpr (TRAIN) {      // registers 4 q .. 4 q + 3 = features 8 q + 4 half + {0..3}: octet q of the plane, one whole block per store
        const char* const xg = reinterpret_cast<const char*>(a.ws_xg + chunk_id * (int64_t)((D / 16) * 256));
#pragma unroll
        for (int q = 0; q < HR / 4; ++q) tile_store(xg, lane_off, q, f32x4{g[4 * q], g[4 * q + 1], g[4 * q + 2], g[4 * q + 3]});
    }
    // rgb head: 3 per-lane dot products over the lane's half of g, halves combined by one shuffle, then sigmoid
    float rgbv[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float* wc = bias + L::wrgb_off + (2 * c + half) * HR;
        float acc0 = 0.f, acc1 = 0.f;
#pragma unroll
        for (int q = 0; q < HR / 4; ++q) {
            const f32x4 w4 = *reinterpret_cast<const f32x4*>(wc + 4 * q);
            acc0 = fmaf(w4[0], g[4 * q], acc0);
            acc1 = fmaf(w4[1], g[4 * q + 1], acc1);
            acc0 = fmaf(w4[2], g[4 * q + 2], acc0);
            acc1 = fmaf(w4[3], g[4 * q + 3], acc1);
        }
        const float part = acc0 + acc1;
        rgbv[c] = part + __shfl_xor(part, 32, 64);
    }
    {
        const float* b = bias + L::bias_off(11);
        f32x4 o;
        o[0] = sigmoid_ref(rgbv[0] + b[0]);
        o[1] = sigmoid_ref(rgbv[1] + b[1]);
        o[2] = sigmoid_ref(rgbv[2] + b[2]);
        o[3] = sigma_raw;
        {   // The one bound of this arithmetic (include/nnr.h, NNR_F_SPLIT2): a hidden activation that rounds to inf in fp16 (>= 65520).  Its terms are
            // inf / -inf, the next layer's products NaN -- and ReLU (v_max_f32 returns the operand that is a number) turns those into ZEROS: left
            // alone the sample would come out finite and wrong.  A sample that saw such an activation (either of its two lanes) is made NaN here:
            // the loss is NaN, the caller's check fires (model/losses.py:204-205), an inference frame shows it.
            const float seen = TRAIN ? fmaxf(mxa, mx) : mx;
            const bool over = !(fmaxf(seen, __shfl_xor(seen, 32, 64)) < 65520.f);
            const float qnan = __uint_as_float(0x7fc00000u);
#pragma unroll
            for (int c = 0; c < 4; ++c) o[c] = over ? qnan : o[c];
        }
        if (!fuse) {
            if (half == 0 && s < a.S) *reinterpret_cast<f32x4*>(a.ws_out4 + 4 * s) = o;
        } else if constexpr (!TRAIN) {
            // model/rendering.py:119-132,145-147 for the 32 samples of this chunk, as in nnr_mlp_fwd.hip (= composite_fwd_kernel 64 at a time)
            const int jn = j + 1;
            float zn = 0.f;
            if (jn < a.N) {
                const float lo1 = a.z_lo[jn], hi1 = a.z_hi[jn];
                zn = a.jitter ? __fadd_rn(lo1, __fmul_rn(__fsub_rn(hi1, lo1), a.jitter[sc + 1])) : lo1;
            }
            float unused;
            const float alpha = half == 0 ? sample_alpha(o[3], jn < a.N ? zn - z : 1e10f, jn == a.N, a.flags, unused) : 0.f;
            const float incl = wave_scan_mul(half == 0 ? (1.f - alpha) + kEpsT : 1.f, lane);
            float excl = __shfl_up(incl, 1, 64);
            if (lane == 0) excl = 1.f;
            const float w = alpha * cT * excl;
            cT *= __shfl(incl, 31, 64);
            cr += w * o[0]; cg += w * o[1]; cb += w * o[2]; cz += w * z; cw += w;
            if (pass + 1 == n_pass) {
                const float sr = wave_sum(cr), sgn = wave_sum(cg), sb = wave_sum(cb), sz = wave_sum(cz), sw = wave_sum(cw);
                if (lane == 0) {
                    const float bg = (a.flags & kFlagWhiteBg) ? 1.f - sw : 0.f;
                    float* out = a.fuse_rgb + 3 * (int64_t)ray;
                    out[0] = sr + bg; out[1] = sgn + bg; out[2] = sb + bg;
                    a.fuse_dist[ray] = sz;
                }
            }
        }
    }
#undef NNR_FINISH
    pipe.next_pass(pass + 2 < n_pass);
    }   // pass
    if constexpr (TRAIN) {      // the workgroup's maxima to the launch's table (NNR_F_SPLIT2: the weight-gradient kernel's scales)
        __syncthreads();
        if (a.plane_max != nullptr && threadIdx.x < 8) atomicMax(reinterpret_cast<uint32_t*>(a.plane_max) + threadIdx.x, wg_max[threadIdx.x]);
        if (a.plane_max != nullptr && threadIdx.x == 8) atomicMax(reinterpret_cast<uint32_t*>(a.plane_max) + 17, wg_max[8]);      // (kPlaneMaxEnc, nnr_wgrad.hip)
    }
}

// One (D, TRAIN) instantiation per translation unit (csrc/build.py: -DNNR_FWD_D=.. -DNNR_FWD_TRAIN=..), as for nnr_mlp_fwd.hip
#ifdef NNR_FWD_D
template <>
hipError_t launch_mlp_fwd_variant<NNR_FWD_D, (NNR_FWD_TRAIN != 0), 3>(const MlpFwdArgs& a, hipStream_t st) {
    dim3 grid((unsigned)(a.chunks_per_ray > 0 ? a.S_pad / kBlockSamples / a.chunks_per_ray : a.S_pad / kBlockSamples)), block(256);
    constexpr bool train = NNR_FWD_TRAIN != 0;
    prof_before(train ? PROF_FWD_TRAIN : PROF_FWD_INFER, st);
    hipLaunchKernelGGL((mlp_fwd_f16_kernel<NNR_FWD_D, train>), grid, block, 0, st, a);
    prof_after(train ? PROF_FWD_TRAIN : PROF_FWD_INFER, st);
    return hipGetLastError();
}
#endif

}  // namespace nnr
