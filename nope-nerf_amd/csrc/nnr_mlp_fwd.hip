// nnr_mlp_fwd.hip -- fused NeRF MLP forward for gfx950: sampling + positional encoding + the MLP, per-sample
// (rgb, sigma_raw) out.  Restates, per sample: model/rendering.py:184-195 (z, points, view dir) and
// model/official_nerf.py:60-96 (the MLP).  One wave = 32 samples; activations stay in VGPRs between layers as MFMA
// B-operands (see nnr_layout.h); weights arrive as pre-packed A fragments through a DMA-fed LDS ring shared by the 4 waves.
//
// Roofline: MFMA-bound.  Algorithmic work 593 408 MACs/sample at D=256; executed 528 000: the feature layer is folded into
// the colour-hidden layer (nnr_layout.h), leaving 8 448 v_mfma_f32_32x32x2_f32 per 32 samples (1.5 % padding of the
// 63 / 27-wide edges) = 541 k cycles per wave against ~40-90 k cycles of everything else.
// Every D-wide layer runs as two half-output passes so that epilogues execute inside the MFMA stream (nnr_device.h).
// HBM per sample: 4 B jitter in, 20 B out (+ the 9.4 KB activation stash when training, written once, never re-read here).
// (fp32 products; the bf16-MFMA mode has its own kernel, nnr_mlp_fwd_bf16.hip)
// MODE 2 (NNR_F_SPLIT3): the same kernel with every GEMM part's products taken as six bf16 MFMA terms -- only the weight stream (24 KiB
// panels of pre-split fragments) and gemm_part (nnr_split.h) differ; 2.7 times fewer matrix-pipe cycles, fp32-equivalent results.
#include "nnr_device.h"
#include "nnr_kernels.h"
#include "nnr_split.h"

namespace nnr {

constexpr bool kAblateNoMask = false;

NNR_TL_DECL(tl_fwd)

template <int D, bool TRAIN, int MODE>
__global__ __launch_bounds__(256, 1) void mlp_fwd_kernel(MlpFwdArgs a) {
    NNR_STAMP(tl_fwd, (TRAIN ? 0 : 16) + 0);
    using L = Layout<D, MODE>;
    constexpr bool kTileX = MODE == 2 && TRAIN && kTileActPlanes;      // tile-major activation planes, non-temporal whole-block stores (nnr_layout.h)
    using Pipe = PanelPipeT<kWavesPerBlock, mode_panel_frags(MODE), kTileX>;
    constexpr int kRingF4 = kNBuf * Pipe::F4;
    constexpr int DT = L::DT, HT = L::HT;
    const int lane0 = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;

    // ---- weight panels (LDS ring, DMA two panels ahead) and the bias / head tables, all in ONE __shared__ array. ----
    constexpr int kPark = kWavesPerBlock * 12 * 64;   // per wave 12 float4 slots per lane: posenc (8) + direnc (4)
    __shared__ __attribute__((aligned(16))) f32x4 smem[kRingF4 + kPark + (L::table_floats + 3) / 4];
    float* const ltab = reinterpret_cast<float*>(smem + kRingF4 + kPark);
    for (int i = threadIdx.x; i < L::table_floats; i += 256) ltab[i] = a.packed[L::bias_base + i];
    __syncthreads();   // before any DMA is in flight: this is the only full barrier of the kernel
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    Pipe pipe{reinterpret_cast<const f32x4*>(a.packed) + wave_u * (Pipe::PW * 64), smem, wave_u, lane0, L::fwd_panels};
    // Work decomposition.  Flat (a.chunks_per_ray == 0): workgroup b takes the samples [128 b, 128 b + 128), wave w the 32 from
    // 32 w on -- one pass over the weight stream per workgroup.  Ray mode (N a multiple of 32, R a multiple of 4): workgroup b takes
    // the rays 4 b .. 4 b + 3, wave w ray 4 b + w, and walks its N / 32 chunks one after the other; the weight stream wraps around
    // (PanelPipe), so only the first chunk of a workgroup waits for its first panels, the tables are copied once, and the grid is
    // R / 4 workgroups -- one per CU at 1024 rays, no tail.  Sample numbering, planes and masks are the same in both modes.
    const int n_pass = a.chunks_per_ray > 0 ? a.chunks_per_ray : 1;
    pipe.more = n_pass > 1;
    pipe.start();
    // fused compositing (inference, ray mode): running transmittance and weighted sums of this wave's ray, carried across its chunks
    float cT = 1.f, cr = 0.f, cg = 0.f, cb = 0.f, cz = 0.f, cw = 0.f;
#pragma unroll 1
    for (int pass = 0; pass < n_pass; ++pass) {
    // Everything lane-dependent is derived INSIDE the pass from an opaque copy of the lane id: left to itself the compiler hoists the
    // dozens of lane-constant addresses and table pointers out of the pass loop and then spills them around the MFMA stream -- and a
    // spill reload is a VMEM load whose wait (vmcnt) drains the weight DMA queue.
    int lane = lane0;
    asm volatile("" : "+v"(lane));
    pipe.lane = lane;
    const int half = lane >> 5;
    const int col = lane & 31;
    f32x4* const park = smem + kRingF4 + wave * (12 * 64) + lane;
    const int64_t chunk_id = a.chunks_per_ray > 0 ? ((int64_t)blockIdx.x * kWavesPerBlock + wave) * n_pass + pass
                                                   : (int64_t)blockIdx.x * kWavesPerBlock + wave;
    const int64_t s = chunk_id * kChunk + col;                                     // this lane's sample
    const int64_t sc = s < a.S ? s : a.S - 1;                                      // clamp: padded samples recompute the last one
    const int64_t ss = s;         // row of the stash planes
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
    const bool fuse = !TRAIN && a.fuse_rgb != nullptr;
    if (half == 0 && s < a.S && !fuse) a.ws_z[s] = z;
    // ---- encodings, straight into fragment layout ----
    float e[32];  // gamma_10(p): 63 -> 64
#pragma unroll
    for (int r = 0; r < 32; ++r) e[r] = enc_register(r, half, kPosReal, px, py, pz);
    float dirv[16];  // gamma_4(v): 27 -> 32
#pragma unroll
    for (int r = 0; r < 16; ++r) dirv[r] = enc_register(r, half, kDirReal, vx, vy, vz);
    NNR_STAMP(tl_fwd, (TRAIN ? 0 : 16) + 1);
    NNR_STAMP(tl_fwd, (TRAIN ? 0 : 16) + 2);

    // the direction encoding is needed once, at the very end: park it in LDS instead of holding 16 registers for 70 panels
#pragma unroll
    for (int q = 0; q < 4; ++q) park[(8 + q) * 64] = f32x4{dirv[4 * q], dirv[4 * q + 1], dirv[4 * q + 2], dirv[4 * q + 3]};
    const float* bias = ltab - L::bias_base;   // index with L::bias_off(layer), L::wsig_off, L::wrgb_off

    uint32_t* mask_base = nullptr;
    if (TRAIN) {
        // masks: [chunk][layer][lane][mask_words]; half A of a layer owns the low words, half B the high words
        mask_base = a.ws_mask + ((chunk_id * L::n_mask_layers) * 64 + lane) * L::mask_words;
    }
    constexpr int HR = 16 * HT;              // registers of half a layer's outputs
    constexpr int NP = HR / 2;               // register pairs per half (the unit of hidden epilogue work)
    constexpr int HW = (HR + 31) / 32;       // mask words per half
    constexpr int PP = mode_panels(DT, HT, MODE);  // panels of one D x D/2 pass

    float h[16 * DT];    // current layer input (activations of the previous layer), rewritten in place
    f32x16 accA[HT], accB[HT];   // halves A ([0,D/2)) and B ([D/2,D)) of the layer being computed
    uint32_t mwA[HW], mwB[HW];

    // accumulators start at the bias, so an epilogue is only ReLU (+ sign bit) or a move
    auto init_acc = [&](f32x16(&acc)[HT], int bias_offset) __attribute__((always_inline)) {
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
    constexpr bool kGateSplit = MODE == 2 && TRAIN && !kAblateNoMask;     // see NNR_RELU_PAIR below
    auto store_gates = [&](int layer_idx) __attribute__((always_inline)) {      // both halves of a hidden layer's gates, from the pipe
        if constexpr (kGateSplit) {
            uint32_t* m = mask_base + (int64_t)layer_idx * 64 * L::mask_words;
#pragma unroll
            for (int w = 0; w < L::mask_words; ++w) m[w] = pipe.gw[w];
        }
    };
    auto store_mask = [&](const uint32_t(&mw)[HW], int layer_idx, int hb) __attribute__((always_inline)) {
        if (TRAIN && !(kGateSplit && layer_idx < 8)) {
            uint32_t* m = mask_base + (int64_t)layer_idx * 64 * L::mask_words + hb * HW;
#pragma unroll
            for (int w = 0; w < HW; ++w) m[w] = mw[w];
        }
    };
// one epilogue unit u (registers 2u, 2u+1 of the half): h[off + 2u + i] = relu(acc) (+ sign bit) or plain move.  Mask bit r = (x > 0).
    // Three-term training mode: the gates of the hidden layers come out of the NEXT layer's split (gate_pair, nnr_split.h) -- two
    // instructions per value less in the epilogue units, and the units' mask words no longer live across the passes.  GATE: this call site
    // still makes its own (the colour-hidden layer, whose output no GEMM part splits).
#define NNR_RELU_PAIR(ACC, OFF, MW) NNR_RELU_PAIR_(ACC, OFF, MW, !kGateSplit)
#define NNR_RELU_PAIR_(ACC, OFF, MW, GATE)                                                                   \
    [&](int u) __attribute__((always_inline)) {                                                              \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                      \
            const int r = 2 * u + i;                                                                         \
            const float x = ACC[r >> 4][r & 15];                                                             \
            h[(OFF) + r] = relu1(x);                                                                         \
            /* (x > 0) == (relu(x) != 0): as integers, min(bits, 1) shifted into place -- two instructions  */ \
            if (TRAIN && !kAblateNoMask && (GATE)) MW[r >> 5] |= min(__float_as_uint(h[(OFF) + r]), 1u) << (r & 31); \
        }                                                                                                    \
    }
#define NNR_MOVE_PAIR(ACC, OFF)                                                                              \
    [&](int u) __attribute__((always_inline)) {                                                              \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) h[(OFF) + 2 * u + i] = ACC[(2 * u + i) >> 4][(2 * u + i) & 15]; \
    }
    auto p0 = [&](int part) { return L::fwd_panel0(part); };
    // stash destinations.  Row-major: this lane's row + its half's four columns; tile-major (kTileX): block (chunk, octet 0) of the plane +
    // 16 bytes per lane -- gemm_part adds 1 KiB per stash store
    float* const xe = !TRAIN ? nullptr : kTileX ? a.ws_xe + chunk_id * (int64_t)((kPosPad / 8) * 256) + 4 * lane : a.ws_xe + ss * kPosPad + 4 * half;
    auto xh = [&](int hidden_idx /*0..7*/) -> float* {
        if constexpr (!TRAIN) return nullptr;
        else if constexpr (kTileX) return a.ws_xh + (int64_t)hidden_idx * a.S_pad * D + chunk_id * (int64_t)((D / 8) * 256) + 4 * lane;
        else return a.ws_xh + ((int64_t)hidden_idx * a.S_pad + ss) * D + 4 * half;
    };

    // ---- hidden 1: 63 -> D, input = posenc.  Pass A, then pass B with A's epilogue hidden under it. ----
    init_acc(accA, L::bias_off(0));
    gemm_part<2, HT, TRAIN>(accA, e, pipe, p0(F_L1A), xe);
    init_acc(accB, L::bias_off(0) + L::Dh);
    clear_mask(mwA);
    gemm_part<2, HT, false, NP, NP / 8, 0>(accB, e, pipe, p0(F_L1B), nullptr, NNR_RELU_PAIR(accA, 0, mwA));
    store_mask(mwA, 0, 0);
    // posenc is needed again only by the skip layer: park it in LDS meanwhile
#pragma unroll
    for (int q = 0; q < 8; ++q) park[q * 64] = f32x4{e[4 * q], e[4 * q + 1], e[4 * q + 2], e[4 * q + 3]};
    NNR_STAMP(tl_fwd, (TRAIN ? 0 : 16) + 3);
    // Invariant from here on: h[0, HR) holds half A of the newest layer, accB holds its half B still to be finished.

    // one D -> D ReLU layer (state_dict index `li`, previous layer index li-1), packed at panel pa
    auto dense_layer = [&](int li, int pa, float* stash) __attribute__((always_inline)) {
        init_acc(accA, L::bias_off(li));
        clear_mask(mwB);
        // pass A: the first half of the k-groups only needs h[0,HR); the previous layer's half B is finished meanwhile
        pipe.gates_on = kGateSplit;
        gemm_part<DT, HT, TRAIN, NP, 2, 0>(accA, h, pipe, pa, stash, NNR_RELU_PAIR(accB, HR, mwB));
        pipe.gates_on = false;
        store_gates(li - 1);
        store_mask(mwB, li - 1, 1);
        init_acc(accB, L::bias_off(li) + L::Dh);
        clear_mask(mwA);
        // pass B: half A of the new layer replaces h[0,HR) in place, one k-group behind the reads
        pipe.part_pre = TRAIN && MODE == 2 && PP >= 2 ? 6 : 0;   // pass A's last rows stashed (nnr_split.h)
        gemm_part<DT, HT, false, NP, 2, 1>(accB, h, pipe, pa + PP, nullptr, NNR_RELU_PAIR(accA, 0, mwA));
        pipe.part_pre = 0;
        store_mask(mwA, li, 0);
    };
    // hidden 2..4
#pragma unroll 1
    for (int l = 0; l < 3; ++l) dense_layer(1 + l, p0(F_L2A) + 2 * PP * l, xh(l));
    NNR_STAMP(tl_fwd, (TRAIN ? 0 : 16) + 4);
    // hidden 5: [h4 ; e] -> D   (skip connection, input order [h, posenc]: model/official_nerf.py:63)
    init_acc(accA, L::bias_off(4));
    clear_mask(mwB);
    pipe.gates_on = kGateSplit;
    gemm_part<DT, HT, TRAIN, NP, 2, 0>(accA, h, pipe, p0(F_L5HA), xh(3), NNR_RELU_PAIR(accB, HR, mwB));
    pipe.gates_on = false;
    store_gates(3);
    store_mask(mwB, 3, 1);
    float e5[32];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const f32x4 v = park[q * 64];
#pragma unroll
        for (int i = 0; i < 4; ++i) e5[4 * q + i] = v[i];
    }
    gemm_part<2, HT>(accA, e5, pipe, p0(F_L5EA));
    init_acc(accB, L::bias_off(4) + L::Dh);
    clear_mask(mwA);
    gemm_part<DT, HT, false, NP, 2, 1>(accB, h, pipe, p0(F_L5HB), nullptr, NNR_RELU_PAIR(accA, 0, mwA));
    gemm_part<2, HT>(accB, e5, pipe, p0(F_L5EB));
    store_mask(mwA, 4, 0);
    NNR_STAMP(tl_fwd, (TRAIN ? 0 : 16) + 5);
    // hidden 6..8
#pragma unroll 1
    for (int l = 0; l < 3; ++l) dense_layer(5 + l, p0(F_L6A) + 2 * PP * l, xh(4 + l));
    NNR_STAMP(tl_fwd, (TRAIN ? 0 : 16) + 6);

    // colour hidden: g = relu(W' h8 + Wg[:, D:] gamma_4(v) + b'), W' = Wg[:, :D] Wf -- the feature layer (no activation,
    // model/official_nerf.py:87-89) is folded into this one by the pack kernel, see nnr_layout.h.  One pass (D/2 outputs).
    // Its side work first finishes hidden 8 (half B), then evaluates the density head -- a per-lane dot product of h8 with
    // the density row (a 1-row GEMM is not MFMA work).
    float* const xf = !TRAIN ? nullptr : kTileX ? a.ws_xf + chunk_id * (int64_t)((kDirPad / 8) * 256) + 4 * lane : a.ws_xf + ss * kDirPad + 4 * half;
    init_acc(accA, L::bias_off(10));
    clear_mask(mwB);
    float sg0 = 0.f, sg1 = 0.f;
    {
        const float* wsg = bias + L::wsig_off + half * (16 * DT);
        auto finish_then_sigma = [&](int u) __attribute__((always_inline)) {
            if (u < NP) {
                NNR_RELU_PAIR(accB, HR, mwB)(u);
            } else {   // pair u - NP of all 16*DT registers of h8; every register is final by now (see PPG below)
                const int v = u - NP;
                const f32x2 w2 = *reinterpret_cast<const f32x2*>(wsg + 2 * v);
                sg0 = fmaf(w2[0], h[2 * v], sg0);
                sg1 = fmaf(w2[1], h[2 * v + 1], sg1);
            }
        };
        // 4 units per k-group: the NP finishing units occupy k-groups [0, NP/4) -- long before k-group 2*DT first reads
        // h[HR..] -- and the 2*NP density units k-groups [NP/4, 3*NP/4) <= 4*DT
        pipe.gates_on = kGateSplit;
        gemm_part<DT, HT, TRAIN, 3 * NP, 4, 0>(accA, h, pipe, p0(F_RGBH_F), xh(7), finish_then_sigma);
        pipe.gates_on = false;
    }
    store_gates(7);
    store_mask(mwB, 7, 1);
    const float sg = sg0 + sg1;
    const float sigma_raw = sg + __shfl_xor(sg, 32, 64) + bias[L::bias_off(8)];
    NNR_STAMP(tl_fwd, (TRAIN ? 0 : 16) + 7);
    float dir2[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 v = park[(8 + q) * 64];
#pragma unroll
        for (int i = 0; i < 4; ++i) dir2[4 * q + i] = v[i];
    }
    gemm_part<1, HT, TRAIN>(accA, dir2, pipe, p0(F_RGBH_D), xf);
    NNR_STAMP(tl_fwd, (TRAIN ? 0 : 16) + 8);
    clear_mask(mwA);
#pragma unroll
    for (int u = 0; u < NP; ++u) NNR_RELU_PAIR_(accA, 0, mwA, true)(u);   // g = h[0, HR)
    store_mask(mwA, 8, 0);
    if (TRAIN) {
        if constexpr (kTileX) {      // registers 4 q .. 4 q + 3 = features 8 q + 4 half + {0..3}: octet q of the plane, one whole block per store
            typedef __attribute__((address_space(1))) f32x4 glb_f32x4;
            glb_f32x4* const xg = (glb_f32x4*)(a.ws_xg + chunk_id * (int64_t)((D / 16) * 256) + 4 * lane);
#pragma unroll
            for (int q = 0; q < HR / 4; ++q) __builtin_nontemporal_store(f32x4{h[4 * q], h[4 * q + 1], h[4 * q + 2], h[4 * q + 3]}, xg + 64 * q);
        } else {
            float* xg = a.ws_xg + ss * (D / 2) + 4 * half;
#pragma unroll
            for (int q = 0; q < HR / 4; ++q)
                *reinterpret_cast<f32x4*>(xg + 8 * q) = f32x4{h[4 * q], h[4 * q + 1], h[4 * q + 2], h[4 * q + 3]};
        }
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
            acc0 = fmaf(w4[0], h[4 * q], acc0);
            acc1 = fmaf(w4[1], h[4 * q + 1], acc1);
            acc0 = fmaf(w4[2], h[4 * q + 2], acc0);
            acc1 = fmaf(w4[3], h[4 * q + 3], acc1);
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
        if (!fuse) {
            if (half == 0 && s < a.S) *reinterpret_cast<f32x4*>(a.ws_out4 + 4 * s) = o;
        } else if constexpr (!TRAIN) {
            // model/rendering.py:119-132,145-147 for the 32 samples of this chunk (lanes 0..31; the upper half is neutral), exactly as
            // composite_fwd_kernel does it 64 at a time: alpha from the density and the distance to the NEXT sample (whose z this
            // lane recomputes from the interval table), transmittance by a product scan across the lanes on top of the carry
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
                const float sr = wave_sum(cr), sg = wave_sum(cg), sb = wave_sum(cb), sz = wave_sum(cz), sw = wave_sum(cw);
                if (lane == 0) {
                    const float bg = (a.flags & kFlagWhiteBg) ? 1.f - sw : 0.f;
                    float* out = a.fuse_rgb + 3 * (int64_t)ray;
                    out[0] = sr + bg; out[1] = sg + bg; out[2] = sb + bg;
                    a.fuse_dist[ray] = sz;
                }
            }
        }
    }
    NNR_STAMP(tl_fwd, (TRAIN ? 0 : 16) + 9);
#undef NNR_RELU_PAIR
#undef NNR_RELU_PAIR_
#undef NNR_MOVE_PAIR
    pipe.next_pass(pass + 2 < n_pass);
    }   // pass
}

#if defined(NNR_TIMELINE) && defined(NNR_FWD_D) && NNR_FWD_D == 256 && NNR_FWD_TRAIN && !defined(NNR_FWD_MODE)
extern "C" int nnr_timeline_fwd(unsigned long long* host32) {
    return (int)hipMemcpyFromSymbol(host32, HIP_SYMBOL(tl_fwd), 32 * sizeof(unsigned long long));
}
#endif

// One (D, TRAIN) instantiation per translation unit: the kernel is ~8 000 MFMAs of straight-line code and hipcc spends minutes on each;
// csrc/build.py compiles this file once per variant (-DNNR_FWD_D=.. -DNNR_FWD_TRAIN=..) in parallel and once without the macros
// for the dispatcher below.
#ifdef NNR_FWD_D
#ifndef NNR_FWD_MODE
#define NNR_FWD_MODE 0
#endif
template <>
hipError_t launch_mlp_fwd_variant<NNR_FWD_D, (NNR_FWD_TRAIN != 0), NNR_FWD_MODE>(const MlpFwdArgs& a, hipStream_t st) {
    // ray mode: one workgroup per 4 rays, chunks_per_ray passes each; flat mode: one workgroup per 128 samples
    dim3 grid((unsigned)(a.chunks_per_ray > 0 ? a.S_pad / kBlockSamples / a.chunks_per_ray : a.S_pad / kBlockSamples)), block(256);
    constexpr bool train = NNR_FWD_TRAIN != 0;
    prof_before(train ? PROF_FWD_TRAIN : PROF_FWD_INFER, st);
    hipLaunchKernelGGL((mlp_fwd_kernel<NNR_FWD_D, train, NNR_FWD_MODE>), grid, block, 0, st, a);
    prof_after(train ? PROF_FWD_TRAIN : PROF_FWD_INFER, st);
    return hipGetLastError();
}
#else
hipError_t launch_mlp_fwd(int D, const MlpFwdArgs& a, bool train, hipStream_t st, int mode) {
    if (mode == 3) {      // nnr_mlp_fwd_f16.hip
        if (D == 256) return train ? launch_mlp_fwd_variant<256, true, 3>(a, st) : launch_mlp_fwd_variant<256, false, 3>(a, st);
        return train ? launch_mlp_fwd_variant<128, true, 3>(a, st) : launch_mlp_fwd_variant<128, false, 3>(a, st);
    }
    if (mode == 2) {
        if (D == 256) return train ? launch_mlp_fwd_variant<256, true, 2>(a, st) : launch_mlp_fwd_variant<256, false, 2>(a, st);
        return train ? launch_mlp_fwd_variant<128, true, 2>(a, st) : launch_mlp_fwd_variant<128, false, 2>(a, st);
    }
    if (D == 256) return train ? launch_mlp_fwd_variant<256, true>(a, st) : launch_mlp_fwd_variant<256, false>(a, st);
    return train ? launch_mlp_fwd_variant<128, true>(a, st) : launch_mlp_fwd_variant<128, false>(a, st);
}
#endif

}  // namespace nnr
