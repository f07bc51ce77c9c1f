// nnr_mlp_dgrad_bf16.hip -- fused input-gradient chain of the NeRF MLP with bf16 MFMA products (NNR_F_BF16), gfx950.  Same function
// as nnr_mlp_dgrad.hip (autograd's chain through model/official_nerf.py:60-96 for the data path: d rgb_pre, d sigma_raw per sample
// down to d point, d view dir, every layer's pre-activation gradient left in the workspace -- as bf16, tile-major -- for the
// weight-gradient kernel), in the arithmetic of the bf16 mode: every transposed hidden layer is bf16(gradient) x bf16(W^T) with fp32
// accumulation; the two heads' input gradients (3 FMAs / 1 FMA per value) and the chain rule through the encodings stay fp32.
// One wave = TWO chunks of 32 samples, gradients packed bf16 between layers: see nnr_mlp_bf16.h.
#include "nnr_kernels.h"
#include "nnr_mlp_bf16.h"

namespace nnr {

NNR_TL_DECL(tl_dgrad16)

template <int D, int T, int W>
__global__ __launch_bounds__(64 * W, 1) void mlp_dgrad_bf16_kernel(MlpDgradArgs a) {
    constexpr int kTiles = T;            // 32-sample chunks per wave (nnr_mlp_bf16.h)
    static_assert(T * W * kChunk == kWideSamples, "a workgroup covers 256 samples per pass");
    using Pipe = PanelPipeT<W>;
    NNR_STAMP(tl_dgrad16, 0);
    using L = Layout<D, true>;
    constexpr int DT = L::DT, HT = L::HT;
    constexpr int HR = 16 * HT;              // fragment registers of half a layer's outputs (fp32 numbering)
    constexpr int NP = HR / 2;               // packed registers per half and tile = epilogue units per half and tile
    constexpr int HW = (HR + 31) / 32;       // mask words per half
    constexpr int PP = part_panels(DT, HT, true);
    constexpr int NQ = 8 * DT;               // packed registers of a D-wide vector
    constexpr int NU = kTiles * NP;          // epilogue units of one half-output pass
    const int lane0 = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;

    // LDS: the panel ring of the transposed weight stream, the fp32 head tables (density row, rgb rows, register order) and, per wave and
    // tile, two 1 KiB slots the ReLU gates of a layer are DMA'd into one GEMM before they are needed, and two more that hold the samples'
    // positions / view directions for the whole pass (below)
    constexpr bool kMaskLds = L::mask_words == 4;     // D = 256: a lane's gates of a layer (halves A and B) are one 16-byte DMA element
    constexpr int kTabF4 = (L::head_floats + 3) / 4;
    __shared__ __attribute__((aligned(16))) f32x4 smem[kNBuf * kPanelF4 + kTabF4 + (kMaskLds ? W * kTiles * 2 * 64 : 0) + W * kTiles * 2 * 64];
    float* const ltab = reinterpret_cast<float*>(smem + kNBuf * kPanelF4);
    for (int i = threadIdx.x; i < L::head_floats; i += 64 * W) ltab[i] = a.packed[L::head_base + i];
    __syncthreads();   // before any DMA is in flight: the only full barrier of the kernel
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    Pipe pipe{reinterpret_cast<const f32x4*>(a.packed + L::bwd_base) + wave_u * (Pipe::PW * 64), smem, wave_u, lane0, L::bwd_panels};
    // flat or ray-mode decomposition in pairs of chunks, exactly as in mlp_fwd_bf16_kernel
    const int n_pass = a.chunks_per_ray > 0 ? a.chunks_per_ray : 1;
    const int64_t last_chunk = a.S_pad / kChunk - 1;
    pipe.more = n_pass > 1;
    pipe.start();
    auto p0 = [&](int part) { return L::bwd_panel0(part); };
#pragma unroll 1
    for (int pass = 0; pass < n_pass; ++pass) {
    int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));   // re-derived and opaque per pass (mlp_fwd_bf16_kernel)
    asm volatile("" : "+v"(lane));
    pipe.lane = lane;
    auto wsig_ = [&]() { return ltab + (lane_id() >> 5) * (16 * DT); };   // density row, this half's registers
    const float* const wrgb = ltab + 2 * 16 * DT;        // rgb rows: [(2c + half) * HR + r]
    const int64_t pair = a.chunks_per_ray > 0 ? ((int64_t)blockIdx.x * W + wave_u) * n_pass + pass
                                               : (int64_t)blockIdx.x * W + wave_u;
    int chunk[kTiles];   // chunk index of either tile
    f32x4 dout[kTiles];
#pragma unroll
    for (int n = 0; n < kTiles; ++n)      // wave-uniform (block index, wave index, pass counter): lives in scalar registers
        chunk[n] = __builtin_amdgcn_readfirstlane((int)(kTiles * pair + n < last_chunk ? kTiles * pair + n : last_chunk));
    // this lane's sample of tile n; every address is derived from it where it is needed (opaque(), nnr_mlp_bf16.h)
    auto row0 = [&](int n) -> int64_t { return (int64_t)opaque_uniform(chunk[n]) * kChunk; };   // first sample of tile n: wave-uniform (scalar)
    auto sample = [&](int n) -> int64_t { return row0(n) + (lane_id() & 31); };
    // ---- ReLU gates through LDS ----
    // A load into registers that hipcc can see is waited for with vmcnt(0) whenever stores are pending (loads and stores share the
    // counter and may, as far as the compiler knows, retire out of order) -- and in this kernel stores are ALWAYS pending: each of the
    // 18 gate loads of a pass drained the whole stash-store queue with the matrix pipe idle (9 % of the kernel).  The gates of a layer
    // now travel like the weights: one LDS-DMA element per lane and tile (a lane's 16 bytes = halves A and B), issued at the head of
    // the GEMM BEFORE the one that needs them, covered by that GEMM's counted panel waits (vmcnt retires in issue order, which the
    // weight ring relies on as well), and read with a hand-waited ds_read_b128.
    f32x4* const mlds = smem + kNBuf * kPanelF4 + kTabF4 + (kMaskLds ? wave_u * (kTiles * 2 * 64) : 0);
    auto gates_dma = [&](int layer_idx, int slot) __attribute__((always_inline)) {
        if constexpr (kMaskLds) {
#pragma unroll
            for (int n = 0; n < kTiles; ++n) {
                const uint32_t* g = a.ws_mask + (((int64_t)opaque_uniform(chunk[n]) * L::n_mask_layers + layer_idx) * 64 + lane_id()) * L::mask_words;
                __builtin_amdgcn_global_load_lds((glb_ptr_t)g, (lds_ptr_t)(mlds + (2 * n + slot) * 64), 16, 0, 0);
            }
        }
    };
#pragma unroll
    for (int n = 0; n < kTiles; ++n) {
        const int64_t sn = sample(n);
        // padded samples carry zero gradients so they add nothing to the weight gradients (their rows of the plane, zeroed here, feed the
        // weight-gradient kernel).  Clamped load times a 0 / 1 weight: no zero vector for the compiler to hoist out of the pass loop.
        const bool live = sn < a.S;
        dout[n] = *reinterpret_cast<const f32x4*>(a.ws_dout4 + 4 * (live ? sn : a.S - 1)) * (live ? 1.f : 0.f);
        if (!live && (lane_id() >> 5) == 0) *reinterpret_cast<f32x4*>(a.ws_dout4 + 4 * sn) = dout[n];
    }
    // The forward left every sample's position and view direction where this kernel will put their gradients (nnr_mlp_bf16.h).  They are
    // needed three GEMMs to a whole pass later: held in registers they are the first thing hipcc spills (and a spill reload drains the
    // store queue like any other load); DMA'd straight into LDS they cost nothing until a hand-waited ds_read fetches them.
    f32x4* const plds = smem + kNBuf * kPanelF4 + kTabF4 + (kMaskLds ? W * kTiles * 2 * 64 : 0) + wave_u * (kTiles * 2 * 64);
#pragma unroll
    for (int n = 0; n < kTiles; ++n) {
        const int64_t sn = sample(n);
        const int64_t sc = sn < a.S ? sn : a.S - 1;
        __builtin_amdgcn_global_load_lds((glb_ptr_t)(a.ws_dpts + 4 * sc), (lds_ptr_t)(plds + (2 * n) * 64), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((glb_ptr_t)(a.ws_dview + 4 * sc), (lds_ptr_t)(plds + (2 * n + 1) * 64), 16, 0, 0);
    }
    auto parked = [&](int n, int which) __attribute__((always_inline)) -> f32x4 {     // which: 0 position, 1 view direction
        f32x4 v = frag_read(lds_byte_address(plds + (2 * n + which) * 64) + 16u * lane_id(), 0);
        wait_frag(v, 0);
        return v;
    };

    // (the row dimension of the packed arrays is padded by 4: with rows adjacent in memory hipcc forms a 32-byte access across the
    // row boundary and then leaves those 8 registers in scratch memory)
    uint32_t dq[kTiles][NQ + 4];   // current D-wide gradient, packed (d pre-activation of hidden 8..1), rewritten in place
    f32x16 accA[kTiles][HT], accB[kTiles][HT];   // halves A ([0,D/2)) and B ([D/2,D)) of the gradient being computed
    uint32_t mwA[kTiles][HW], mwB[kTiles][HW];   // ReLU gates of the layer whose gradient sits in accA / accB (gate_append's layout)
    // gates of both halves from LDS slot `slot` (D = 256); the DMA that filled it is older than a counted panel wait this wave has passed
    auto gates_read = [&](int slot) __attribute__((always_inline)) {
        static_assert(!kMaskLds || HW == 2, "a lane's 16 bytes are halves A and B");
        if constexpr (kMaskLds) {
#pragma unroll
            for (int n = 0; n < kTiles; ++n) {
                f32x4 v = frag_read(lds_byte_address(mlds + (2 * n + slot) * 64) + 16u * lane_id(), 0);
                wait_frag(v, 0);
                const u32x4 q = __builtin_bit_cast(u32x4, v);
#pragma unroll
                for (int w = 0; w < HW; ++w) { mwA[n][w] = q[w]; mwB[n][w] = q[HW + w]; }
            }
        }
    };
    auto load_mask = [&](uint32_t(&mw)[kTiles][HW], int layer_idx, int hb) __attribute__((always_inline)) {   // D = 128: direct loads
#pragma unroll
        for (int n = 0; n < kTiles; ++n) {
            const uint32_t* m = a.ws_mask + (((int64_t)opaque_uniform(chunk[n]) * L::n_mask_layers + layer_idx) * 64 + lane_id()) * L::mask_words + hb * HW;
#pragma unroll
            for (int w = 0; w < HW; ++w) {
                mw[n][w] = m[w];
            }
        }
    };
    // gates of layer `layer_idx`, both halves: requested one GEMM ahead (prefetch), taken into mwA / mwB where the old ones have been used up
    auto gates_prefetch = [&](int layer_idx) __attribute__((always_inline)) {
        if constexpr (kMaskLds) gates_dma(layer_idx, layer_idx & 1);
    };
    auto gates_fetch = [&](int layer_idx, bool both) __attribute__((always_inline)) {
        if constexpr (kMaskLds) {
            gates_read(layer_idx & 1);
        } else {
            load_mask(mwA, layer_idx, 0);
            if (both) load_mask(mwB, layer_idx, 1);
        }
    };
// one epilogue unit u: tile u % T, packed register u / T of the half -- dq[tile][OFF + u / T] = (relu'(.) ? acc : 0) x 2 as bf16
// (the gates of a pair as the AND mask of the packed pair: shift, smear, and -- gate_mask, nnr_mlp_bf16.h)
#define NNR_SEL_UNIT(ACC, OFF, MW)                                                                           \
    [&](int uu) __attribute__((always_inline)) {                                                             \
        if constexpr (kPairs && T == 2 && kPh == 1 && !kSplitAsm) {                                          \
            /* paired units (nnr_mlp_bf16.h): call 2 p = phase 0, 2 p + 1 = phase 1 of packed register p of BOTH tiles */ \
            const int p = uu / 2, ph = uu % 2;                                                               \
            if (ph == 0) {                                                                                   \
                pack2(dq[0][(OFF) + p], dq[1][(OFF) + p], ACC[0][(2 * p) >> 4][(2 * p) & 15], ACC[0][(2 * p + 1) >> 4][(2 * p + 1) & 15], \
                      ACC[1][(2 * p) >> 4][(2 * p) & 15], ACC[1][(2 * p + 1) >> 4][(2 * p + 1) & 15]);      \
            } else {                                                                                         \
                gate2_at(dq[0][(OFF) + p], dq[1][(OFF) + p], MW[0][p >> 4], MW[1][p >> 4], p & 15);          \
            }                                                                                                \
        } else {                                                                                             \
        const int u = uu / kPh, ph = uu % kPh;                                                               \
        const int n = u % T, p = u / T;                                                                      \
        const float x0 = ACC[n][(2 * p) >> 4][(2 * p) & 15], x1 = ACC[n][(2 * p + 1) >> 4][(2 * p + 1) & 15]; \
        if constexpr (kPh == 1 && !kSplitAsm) {   /* one asm statement per pair: no compiler-inserted s_nops (nnr_mlp_bf16.h) */ \
            dq[n][(OFF) + p] = sel_pair(x0, x1, MW[n][p >> 4], p & 15);                                      \
        } else {                                                                                             \
            if (kPh == 1 || ph == 0) dq[n][(OFF) + p] = pack_bf16(x0, x1);                                   \
            if (kPh == 1 || ph == 1) dq[n][(OFF) + p] &= gate_mask(MW[n][p >> 4], p & 15);                   \
        }                                                                                                    \
        }                                                                                                    \
    }
    __bf16* const no_stash[kTiles] = {};
    constexpr int PA = 2 * T + 1, PB = 2 * T;   // epilogue units per row of a pass A / pass B (mlp_fwd_bf16_kernel)

    // ---- colour branch ----
    // d g = relu'(g) .* (Wc^T d rgb_pre): three FMAs per value against the rgb rows in LDS (a 3-deep GEMM is not MFMA work)
    uint32_t dgq[kTiles][NP + 4];
    gates_prefetch(8);   // colour-hidden gates and hidden 8's: needed at once -- this is the one place a pass waits for them
    gates_prefetch(7);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the loads above are waited for here anyway; nothing else is in flight yet)
    gates_fetch(8, false);
#pragma unroll
    for (int q = 0; q < HR / 4; ++q) {
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(wrgb + (0 + (lane_id() >> 5)) * HR + 4 * q);
        const f32x4 w1 = *reinterpret_cast<const f32x4*>(wrgb + (2 + (lane_id() >> 5)) * HR + 4 * q);
        const f32x4 w2 = *reinterpret_cast<const f32x4*>(wrgb + (4 + (lane_id() >> 5)) * HR + 4 * q);
#pragma unroll
        for (int n = 0; n < kTiles; ++n) {
            float v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = fmaf(w2[i], dout[n][2], fmaf(w1[i], dout[n][1], w0[i] * dout[n][0]));
            dgq[n][2 * q] = pack_bf16(v[0], v[1]) & gate_mask(mwA[n][(2 * q) >> 4], (2 * q) & 15);
            dgq[n][2 * q + 1] = pack_bf16(v[2], v[3]) & gate_mask(mwA[n][(2 * q + 1) >> 4], (2 * q + 1) & 15);
        }
    }
    // [d h8 ; d gamma(v)] from d g.  The feature layer is folded into the colour-hidden layer (nnr_layout.h): d h8 =
    // relu'(h8) .* (W'^T d g + w_sigma^T d sigma_raw), the rank-1 density term being the accumulator's initial value.
    auto init_sigma = [&](f32x16(&acc)[kTiles][HT], int hb) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < HT; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 w4 = *reinterpret_cast<const f32x4*>(wsig_() + hb * HR + 16 * t + 4 * q);
#pragma unroll
                for (int n = 0; n < kTiles; ++n)
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[n][t][4 * q + i] = w4[i] * dout[n][3];
            }
    };
    gates_fetch(7, true);     // mwA for the side units of B_RGBH_FB, mwB for the first trunk layer's pass A
    init_sigma(accA, 0);
    // d g goes to P_DG (tile-major, one more group than D/32: the output gradients themselves -- d rgb_pre[0..2], d sigma_raw, zeros --
    // as bf16, the gradient operand of the two head layers in the weight-gradient kernel)
    __bf16* dg_stash[kTiles];
#pragma unroll
    for (int n = 0; n < kTiles; ++n) {
        dg_stash[n] = tile_lane(a.ws_dg, row0(n), D / 2 + 16, lane_id());
        u32x4 q = {0u, 0u, 0u, 0u};
        if ((lane_id() >> 5) == 0) {
            q[0] = pack_bf16(dout[n][0], dout[n][1]);
            q[1] = pack_bf16(dout[n][2], dout[n][3]);
        }
        stash_store(dg_stash[n] + kBlockBf16 * (D / 32), q);
    }
    init_sigma(accB, 1);      // (before the GEMM, not after it: the output gradients then die here instead of being spilled across it)
    gemm_wide<HT, HT, true, 0, 1, 0, 0>(accA, dgq, pipe, p0(B_RGBH_FA), dg_stash, NoSide{});
    // G = 2 HT rows; unit u writes dq[.][u >> 1] -- not an input of this part
    gemm_wide<HT, HT, false, kPh * NU, kPh * (NU / (2 * HT)), 0, stash_tail<HT, HT, T>()>(accB, dgq, pipe, p0(B_RGBH_FB), no_stash, NNR_SEL_UNIT(accA, 0, mwA));
    {
        f32x16 accd[kTiles][1];
        zero_acc2(accd);
        gemm_wide<HT, 1>(accd, dgq, pipe, p0(B_RGBH_D));
#pragma unroll
        for (int n = 0; n < kTiles; ++n) {
            const int64_t sn = sample(n);
            const bool live = sn < a.S;
            const f32x4 vd = parked(n, 1);
            const f32x4 gv = enc_chain<16, 4>([&](int r) { return accd[n][0][r]; }, vd[0], vd[1], vd[2], (lane_id() >> 5));
            if ((lane_id() >> 5) == 0 && live) *reinterpret_cast<f32x4*>(a.ws_dview + 4 * sn) = gv;
        }
    }
    NNR_STAMP(tl_dgrad16, 1);

    // ---- trunk ----
    auto dh = [&](int hidden_idx /*0..7*/, int n) -> __bf16* { return tile_lane(a.ws_dh, (int64_t)hidden_idx * a.S_pad + row0(n), D, lane_id()); };
    // Invariant from here on: dq[.][0, NP) holds half A of the newest gradient, accB its half B still to be masked (mwB).

    // one transposed D x D layer at panel pa: consumes the gradient in dq (stashing it to st[]), produces the gradient of the layer
    // below, masked by the ReLU gates of hidden layer `mask_idx`
    auto bwd_layer = [&](int pa, __bf16* const (&st)[kTiles], int mask_idx) __attribute__((always_inline)) {
        zero_acc2(accA);
        gates_prefetch(mask_idx);      // T DMA elements, younger than the first panel's pieces: PRE = T
        // pass A: rows [0, G/2) only read dq[.][0, NP); the previous gradient's half B is finished meanwhile (unit u at row u / PA, see
        // mlp_fwd_bf16_kernel)
        gemm_wide<DT, HT, true, kPh * NU, kPh * PA, 0, kMaskLds ? T : 0>(accA, dq, pipe, pa, st, NNR_SEL_UNIT(accB, NP, mwB));
        gates_fetch(mask_idx, true);   // mwB's old contents were used up by the units of pass A
        zero_acc2(accB);
        // pass B: half A of the new gradient replaces dq[.][0, NP) in place behind the reads (unit u at row u / PB + 1)
        gemm_wide<DT, HT, false, kPh * NU, kPh * PB, 1, stash_tail<DT, HT, T>()>(accB, dq, pipe, pa + PP, no_stash, NNR_SEL_UNIT(accA, 0, mwA));
    };
    // hidden 8,7,6 -> d pre-activation of 7,6,5
#pragma unroll 1
    for (int l = 0; l < 3; ++l) {
        __bf16* st[kTiles];
#pragma unroll
        for (int n = 0; n < kTiles; ++n) st[n] = dh(7 - l, n);
        bwd_layer(p0(B_L8A) + 2 * PP * l, st, 6 - l);
    }
    NNR_STAMP(tl_dgrad16, 2);
    // hidden 5 (skip layer), three passes over W5^T: rows [D, D+63) -> d posenc, rows [0,D) -> d h4.  The chain rule through gamma_10
    // is linear in d posenc, so this part's share of d point is formed right away and added to the first layer's at the end.
    f32x4 gp5[kTiles];
    {
        __bf16* st[kTiles];
#pragma unroll
        for (int n = 0; n < kTiles; ++n) st[n] = dh(4, n);
        f32x16 acce[kTiles][2];
        zero_acc2(acce);
        gates_prefetch(3);
        gemm_wide<DT, 2, true, kPh * NU, kPh * PA, 0, kMaskLds ? T : 0>(acce, dq, pipe, p0(B_L5E), st, NNR_SEL_UNIT(accB, NP, mwB));
        gates_fetch(3, true);
#pragma unroll
        for (int n = 0; n < kTiles; ++n) {
            const f32x4 ps = parked(n, 0);
            gp5[n] = enc_chain<32, 10>([&](int r) { return acce[n][r >> 4][r & 15]; }, ps[0], ps[1], ps[2], (lane_id() >> 5));
        }
    }
    zero_acc2(accA);
    gemm_wide<DT, HT, stash_tail<DT, 2, T>()>(accA, dq, pipe, p0(B_L5HA));   // (B_L5E's last stash stores may stay in flight)
    zero_acc2(accB);
    gemm_wide<DT, HT, false, kPh * NU, kPh * PB, 1, 0>(accB, dq, pipe, p0(B_L5HB), no_stash, NNR_SEL_UNIT(accA, 0, mwA));
    NNR_STAMP(tl_dgrad16, 3);
    // hidden 4,3,2 -> d pre-activation of 3,2,1
#pragma unroll 1
    for (int l = 0; l < 3; ++l) {
        __bf16* st[kTiles];
#pragma unroll
        for (int n = 0; n < kTiles; ++n) st[n] = dh(3 - l, n);
        bwd_layer(p0(B_L4A) + 2 * PP * l, st, 2 - l);
    }
    NNR_STAMP(tl_dgrad16, 4);
    // hidden 1: d posenc = W1^T d1, chain rule through gamma_10, + the skip layer's share -> d point
    {
        __bf16* st[kTiles];
#pragma unroll
        for (int n = 0; n < kTiles; ++n) st[n] = dh(0, n);
        f32x16 acc2[kTiles][2];
        zero_acc2(acc2);
        gemm_wide<DT, 2, true, kPh * NU, kPh * PA, 0, 0>(acc2, dq, pipe, p0(B_L1), st, NNR_SEL_UNIT(accB, NP, mwB));
#pragma unroll
        for (int n = 0; n < kTiles; ++n) {
            const int64_t sn = sample(n);
            const bool live = sn < a.S;
            const f32x4 ps = parked(n, 0);
            const f32x4 gp = enc_chain<32, 10>([&](int r) { return acc2[n][r >> 4][r & 15]; }, ps[0], ps[1], ps[2], (lane_id() >> 5));
            if ((lane_id() >> 5) == 0 && live) *reinterpret_cast<f32x4*>(a.ws_dpts + 4 * sn) = gp + gp5[n];
        }
    }
    NNR_STAMP(tl_dgrad16, 5);
#undef NNR_SEL_UNIT
    pipe.next_pass(pass + 2 < n_pass);
    }   // pass
}

#ifdef NNR_TIMELINE
extern "C" int nnr_timeline_dgrad16(unsigned long long* host32) {
    return (int)hipMemcpyFromSymbol(host32, HIP_SYMBOL(tl_dgrad16), 32 * sizeof(unsigned long long));
}
#endif

template <int D, int T, int W>
static hipError_t launch(const MlpDgradArgs& a, hipStream_t st) {
    const int64_t per_block = (int64_t)kWideSamples * (a.chunks_per_ray > 0 ? a.chunks_per_ray : 1);
    dim3 grid((unsigned)((a.S_pad + per_block - 1) / per_block)), block(64 * W);
    prof_before(PROF_DGRAD, st);
    hipLaunchKernelGGL((mlp_dgrad_bf16_kernel<D, T, W>), grid, block, 0, st, a);
    prof_after(PROF_DGRAD, st);
    return hipGetLastError();
}

hipError_t launch_mlp_dgrad_bf16(int D, const MlpDgradArgs& a, hipStream_t st) {
    return D == 256 ? launch<256, kBf16Tiles, kBf16Waves>(a, st) : launch<128, kBf16Tiles, kBf16Waves>(a, st);
}

}  // namespace nnr
