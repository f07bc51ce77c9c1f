// nnr_mlp_fwd_bf16.hip -- fused NeRF MLP forward with bf16 MFMA products (NNR_F_BF16), gfx950.  Same function as nnr_mlp_fwd.hip
// (model/rendering.py:184-195 sampling + encodings, model/official_nerf.py:60-96 the MLP) in the arithmetic of BASELINE configs[2]:
// every nn.Linear is a bf16 x bf16 product with fp32 accumulation -- hidden layers on the matrix pipe, the 1-row density head and
// the 3-row rgb head as per-lane bf16 dot products (v_dot2c_f32_bf16) -- biases, activations functions and outputs in fp32.
// One wave = TWO chunks of 32 samples, activations packed bf16 between layers: see nnr_mlp_bf16.h for why.
//
// Roofline: with the stash (training) HBM writes, 4.6 KB/sample; the matrix pipe needs 0.24 ms for 4096 x 128 samples at D = 256.
#include "nnr_kernels.h"
#include "nnr_mlp_bf16.h"

namespace nnr {

NNR_TL_DECL(tl_fwd16)
constexpr bool kAblateMask = false;
constexpr bool kAblateEncStash = false;
#ifdef NNR_TIMELINE
__device__ unsigned long long tl_fwd16_all[3 * 4096];   // per workgroup: start, end (s_memtime), HW_ID
#endif

template <int D, bool TRAIN, int T, int W>
__global__ __launch_bounds__(64 * W, 1) void mlp_fwd_bf16_kernel(MlpFwdArgs a) {
    constexpr int kTiles = T;            // 32-sample chunks per wave
    static_assert(T * W * kChunk == kWideSamples, "a workgroup covers 256 samples per pass");
    using Pipe = PanelPipeT<W>;
    NNR_STAMP(tl_fwd16, (TRAIN ? 0 : 16) + 0);
#ifdef NNR_TIMELINE
    if (TRAIN && threadIdx.x == 0 && blockIdx.x < 4096) {
        tl_fwd16_all[3 * blockIdx.x] = __builtin_amdgcn_s_memtime();
        tl_fwd16_all[3 * blockIdx.x + 2] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_REG_HW_ID
    }
#endif
    using L = Layout<D, true>;
    constexpr int DT = L::DT, HT = L::HT;
    constexpr int HR = 16 * HT;              // fragment registers of half a layer's outputs (fp32 numbering)
    constexpr int NP = HR / 2;               // = packed registers per half and tile = epilogue units per half and tile
    constexpr int HW = (HR + 31) / 32;       // mask words per half
    constexpr int PP = part_panels(DT, HT, true);   // panels of one D x D/2 pass
    constexpr int NQ = 8 * DT;               // packed registers of a D-wide vector
    const int lane0 = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;

    // LDS: the panel ring, a parking area (per wave and lane 12 x 16 bytes: the packed encodings of both tiles), the fp32 bias / head
    // tables of the packed buffer and the two heads' rows once more as packed bf16 pairs in register order
    constexpr int kPark = W * (6 * T) * 64;   // per wave and tile: 4 + 2 slots of 16 bytes per lane
    constexpr int kHead16 = 2 * NQ + 3 * 2 * NP;   // uint32: density row [half][NQ], rgb rows [c][half][NP]
    __shared__ __attribute__((aligned(16))) f32x4 smem[kNBuf * kPanelF4 + kPark + (L::table_floats + 3) / 4 + (kHead16 + 3) / 4];
    float* const ltab = reinterpret_cast<float*>(smem + kNBuf * kPanelF4 + kPark);
    uint32_t* const head16 = reinterpret_cast<uint32_t*>(smem + kNBuf * kPanelF4 + kPark + (L::table_floats + 3) / 4);
    for (int i = threadIdx.x; i < L::table_floats; i += 64 * W) ltab[i] = a.packed[L::bias_base + i];
    __syncthreads();
    {
        const float* hd = ltab + (L::head_base - L::bias_base);   // [2][16 DT] density row, then [3][2][16 HT] rgb rows, register order
        for (int i = threadIdx.x; i < kHead16; i += 64 * W) head16[i] = pack_bf16(hd[2 * i], hd[2 * i + 1]);
    }
    __syncthreads();   // before any DMA is in flight: the last full barrier of the kernel
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    Pipe pipe{reinterpret_cast<const f32x4*>(a.packed) + wave_u * (Pipe::PW * 64), smem, wave_u, lane0, L::fwd_panels};
    // Work decomposition as in mlp_fwd_kernel, in PAIRS of chunks: flat -- workgroup b takes the samples [256 b, 256 b + 256), wave w
    // the 64 from 64 w on; ray mode (a.chunks_per_ray = N / 64 > 0) -- wave w of workgroup b walks the chunk pairs of ray 4 b + w, the
    // weight stream wrapping around from pass to pass.  A chunk index past the end is clamped: the wave recomputes the last chunk and
    // stores the same values again.
    const int n_pass = a.chunks_per_ray > 0 ? a.chunks_per_ray : 1;
    const int64_t last_chunk = a.S_pad / kChunk - 1;
    pipe.more = n_pass > 1;
    pipe.start();
    // Ray mode: the wave walks ONE ray through all its passes -- origin, direction and view direction are fetched once, here (wave-uniform:
    // they live in scalar registers), not once per pass behind the previous pass's stores
    const int ray_u = (int)blockIdx.x * W + wave_u;
    float ray_o[3] = {0.f, 0.f, 0.f}, ray_d[3] = {0.f, 0.f, 0.f}, ray_v[3] = {0.f, 0.f, 0.f};
    if (TRAIN && a.chunks_per_ray > 0) {     // (the inference kernel keeps its composite carry in the registers these would take)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            ray_o[c] = a.pts_o[3 * (int64_t)ray_u + c];
            ray_d[c] = a.pts_d[3 * (int64_t)ray_u + c];
            ray_v[c] = a.view_d[3 * (int64_t)ray_u + c];
        }
    }
    // fused compositing (inference, ray mode): running transmittance and weighted sums of this wave's ray, carried across its chunks
    float cT = 1.f, cr = 0.f, cg = 0.f, cb = 0.f, cz = 0.f, cw = 0.f;
    const bool fuse = !TRAIN && a.fuse_rgb != nullptr;
#pragma unroll 1
    for (int pass = 0; pass < n_pass; ++pass) {
    // the lane index is RE-DERIVED every pass (two instructions, no register carried across the loop) and opaque: keeps lane-constant
    // addresses from being hoisted out of the pass loop and spilled (mlp_fwd_kernel)
    int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    asm volatile("" : "+v"(lane));
    pipe.lane = lane;
    f32x4* const park_w = smem + kNBuf * kPanelF4 + wave_u * (6 * T * 64);      // this wave's parking area (uniform); + lane at use
    const float* bias = ltab - L::bias_base;   // index with L::bias_off(layer)
    auto wsig16_ = [&]() { return head16 + (lane_id() >> 5) * NQ; };
    const int64_t pair = a.chunks_per_ray > 0 ? ((int64_t)blockIdx.x * W + wave_u) * n_pass + pass
                                               : (int64_t)blockIdx.x * W + wave_u;
    int chunk[kTiles];   // chunk index of either tile (< 2^26: S_pad < 2^31)
#pragma unroll
    for (int n = 0; n < kTiles; ++n)      // wave-uniform (block index, wave index, pass counter): lives in scalar registers
        chunk[n] = __builtin_amdgcn_readfirstlane((int)(kTiles * pair + n < last_chunk ? kTiles * pair + n : last_chunk));
    auto row0 = [&](int n) -> int64_t { return (int64_t)opaque_uniform(chunk[n]) * kChunk; };   // first sample of tile n: wave-uniform (scalar)
    auto sample = [&](int n) -> int64_t { return row0(n) + (lane_id() & 31); };                              // this lane's sample of tile n

    // (the row dimension of the packed arrays is padded by 4: with rows adjacent in memory hipcc forms a 32-byte access across the
    // row boundary and then leaves those 8 registers in scratch memory)
    uint32_t hq[kTiles][NQ + 4];   // current layer input, packed; rewritten in place
    f32x16 accA[kTiles][HT], accB[kTiles][HT];   // halves A ([0,D/2)) and B ([D/2,D)) of the layer being computed
    uint32_t mwA[kTiles][HW], mwB[kTiles][HW];
    uint32_t eq[kTiles][16 + 4];   // gamma_10(p) packed: 63 -> 64

    // ---- sampling (model/rendering.py:184-195; unfused mul/add to round like the reference) and encodings, tile by tile ----
    // Everything a pass reads from memory is requested HERE, in one batch, before anything is waited for: every s_waitcnt vmcnt of this
    // phase also waits for the stash stores of the previous pass (vmcnt counts stores, in order), so each separate round trip costs an
    // HBM write latency -- the tile-by-tile version (ten serialised waits, two 64-bit divisions) was ~15 % of a pass.
    // Ray mode: the wave walks ONE ray (wave-uniform: its origin / direction / view direction come through the scalar cache) and the
    // sample index within the ray needs no division; flat mode divides in 32 bits (S_pad < 2^31).
    float s_z[kTiles], s_p[kTiles][3], s_v[kTiles][3];
    {
        int jj[kTiles], rr[kTiles];
        float zlo[kTiles], zhi[kTiles], jit[kTiles];
        float ro[kTiles][3], rd[kTiles][3];
        if (a.chunks_per_ray > 0) {
#pragma unroll
            for (int n = 0; n < kTiles; ++n) {
                rr[n] = ray_u;
                jj[n] = (pass * kTiles + n) * kChunk + (lane_id() & 31);
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    if constexpr (TRAIN) {
                        ro[n][c] = ray_o[c]; rd[n][c] = ray_d[c]; s_v[n][c] = ray_v[c];
                    } else {
                        ro[n][c] = a.pts_o[3 * (int64_t)ray_u + c]; rd[n][c] = a.pts_d[3 * (int64_t)ray_u + c]; s_v[n][c] = a.view_d[3 * (int64_t)ray_u + c];
                    }
                }
            }
        } else {
#pragma unroll
            for (int n = 0; n < kTiles; ++n) {
                const int64_t sn = sample(n);
                const unsigned sc = (unsigned)(sn < a.S ? sn : a.S - 1);   // padded samples recompute the last one
                const unsigned nn = (unsigned)opaque(a.N);      // opaque: the reciprocal sequence of the division is not hoisted out of
                rr[n] = (int)(sc / nn);                           // the pass loop (where it would sit in registers the ray mode needs)
                jj[n] = (int)(sc - (unsigned)rr[n] * nn);
            }
#pragma unroll
            for (int n = 0; n < kTiles; ++n)
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    ro[n][c] = a.pts_o[3 * (int64_t)rr[n] + c];
                    rd[n][c] = a.pts_d[3 * (int64_t)rr[n] + c];
                    s_v[n][c] = a.view_d[3 * (int64_t)rr[n] + c];
                }
        }
#pragma unroll
        for (int n = 0; n < kTiles; ++n) {
            zlo[n] = a.z_lo[jj[n]];
            zhi[n] = a.z_hi[jj[n]];
            jit[n] = a.jitter ? a.jitter[(int64_t)rr[n] * a.N + jj[n]] : 0.f;
        }
#pragma unroll
        for (int n = 0; n < kTiles; ++n) {
            float z = zlo[n];
            if (a.jitter) z = __fadd_rn(zlo[n], __fmul_rn(__fsub_rn(zhi[n], zlo[n]), jit[n]));
            s_z[n] = z;
#pragma unroll
            for (int c = 0; c < 3; ++c) s_p[n][c] = __fadd_rn(ro[n][c], __fmul_rn(rd[n][c], z));
        }
    }
#pragma unroll
    for (int n = 0; n < kTiles; ++n) {
        const int64_t sn = sample(n);
        const float z = s_z[n];
        const float px = s_p[n][0], py = s_p[n][1], pz = s_p[n][2];
        const float vx = s_v[n][0], vy = s_v[n][1], vz = s_v[n][2];
        if ((lane_id() >> 5) == 0 && sn < a.S && !fuse) a.ws_z[sn] = z;
        float e[32];
#pragma unroll
        for (int r = 0; r < 32; ++r) e[r] = enc_register(r, (lane_id() >> 5), kPosReal, px, py, pz);
#pragma unroll
        for (int p = 0; p < 16; ++p) eq[n][p] = pack_bf16(e[2 * p], e[2 * p + 1]);
        float dirv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) dirv[r] = enc_register(r, (lane_id() >> 5), kDirReal, vx, vy, vz);
        uint32_t dq[8];
#pragma unroll
        for (int p = 0; p < 8; ++p) dq[p] = pack_bf16(dirv[2 * p], dirv[2 * p + 1]);
        // both encodings wait in LDS: the position encoding for the skip layer, the direction encoding for the colour layer
#pragma unroll
        for (int q = 0; q < 4; ++q)
            (park_w + lane_id())[(6 * n + q) * 64] = __builtin_bit_cast(f32x4, u32x4{eq[n][4 * q], eq[n][4 * q + 1], eq[n][4 * q + 2], eq[n][4 * q + 3]});
#pragma unroll
        for (int q = 0; q < 2; ++q)
            (park_w + lane_id())[(6 * n + 4 + q) * 64] = __builtin_bit_cast(f32x4, u32x4{dq[4 * q], dq[4 * q + 1], dq[4 * q + 2], dq[4 * q + 3]});
        if constexpr (TRAIN && !kAblateEncStash) {
            // for the input-gradient kernel: the sample's position and view direction (it recomputes the chain-rule factors of the two
            // encodings from them, nnr_mlp_bf16.h -- 32 bytes per sample instead of 384 of factors), parked in the planes that kernel
            // overwrites with their gradients; for the weight-gradient kernel: tile-major bf16 copies of the encodings = its MFMA operands
            if ((lane_id() >> 5) == 0 && sn < a.S) {
                *reinterpret_cast<f32x4*>(a.ws_pts + 4 * sn) = f32x4{px, py, pz, 0.f};
                *reinterpret_cast<f32x4*>(a.ws_view + 4 * sn) = f32x4{vx, vy, vz, 0.f};
            }
            __bf16* e16 = tile_lane(a.ws_xe16, row0(n), kPosPad, lane_id());
#pragma unroll
            for (int b = 0; b < 4; ++b)
                stash_store(e16 + kBlockBf16 * b, u32x4{eq[n][4 * b], eq[n][4 * b + 1], eq[n][4 * b + 2], eq[n][4 * b + 3]});
            __bf16* f16 = tile_lane(a.ws_xf16, row0(n), kDirPad, lane_id());
#pragma unroll
            for (int b = 0; b < 2; ++b) stash_store(f16 + kBlockBf16 * b, u32x4{dq[4 * b], dq[4 * b + 1], dq[4 * b + 2], dq[4 * b + 3]});
        }
    }
    NNR_STAMP(tl_fwd16, (TRAIN ? 0 : 16) + 1);

    // accumulators start at the bias (the same for both tiles), so an epilogue is only ReLU + gate bits + pack
    auto init_acc = [&](f32x16(&acc)[kTiles][HT], int bias_offset) __attribute__((always_inline)) {
        const float* b = bias + bias_offset + 4 * (lane_id() >> 5);
#pragma unroll
        for (int t = 0; t < HT; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 bb = *reinterpret_cast<const f32x4*>(b + 32 * t + 8 * q);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
#pragma unroll
                    for (int n = 0; n < kTiles; ++n) acc[n][t][4 * q + i] = bb[i];
                }
            }
    };
    auto clear_mask = [&](uint32_t(&mw)[kTiles][HW]) __attribute__((always_inline)) {
#pragma unroll
        for (int n = 0; n < kTiles; ++n)
#pragma unroll
            for (int w = 0; w < HW; ++w) mw[n][w] = 0;
    };
    auto store_mask = [&](const uint32_t(&mw)[kTiles][HW], int layer_idx, int hb) __attribute__((always_inline)) {
        if (TRAIN && !kAblateMask) {
#pragma unroll
            for (int n = 0; n < kTiles; ++n) {
                // masks: [chunk][layer][lane][mask_words]; half A of a layer owns the low words, half B the high words.  An ordinary
                // store: the two halves of a lane's 16 bytes arrive from different passes and rely on L2 to merge them into lines.
                uint32_t* m = a.ws_mask + (((int64_t)opaque_uniform(chunk[n]) * L::n_mask_layers + layer_idx) * 64 + lane_id()) * L::mask_words + hb * HW;
                if constexpr (HW == 2) {
                    *reinterpret_cast<u32x2*>(m) = u32x2{mw[n][0], mw[n][1]};
                } else {
#pragma unroll
                    for (int w = 0; w < HW; ++w) m[w] = mw[n][w];
                }
            }
        }
    };
// One epilogue unit u: tile u % T, packed register u / T of the half -- hq[tile][OFF + u / T] = (relu(x0), relu(x1)) as bf16.
// The ReLU gates of the pair are appended to the mask word of its 16 registers (gate_append, nnr_mlp_bf16.h): units run in register order.
// With kPh = 2 a unit is issued as two half-units (index 2 u: read + pack, 2 u + 1: ReLU + gates) in different gaps between MFMAs.
#define NNR_RELU_UNIT(ACC, OFF, MW)                                                                          \
    [&](int uu) __attribute__((always_inline)) {                                                             \
        if constexpr (kPairs && T == 2 && kPh == 1 && !kSplitAsm) {                                          \
            /* paired units (nnr_mlp_bf16.h): call 2 p = phase 0, 2 p + 1 = phase 1 of packed register p of BOTH tiles */ \
            const int p = uu / 2, ph = uu % 2;                                                               \
            if (ph == 0) {                                                                                   \
                pack2(hq[0][(OFF) + p], hq[1][(OFF) + p], ACC[0][(2 * p) >> 4][(2 * p) & 15], ACC[0][(2 * p + 1) >> 4][(2 * p + 1) & 15], \
                      ACC[1][(2 * p) >> 4][(2 * p) & 15], ACC[1][(2 * p + 1) >> 4][(2 * p + 1) & 15]);      \
            } else if (TRAIN && !kAblateMask) {                                                              \
                relu_gate2(hq[0][(OFF) + p], hq[1][(OFF) + p], MW[0][p >> 4], MW[1][p >> 4]);                \
            } else {                                                                                         \
                relu2(hq[0][(OFF) + p], hq[1][(OFF) + p]);                                                   \
            }                                                                                                \
        } else {                                                                                             \
        const int u = uu / kPh, ph = uu % kPh;                                                               \
        const int n = u % T, p = u / T;                                                                      \
        const float x0 = ACC[n][(2 * p) >> 4][(2 * p) & 15], x1 = ACC[n][(2 * p + 1) >> 4][(2 * p + 1) & 15]; \
        if constexpr (kPh == 1 && !kSplitAsm) {   /* one asm statement per pair: no compiler-inserted s_nops (nnr_mlp_bf16.h) */ \
            if (TRAIN && !kAblateMask) hq[n][(OFF) + p] = relu_pack_gate(x0, x1, MW[n][p >> 4]);             \
            else hq[n][(OFF) + p] = relu_pack(x0, x1);                                                       \
        } else {                                                                                             \
            if (kPh == 1 || ph == 0) hq[n][(OFF) + p] = pack_bf16(x0, x1);                                   \
            if (kPh == 1 || ph == 1) {                                                                       \
                hq[n][(OFF) + p] = relu_bf16x2(hq[n][(OFF) + p]);   /* rounding keeps the sign: relu commutes with it */ \
                if (TRAIN && !kAblateMask) MW[n][p >> 4] = gate_append(MW[n][p >> 4], hq[n][(OFF) + p]);     \
            }                                                                                                \
        }                                                                                                    \
        }                                                                                                    \
    }
    auto p0 = [&](int part) { return L::fwd_panel0(part); };
    __bf16* const no_stash[kTiles] = {};
    auto xh = [&](int hidden_idx /*0..7*/, int n) -> __bf16* {
        return TRAIN ? tile_lane(a.ws_xh, (int64_t)hidden_idx * a.S_pad + row0(n), D, lane_id()) : nullptr;
    };
    constexpr int NU = kTiles * NP;   // epilogue units of one half-output pass
    constexpr int TAIL = TRAIN ? stash_tail<DT, HT, T>() : 0;
    constexpr int PA = 2 * T + 1, PB = 2 * T;   // epilogue units per row of a pass A (strictly ahead of the reads) / pass B (behind them)   // stores a stashing D-wide pass leaves in flight for the part after it

    // ---- hidden 1: 63 -> D, input = posenc.  Pass A, then pass B with A's epilogue hidden under it. ----
    init_acc(accA, L::bias_off(0));
    // (training: the 16 stash stores of the encodings above -- 6 bf16 blocks + position + view direction per tile -- are younger than
    // every piece of the first two panels)
    gemm_wide<2, HT, TRAIN ? T * 8 : 0>(accA, eq, pipe, p0(F_L1A));
    init_acc(accB, L::bias_off(0) + L::Dh);
    clear_mask(mwA);
    gemm_wide<2, HT, false, kPh * NU, kPh * (NU / 4), 0, TRAIN ? T * 8 : 0>(accB, eq, pipe, p0(F_L1B), no_stash, NNR_RELU_UNIT(accA, 0, mwA));   // (the same stores are younger than panel 1's pieces too)
    store_mask(mwA, 0, 0);
    NNR_STAMP(tl_fwd16, (TRAIN ? 0 : 16) + 2);
    // Invariant from here on: hq[.][0, NP) holds half A of the newest layer, accB holds its half B still to be finished.

    // one D -> D ReLU layer (state_dict index `li`), packed at panel pa; its input goes to the stash planes st[]
    auto dense_layer = [&](int li, int pa, __bf16* const (&st)[kTiles]) __attribute__((always_inline)) {
        init_acc(accA, L::bias_off(li));
        clear_mask(mwB);
        // pass A: rows [0, G/2) only read hq[.][0, NP); the previous layer's half B is finished meanwhile, unit u at row u / PA --
        // packed register NP + u / T is first read at row G/2 + (u / T) / 4, always a later row (PA = 2 T + 1 units per row)
        gemm_wide<DT, HT, TRAIN, kPh * NU, kPh * PA, 0, 0>(accA, hq, pipe, pa, st, NNR_RELU_UNIT(accB, NP, mwB));
        store_mask(mwB, li - 1, 1);
        init_acc(accB, L::bias_off(li) + L::Dh);
        clear_mask(mwA);
        // pass B: half A of the new layer replaces hq[.][0, NP) in place behind the reads -- unit u runs at row u / PB + 1, its
        // register u / T was last read at row (u / T) / 4
        gemm_wide<DT, HT, false, kPh * NU, kPh * PB, 1, TAIL>(accB, hq, pipe, pa + PP, no_stash, NNR_RELU_UNIT(accA, 0, mwA));
        store_mask(mwA, li, 0);
    };
    // hidden 2..4
#pragma unroll 1
    for (int l = 0; l < 3; ++l) {
        __bf16* st[kTiles];
#pragma unroll
        for (int n = 0; n < kTiles; ++n) st[n] = xh(l, n);
        dense_layer(1 + l, p0(F_L2A) + 2 * PP * l, st);
    }
    NNR_STAMP(tl_fwd16, (TRAIN ? 0 : 16) + 3);
    // hidden 5: [h4 ; e] -> D   (skip connection, input order [h, posenc]: model/official_nerf.py:63)
    {
        __bf16* st[kTiles];
#pragma unroll
        for (int n = 0; n < kTiles; ++n) st[n] = xh(3, n);
        init_acc(accA, L::bias_off(4));
        clear_mask(mwB);
        gemm_wide<DT, HT, TRAIN, kPh * NU, kPh * PA, 0, 0>(accA, hq, pipe, p0(F_L5HA), st, NNR_RELU_UNIT(accB, NP, mwB));
        store_mask(mwB, 3, 1);
#pragma unroll
        for (int n = 0; n < kTiles; ++n)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const u32x4 v = __builtin_bit_cast(u32x4, (park_w + lane_id())[(6 * n + q) * 64]);
#pragma unroll
                for (int i = 0; i < 4; ++i) eq[n][4 * q + i] = v[i];
            }
        gemm_wide<2, HT, TAIL>(accA, eq, pipe, p0(F_L5EA));
        init_acc(accB, L::bias_off(4) + L::Dh);
        clear_mask(mwA);
        gemm_wide<DT, HT, false, kPh * NU, kPh * PB, 1, 0>(accB, hq, pipe, p0(F_L5HB), no_stash, NNR_RELU_UNIT(accA, 0, mwA));
        gemm_wide<2, HT>(accB, eq, pipe, p0(F_L5EB));
        store_mask(mwA, 4, 0);
    }
    NNR_STAMP(tl_fwd16, (TRAIN ? 0 : 16) + 4);
    // hidden 6..8
#pragma unroll 1
    for (int l = 0; l < 3; ++l) {
        __bf16* st[kTiles];
#pragma unroll
        for (int n = 0; n < kTiles; ++n) st[n] = xh(4 + l, n);
        dense_layer(5 + l, p0(F_L6A) + 2 * PP * l, st);
    }
    NNR_STAMP(tl_fwd16, (TRAIN ? 0 : 16) + 5);

    // colour hidden: g = relu(W' h8 + Wg[:, D:] gamma_4(v) + b'), the feature layer folded in by the pack kernel (nnr_layout.h); one
    // pass (D/2 outputs).  Its side work first finishes hidden 8 (half B), then evaluates the density head: a per-lane dot product
    // of the packed h8 with the packed density row, four registers per unit.
    float sg[kTiles][2] = {};
    {
        __bf16* st[kTiles];
#pragma unroll
        for (int n = 0; n < kTiles; ++n) st[n] = xh(7, n);
        init_acc(accA, L::bias_off(10));
        clear_mask(mwB);
        auto finish_then_sigma = [&](int u) __attribute__((always_inline)) {
            if (u < kPh * NU) {
                NNR_RELU_UNIT(accB, NP, mwB)(u);
            } else {   // registers 4v .. 4v+3 of h8, both tiles; every register is final by now: the NU finishing units come first
                const int v = u - kPh * NU;
                const u32x4 w4 = *reinterpret_cast<const u32x4*>(wsig16_() + 4 * v);
#pragma unroll
                for (int n = 0; n < kTiles; ++n)
#pragma unroll
                    for (int i = 0; i < 4; ++i) dot2_bf16(sg[n][i & 1], hq[n][4 * v + i], w4[i]);
            }
        };
        // all units spread evenly over the rows: finishing unit u (register NP + u / T, first read at row G/2 + (u / T) / 4) runs
        // well before that row
        gemm_wide<DT, HT, TRAIN, kPh * NU + NQ / 4, (kPh * NU + NQ / 4 + 2 * DT - 1) / (2 * DT), 0, 0>(accA, hq, pipe, p0(F_RGBH_F), st, finish_then_sigma);
        store_mask(mwB, 7, 1);
    }
    NNR_STAMP(tl_fwd16, (TRAIN ? 0 : 16) + 6);
    {
        uint32_t dq[kTiles][8];
#pragma unroll
        for (int n = 0; n < kTiles; ++n)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const u32x4 v = __builtin_bit_cast(u32x4, (park_w + lane_id())[(6 * n + 4 + q) * 64]);
#pragma unroll
                for (int i = 0; i < 4; ++i) dq[n][4 * q + i] = v[i];
            }
        gemm_wide<1, HT, TAIL>(accA, dq, pipe, p0(F_RGBH_D));
    }
    clear_mask(mwA);
#pragma unroll
    for (int u = 0; u < kPh * NU; ++u) NNR_RELU_UNIT(accA, 0, mwA)(u);   // g = hq[.][0, NP)
    store_mask(mwA, 8, 0);
#pragma unroll
    for (int n = 0; n < kTiles; ++n) {
        const int64_t sn = sample(n);
        if constexpr (TRAIN) {   // tile-major bf16 plane: group gq = packed registers 4 gq .. 4 gq + 3
            __bf16* xg = tile_lane(a.ws_xg, row0(n), D / 2, lane_id());
#pragma unroll
            for (int gq = 0; gq < NP / 4; ++gq)
                stash_store(xg + kBlockBf16 * gq, u32x4{hq[n][4 * gq], hq[n][4 * gq + 1], hq[n][4 * gq + 2], hq[n][4 * gq + 3]});
        }
        const float sgn = dot2_result(sg[n][0]) + dot2_result(sg[n][1]);
        const float sigma_raw = sum_halves(sgn) + bias[L::bias_off(8)];
        // rgb head: 3 per-lane dot products over the lane's half of g, halves combined by one shuffle, then sigmoid
        float rgbv[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const uint32_t* wc = head16 + 2 * NQ + (2 * c + (lane_id() >> 5)) * NP;
            float acc0 = 0.f, acc1 = 0.f;
#pragma unroll
            for (int q = 0; q < NP / 4; ++q) {
                const u32x4 w4 = *reinterpret_cast<const u32x4*>(wc + 4 * q);
                dot2_bf16(acc0, hq[n][4 * q], w4[0]);
                dot2_bf16(acc1, hq[n][4 * q + 1], w4[1]);
                dot2_bf16(acc0, hq[n][4 * q + 2], w4[2]);
                dot2_bf16(acc1, hq[n][4 * q + 3], w4[3]);
            }
            const float part = dot2_result(acc0) + dot2_result(acc1);
            rgbv[c] = sum_halves(part);
        }
        const float* bo = bias + L::bias_off(11);
        f32x4 o;
        o[0] = sigmoid_ref(rgbv[0] + bo[0]);
        o[1] = sigmoid_ref(rgbv[1] + bo[1]);
        o[2] = sigmoid_ref(rgbv[2] + bo[2]);
        o[3] = sigma_raw;
        if (!fuse) {
            if ((lane_id() >> 5) == 0 && sn < a.S) *reinterpret_cast<f32x4*>(a.ws_out4 + 4 * sn) = o;
        } else if constexpr (!TRAIN) {
            // compositing of this chunk's 32 samples on top of the ray's carry, as in mlp_fwd_kernel (ray mode: sn < S, whole chunks)
            const int ray = (int)(sn / a.N);
            const int j = (int)(sn - (int64_t)ray * a.N), jn = j + 1;
            const float lo0 = a.z_lo[j], hi0 = a.z_hi[j];
            const float z = a.jitter ? __fadd_rn(lo0, __fmul_rn(__fsub_rn(hi0, lo0), a.jitter[sn])) : lo0;
            float zn = 0.f;
            if (jn < a.N) {
                const float lo1 = a.z_lo[jn], hi1 = a.z_hi[jn];
                zn = a.jitter ? __fadd_rn(lo1, __fmul_rn(__fsub_rn(hi1, lo1), a.jitter[sn + 1])) : lo1;
            }
            float unused;
            const float alpha = (lane_id() >> 5) == 0 ? sample_alpha(o[3], jn < a.N ? zn - z : 1e10f, jn == a.N, a.flags, unused) : 0.f;
            const float incl = wave_scan_mul((lane_id() >> 5) == 0 ? (1.f - alpha) + kEpsT : 1.f, lane_id());
            float excl = __shfl_up(incl, 1, 64);
            if (lane_id() == 0) excl = 1.f;
            const float w = alpha * cT * excl;
            cT *= __shfl(incl, 31, 64);
            cr += w * o[0]; cg += w * o[1]; cb += w * o[2]; cz += w * z; cw += w;
            if (pass + 1 == n_pass && n == kTiles - 1) {
                const float sr = wave_sum(cr), sg2 = wave_sum(cg), sb = wave_sum(cb), sz = wave_sum(cz), sw = wave_sum(cw);
                if (lane_id() == 0) {
                    const float bg = (a.flags & kFlagWhiteBg) ? 1.f - sw : 0.f;
                    float* out = a.fuse_rgb + 3 * (int64_t)ray;
                    out[0] = sr + bg; out[1] = sg2 + bg; out[2] = sb + bg;
                    a.fuse_dist[ray] = sz;
                }
            }
        }
    }
    NNR_STAMP(tl_fwd16, (TRAIN ? 0 : 16) + 7);
#undef NNR_RELU_UNIT
    pipe.next_pass(pass + 2 < n_pass);
    }   // pass
#ifdef NNR_TIMELINE
    if (TRAIN && threadIdx.x == 0 && blockIdx.x < 4096) tl_fwd16_all[3 * blockIdx.x + 1] = __builtin_amdgcn_s_memtime();
#endif
}

#ifdef NNR_TIMELINE
extern "C" int nnr_timeline_fwd16_all(unsigned long long* host) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(tl_fwd16_all), 3 * 4096 * sizeof(unsigned long long));
}
extern "C" int nnr_timeline_fwd16(unsigned long long* host32) {
    return (int)hipMemcpyFromSymbol(host32, HIP_SYMBOL(tl_fwd16), 32 * sizeof(unsigned long long));
}
#endif

template <int D, int T, int W>
static hipError_t launch(const MlpFwdArgs& a, bool train, hipStream_t st) {
    // ray mode: one workgroup per W rays, chunks_per_ray passes of 32 T samples each; flat: one workgroup per 256 samples
    const int64_t per_block = (int64_t)kWideSamples * (a.chunks_per_ray > 0 ? a.chunks_per_ray : 1);
    dim3 grid((unsigned)((a.S_pad + per_block - 1) / per_block)), block(64 * W);
    prof_before(train ? PROF_FWD_TRAIN : PROF_FWD_INFER, st);
    if (train) hipLaunchKernelGGL((mlp_fwd_bf16_kernel<D, true, T, W>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((mlp_fwd_bf16_kernel<D, false, T, W>), grid, block, 0, st, a);
    prof_after(train ? PROF_FWD_TRAIN : PROF_FWD_INFER, st);
    return hipGetLastError();
}

hipError_t launch_mlp_fwd_bf16(int D, const MlpFwdArgs& a, bool train, hipStream_t st) {
    return D == 256 ? launch<256, kBf16Tiles, kBf16Waves>(a, train, st) : launch<128, kBf16Tiles, kBf16Waves>(a, train, st);
}

}  // namespace nnr
