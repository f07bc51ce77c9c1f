// nnr_wgrad.hip -- (fp32 mode; the bf16 training mode has its own HBM-bound kernel, nnr_wgrad_bf16.hip)
// weight / bias gradients of the 12 nn.Linear layers from the stashed layer inputs X and the stashed
// pre-activation gradients Dlt:   dW_l[out][in] = sum_s Dlt_l[s][out] * X_l[s][in],   db_l[out] = sum_s Dlt_l[s][out].
// Replaces autograd's `mm` wgrad calls (24 of the 36 backward GEMMs) for model/official_nerf.py:20-37.
//
// Shape of the problem: outputs are tiny (2.4 MB), the reduction dimension is every sample of the step.  So this is a
// split-K kernel: a *wave job* owns a (32*MI) x (32*NI) tile of one dW and a contiguous range of samples, keeps the
// tile in MI*NI MFMA accumulators (256 registers for 4x4) for the whole range and writes it once to its own partial
// slot; wgrad_reduce_kernel then adds the slots of each tile into dW.
// Both operands are read straight from the (sample, feature) row-major stashes: with interleaved sub-tiles
// (row = MI*m + i) one MI-wide and one NI-wide vector load per lane feed MI*NI MFMAs; the four jobs of a 256x256 layer
// that share a sample range sit in one workgroup so their re-reads hit L1/L2.  The job table comes from the host plan
// (nnr_api.cpp: build_plan -- a balanced static schedule, one wave program per SIMD of the chip).  The feature layer and the
// first D columns of the colour-hidden layer get their gradients from the merged matrix W' (wgrad_unmerge_kernel).
#include "nnr_device.h"
#include "nnr_kernels.h"
#include "nnr_split.h"
#include <cstdlib>

namespace nnr {

NNR_TL_DECL(tl_wgrad)
#ifdef NNR_TIMELINE
__device__ unsigned long long tl_wgrad_all[4096];   // [wave slot][start, end]
extern "C" int nnr_timeline_wgrad_all(unsigned long long* host4096) {
    return (int)hipMemcpyFromSymbol(host4096, HIP_SYMBOL(tl_wgrad_all), 4096 * sizeof(unsigned long long));
}
#endif

template <int W>
struct Vec { float v[W]; };

// Unconditional vector load, no masking of the value.  A lane outside the valid column range reads a valid address (its
// pointer is clamped to column 0 by the caller); what it contributes lands only in its own row / column of the MFMA
// result, which the flush never writes.  Neither a branch around the load nor a select on the loaded value is allowed
// here: both make hipcc wait for the just-issued prefetch (vmcnt(0) at the join / before the v_cndmask) and serialise
// the pipeline (cdna_hip_programming.md, "three .s-level traps" (c)).
template <int W>
__device__ __forceinline__ Vec<W> load_vec(const float* p) {
    Vec<W> r;
    if constexpr (W == 4) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(p);
        r.v[0] = t[0]; r.v[1] = t[1]; r.v[2] = t[2]; r.v[3] = t[3];
    } else if constexpr (W == 2) {
        const f32x2 t = *reinterpret_cast<const f32x2*>(p);
        r.v[0] = t[0]; r.v[1] = t[1];
    } else {
        r.v[0] = *p;
    }
    return r;
}

constexpr int kU = 4;  // k-steps (pairs of samples) per pipeline stage
constexpr int32_t kPlanMagic = 0x4e4e5235;      // 'NNR5' (nnr_api.cpp: nnr_plan_build writes the trailer)

template <int MI, int NI, int BIAS>   // BIAS: WgradJob::bias
__device__ __forceinline__ void wgrad_job(const WgradJob& jb, const WgradArgs& a, int lane, int ji) {
    const int half = lane >> 5, m = lane & 31;
    const int dp = a.plane_pitch[jb.d_plane], xp = a.plane_pitch[jb.x_plane];
    const bool dok = MI * m < jb.d_valid, xok = NI * m < jb.x_valid;
    // Row of sample s of an operand = base + (s >> 5) * A + (s & 31) * B floats, for both plane layouts (nnr_layout.h): row-major
    // A = 32 pitch, B = pitch; tile-major fp32 (the gradient planes of the three-term mode) A = one chunk of blocks = 32 pitch, B = 4,
    // and the lane's columns c .. c + W - 1 (W <= 4, c a multiple of W) sit at (c >> 3) * 256 + ((c >> 2) & 1) * 128 + (c & 3) of the chunk.
    // The samples of a stage are k + 2 u + half with k a multiple of 16: (s & 31) = (k & 31) + 2 u + half never carries.
    const bool dt = a.plane_tile[jb.d_plane] != 0, xt = a.plane_tile[jb.x_plane] != 0;
    const int dc = jb.d_col0 + (dok ? MI * m : 0), xc = jb.x_col0 + (xok ? NI * m : 0);
    const int d_row = dt ? 4 : dp, x_row = xt ? 4 : xp;                         // B
    const int64_t d_chunk = 32 * (int64_t)dp, x_chunk = 32 * (int64_t)xp;      // A
    const float* dptr = a.ws + a.plane_off[jb.d_plane] + (dt ? (dc >> 3) * 256 + ((dc >> 2) & 1) * 128 + (dc & 3) : dc) + half * d_row;
    const float* xptr = a.ws + a.plane_off[jb.x_plane] + (xt ? (xc >> 3) * 256 + ((xc >> 2) & 1) * 128 + (xc & 3) : xc) + half * x_row;

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float bsum[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) bsum[i] = 0.f;

    // Two register stages (A, B) of kU k-steps each, explicitly ping-ponged: the loads of one stage are issued before
    // the 16*kU MFMAs of the other, and no register copies sit between a load and its first use (a rotating
    // "cur = next" copy makes hipcc wait for the prefetch right after issuing it).  Sample ranges are multiples of
    // 4*kU = 16 samples (kGranule in nnr_api.cpp).
    Vec<MI> dA[kU], dB[kU];
    Vec<NI> xA[kU], xB[kU];
    auto load_stage = [&](Vec<MI>(&d)[kU], Vec<NI>(&x)[kU], int64_t kk) {
        const float* const dk = dptr + (kk >> 5) * d_chunk + (kk & 31) * d_row;      // (the wave-uniform part of the address)
        const float* const xk = xptr + (kk >> 5) * x_chunk + (kk & 31) * x_row;
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            d[u] = load_vec<MI>(dk + 2 * u * d_row);
            x[u] = load_vec<NI>(xk + 2 * u * x_row);
        }
    };
    auto compute = [&](const Vec<MI>(&d)[kU], const Vec<NI>(&x)[kU]) {
#pragma unroll
        for (int u = 0; u < kU; ++u) {
#pragma unroll
            for (int i = 0; i < MI; ++i) {
#pragma unroll
                for (int j = 0; j < NI; ++j) acc[i][j] = mfma32(d[u].v[i], x[u].v[j], acc[i][j]);
                if (BIAS == 1 || (BIAS == 2 && u % 2 == 0) || (BIAS == 3 && u % 2 == 1)) bsum[i] += d[u].v[i];   // u is unrolled: folds
            }
        }
    };
    NNR_STAMP(tl_wgrad, 0);
    load_stage(dA, xA, jb.k0);
    for (int64_t k = jb.k0; k < jb.k1; k += 4 * kU) {
        // sched_barrier(0): nothing moves across.  Without it the scheduler sinks each load down to its first use
        // (load; vmcnt(0); mfma) and the 4096-cycle MFMA block no longer covers the memory latency.
        load_stage(dB, xB, k + 2 * kU);
        __builtin_amdgcn_sched_barrier(0);
        compute(dA, xA);
        __builtin_amdgcn_sched_barrier(0);
        load_stage(dA, xA, (k + 4 * kU < jb.k1) ? k + 4 * kU : k);   // the last refill re-reads (unused): no branch around loads
        __builtin_amdgcn_sched_barrier(0);
        compute(dB, xB);
        __builtin_amdgcn_sched_barrier(0);
    }

    NNR_STAMP(tl_wgrad, 1);
#ifdef NNR_TIMELINE
    if (blockIdx.x == 700 / 8 && threadIdx.x == 0) { tl_wgrad[4] = (unsigned long long)(jb.k1 - jb.k0); tl_wgrad[5] = MI * 8 + NI; }
#endif
    // flush: D[row m'][col n] of sub-tile (i,j) -> slot[(MI*m' + i) * 32*NI + NI*n + j].  One NI-wide store per (i, r):
    // 32 lanes cover a whole tile row (32*NI contiguous floats).  Rows / columns outside the valid range hold garbage
    // (see load_vec); the reduction never reads them.
    float* slot = a.slots + (int64_t)ji * kSlotFloats;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int mr = (r & 3) + 8 * (r >> 2) + 4 * half;
            float* dst = slot + (MI * mr + i) * (32 * NI) + NI * m;
            if constexpr (NI == 4) *reinterpret_cast<f32x4*>(dst) = f32x4{acc[i][0][r], acc[i][1][r], acc[i][2][r], acc[i][3][r]};
            else if constexpr (NI == 2) *reinterpret_cast<f32x2*>(dst) = f32x2{acc[i][0][r], acc[i][1][r]};
            else *dst = acc[i][0][r];
        }
    if constexpr (BIAS != 0) {   // this half-wave's share of sum_s Dlt[s][row0 + MI*m + i]
        float* dst = slot + kSlotTile + half * (32 * MI) + MI * m;
#pragma unroll
        for (int i = 0; i < MI; ++i) dst[i] = bsum[i];
    }
    NNR_STAMP(tl_wgrad, 2);
}

// ---- the 4 x 4 tile with every product as six bf16 MFMA terms (NNR_F_SPLIT3; the idea: nnr_split.h) -----------------------------------
// Same job, same slot, same flush.  The reduction index (samples) is the k of v_mfma_f32_32x32x16_bf16, so a lane supplies EIGHT samples
// per operand and MFMA -- lane (h, m) the samples k + 8 h + s, s = 0..7, of feature columns MI m .. MI m + 3 (the four interleaved
// sub-tiles) -- and a step is 16 samples: 8 rows of 16 bytes per lane and operand, the fp32 kernel's bytes.  The 256 accumulators fill
// the AGPR file, and a register-resident pipeline of the loaded rows does not fit beside the term registers (thousands of
// spills), so the rows are STAGED IN LDS: each wave owns two 16 KiB buffers, filled by global_load_lds (no VGPR round trip, no barrier --
// the area is private to the wave) one step ahead, and the values come back two at a time, one gap before they are split.
//   Xc / Xn    the activation operand's terms of this step and the next (2 x 48 registers: every block needs all four sub-tiles)
//   Dq[2]      the terms of ONE gradient sub-tile, made one block ahead (2 x 12)
// Block i of a step = 24 MFMAs (gradient sub-tile i against the four activation sub-tiles, six terms each); under them
//   block 0: DMA of the next step's 16 rows; terms of gradient sub-tile 1        block 1: terms of gradient sub-tile 2
//   block 2: terms of gradient sub-tile 3 + activation sub-tiles 0, 1 of the NEXT step (its rows have landed: vmcnt(0) at the block start)
//   block 3: terms of gradient sub-tile 0 of the next step + activation sub-tiles 2, 3 of the next step
constexpr int kStageF4 = 2 * 16 * 64;      // f32x4 per wave: 2 buffers x (8 gradient + 8 activation rows) x 64 lanes

// NI = 2: the 128 x 64 tiles against the position encoding (the bulk of the narrow tiles).  The activation rows are still fetched 16 bytes
// per lane -- by the lanes' (m & 15): 16 lanes cover the 64 columns -- and lane n finds its two interleaved columns 2 n, 2 n + 1 in the
// slot of lane n / 2; one activation sub-tile is made per block (in blocks 2 and 3), a block is 12 MFMAs.
// DTILE: the gradient operand's plane is tile-major fp32 (nnr_layout.h: tile32_index; the three-term input-gradient kernel writes it that way).
// Its eight DMA instructions of a step then fetch 64-byte runs -- instruction j, lane i: feature quad 16 (j & 1) + (i & 15), samples
// 4 (j >> 1) + (i >> 4) of the step, four consecutive samples of a quad being 64 contiguous bytes of a block -- into LDS slot 64 j + i, an
// image in which lane (h, m)'s two samples of pair P sit 16 slots apart (one ds_read2st64_b32) and the 16 lanes a read services together hit
// 16 different bank groups, exactly as in the row-major image.  Which samples a lane supplies to the MFMAs is unchanged.  All 128 columns
// of a 4 x 4 tile's gradient operand are valid in every unit (wgrad_units), so this path has no clamped lanes.
template <int BIAS, int NI, bool DTILE, bool XTILE>      // BIAS: WgradJob::bias (compile-time: the d(bias) sums are one add per sample pair in the tiles that carry them)
__device__ __forceinline__ void wgrad_job_split(const WgradJob& jb, const WgradArgs& a, int lane, int ji, f32x4* stage) {
    constexpr int MI = 4;
    static_assert(NI == 4 || NI == 2, "activation sub-tiles per job");
    static_assert(!XTILE || NI == 4, "the tile-major activation image is laid out for whole 128-column tiles");
    constexpr int G = 6 * NI;                  // MFMAs (= gaps) per block
    static_assert(G >= 16, "block 3 issues the 16 DMA rows of the step after next, one per gap");
    const int half = lane >> 5, m = lane & 31;
    const int dp = a.plane_pitch[jb.d_plane], xp = a.plane_pitch[jb.x_plane];
    const bool dok = MI * m < jb.d_valid, xok = NI * m < jb.x_valid;
    // (wave-uniform row base in scalar registers) + (32-bit lane offset): no vector address arithmetic per DMA
    const char* const dg = reinterpret_cast<const char*>(a.ws + a.plane_off[jb.d_plane] + (DTILE ? (jb.d_col0 >> 3) * 256 : jb.d_col0));
    const char* const xg = reinterpret_cast<const char*>(a.ws + a.plane_off[jb.x_plane] + (XTILE ? (jb.x_col0 >> 3) * 256 : jb.x_col0));
    const int tlane = ((lane & 15) >> 1) * 1024 + (lane & 1) * 512 + (lane >> 4) * 16;      // tile-major: this lane's 16 bytes of a DMA instruction
    const int dlane = DTILE ? tlane : 4 * ((dok ? MI * m : 0) + 8 * half * dp);
    const int64_t d_chunk_bytes = 128 * (int64_t)dp, x_chunk_bytes = 128 * (int64_t)xp;       // tile-major: bytes per 32-sample chunk of the plane
    const int xlane = XTILE ? tlane : NI == 4 ? 4 * ((xok ? NI * m : 0) + 8 * half * xp) : 4 * (4 * (m & 15) + 8 * half * xp);
    // d(bias): which of a pair's two samples this tile sums (WgradJob::bias: 1 all, 2 / 3: the two tiles of a row block share the samples
    // -- here by the parity of s)

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float bsum[MI] = {0.f, 0.f, 0.f, 0.f};
    uint32_t Xc[NI][3][4], Xn[NI][3][4], Dq[2][3][4];   // current / next step's activation terms; [..][term: 0 = l, 1 = m, 2 = h][pair of samples]
    f32x2 fp[2][4];               // a pair's two values between its fetch and its split, then the residuals between the stages (adjacent: packed subtracts)
    const float* const lrow = reinterpret_cast<const float*>(stage) + 4 * lane;   // this lane's 4 floats of staged row r: lrow[256 r + c]
    // tile-major gradient image: sample 8 h + 2 P (+ 1) of quad m at float 4 (256 h + 64 (m >> 4) + (m & 15)) + 512 (P >> 1) + 128 (P & 1) (+ 64)
    const float* const lrowd = DTILE ? reinterpret_cast<const float*>(stage) + 4 * (256 * half + 64 * (m >> 4) + (m & 15)) : lrow;
    // the activation operand's columns of this lane in a staged row: its own slot (NI = 4), or half of the slot of lane m / 2 (NI = 2)
    // (tile-major: the same image as the gradient's, behind the eight gradient rows)
    const float* const lrowx = XTILE ? reinterpret_cast<const float*>(stage) + 4 * (256 * half + 64 * (m >> 4) + (m & 15)) + 2048
                             : NI == 4 ? lrow : reinterpret_cast<const float*>(stage) + 4 * (32 * half + (m >> 1)) + 2 * (m & 1);

// row S (0..7) of operand G (pitch P floats) of the step at sample KK -> staged row ROW of this wave
#define NNR_WDMA(G, LANE, P, KK, S, DST, ROW) \
    __builtin_amdgcn_global_load_lds((glb_ptr_t)((G) + ((KK) + (S)) * (int64_t)(P) * 4 + (LANE)), (lds_ptr_t)((DST) + (ROW) * 64), 16, 0, 0)
// (the conversion as the compiler's own: as inline asm -- pack_bf16 -- every use drags an s_nop along, 55 per step)
#define NNR_WPACK(V) __builtin_bit_cast(uint32_t, __builtin_convertvector(V, bf16x2))
#define NNR_WOFF(ROW0, P, C, SECOND) (((ROW0) == 0 ? DTILE : XTILE) ? 512 * ((P) >> 1) + 128 * ((P) & 1) + (C) + 64 * (SECOND) : 256 * ((ROW0) + 2 * (P) + (SECOND)) + (C))
#define NNR_WFETCH(W, P, LR, ROW0, C) (fp[W][P] = f32x2{(LR)[NNR_WOFF(ROW0, P, C, 0)], (LR)[NNR_WOFF(ROW0, P, C, 1)]})
// the gradient operand's DMA instruction S (0..7) of the step at sample KK: a staged row of the row-major plane, or 64-byte runs of the tile-major one
#define NNR_WDMA_D(KK, S, DST, ROW)                                                                                                      \
    do {                                                                                                                                 \
        if constexpr (DTILE)                                                                                                             \
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(dg + ((KK) >> 5) * d_chunk_bytes + ((KK) & 31) * 16 + ((S) & 1) * 8192 + ((S) >> 1) * 64 + dlane), \
                                             (lds_ptr_t)((DST) + (ROW) * 64), 16, 0, 0);                                                 \
        else NNR_WDMA(dg, dlane, dp, KK, S, DST, ROW);                                                                                    \
    } while (0)
#define NNR_WDMA_X(KK, S, DST, ROW)                                                                                                      \
    do {                                                                                                                                 \
        if constexpr (XTILE)                                                                                                             \
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(xg + ((KK) >> 5) * x_chunk_bytes + ((KK) & 31) * 16 + ((S) & 1) * 8192 + ((S) >> 1) * 64 + xlane), \
                                             (lds_ptr_t)((DST) + (ROW) * 64), 16, 0, 0);                                                 \
        else NNR_WDMA(xg, xlane, xp, KK, S, DST, ROW);                                                                                    \
    } while (0)
// the split of pair P of component C of the rows [ROW0, ROW0 + 8), stage ST: 0 fetch, 1 h, 2 m, 3 l; W = residual set, BI >= 0: d(bias) slot
#define NNR_WSPLIT(LR, ROW0, C, P, ST, Q, W, BI, BWT)                                        \
    do {                                                                                     \
        if ((ST) == 0) {                                                                     \
            NNR_WFETCH(W, P, LR, ROW0, C);                                                   \
        } else if ((ST) == 1) {                                                              \
            if ((BI) >= 0 && BIAS != 0)                                                      \
                bsum[(BI) >= 0 ? (BI) : 0] += (BWT) * (BIAS == 1 ? fp[W][P][0] + fp[W][P][1] : (BIAS == 2 ? fp[W][P][0] : fp[W][P][1])); \
            Q[2][P] = NNR_WPACK(fp[W][P]);                                   \
            fp[W][P] = pair_residual(fp[W][P], Q[2][P]);                                      \
        } else if ((ST) == 2) {                                                              \
            Q[1][P] = NNR_WPACK(fp[W][P]);                                   \
            fp[W][P] = pair_residual(fp[W][P], Q[1][P]);                                      \
        } else {                                                                             \
            Q[0][P] = NNR_WPACK(fp[W][P]);                                   \
        }                                                                                    \
    } while (0)
#define NNR_WSPLIT_ALL(LR, ROW0, C, Q, W, BI)                                                \
    _Pragma("unroll") for (int st_ = 0; st_ < 4; ++st_)                                      \
        _Pragma("unroll") for (int p_ = 0; p_ < 4; ++p_) NNR_WSPLIT(LR, ROW0, C, p_, st_, Q, W, BI, 1.f)

    // prologue: the first step's rows, all terms of the activation operand, the first gradient sub-tile
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        NNR_WDMA_D((int64_t)jb.k0, s, stage, s);
        NNR_WDMA_X((int64_t)jb.k0, s, stage, 8 + s);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    {   // the SECOND step's rows into the other buffer (round 4: the fetch runs two steps ahead, see the loop)
        const int64_t k1st = jb.k0 + 16 < jb.k1 ? jb.k0 + 16 : jb.k0;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            NNR_WDMA_D(k1st, s, stage + 1024, s);
            NNR_WDMA_X(k1st, s, stage + 1024, 8 + s);
        }
    }
#pragma unroll
    for (int j = 0; j < NI; ++j) NNR_WSPLIT_ALL(lrowx, 8, j, Xc[j], 1, -1);
    NNR_WSPLIT_ALL(lrowd, 0, 0, Dq[0], 0, 0);

    // One 16-sample step per iteration of ONE loop body (two textual copies for the two buffer parities made hipcc assign the 256
    // accumulators to different registers in the copies and shuffle them in between: hundreds of moves and spills per step); the staging
    // buffers alternate by address, and the next step's activation terms are copied over the current ones at the end of the step (48 moves).
    // Fetch distance (round 4): the rows of step k + 32 are requested in block 3 of step k, into the buffer step k itself used -- its last read
    // is the fetch of gradient sub-tile 3 in block 2 -- and are first needed in block 2 of step k + 16: THREE blocks (72 MFMAs, ~1.2 us) in
    // flight.  Round 3 requested step k + 16 in block 0 of step k for block 2 of the same step: two blocks, 0.8 us -- enough for rows that sit
    // in the L2 or the memory-side cache, not for HBM under load, which is where rows written with non-temporal stores come from.
    // kn = the step to prefetch: past the end a wave re-reads rows of its own range (valid data, no branch around the DMA); nf = 0 keeps
    // the step after the last one out of d(bias).
    for (int64_t k = jb.k0; k < jb.k1; k += 16) {
        const int par = (int)(((k - jb.k0) >> 4) & 1);
        const bool more = k + 16 < jb.k1;
        const int64_t kn = k + 32 < jb.k1 ? k + 32 : k;
        const float nf = more ? 1.f : 0.f;
        const float* const lc = lrowd + 4096 * par;           // this step's staged gradient rows, the next step's
        const float* const ln = lrowd + 4096 * (1 - par);
        const float* const lnx = lrowx + 4096 * (1 - par);
        f32x4* const dst = stage + 1024 * par;                // (free from block 3 on)
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            if (i == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the next step's rows have landed
            constexpr int kXOps = 16 * (NI / 2);                                  // activation operations of a block that makes NI / 2 sub-tiles
            const int n_ops = 16 + (i >= 2 ? kXOps : 0);
            const int stride = n_ops / 16;                                        // every stride-th operation is the gradient's
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const int j = g / 6, t = g % 6;
                const int wc = t == 0 ? 0 : (t < 3 ? 1 : 2), xc = t == 0 ? 2 : (t == 1 ? 1 : (t == 2 ? 2 : t - 3));
                __builtin_amdgcn_sched_barrier(0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                    __builtin_bit_cast(bf16x8, u32x4{Dq[i & 1][wc][0], Dq[i & 1][wc][1], Dq[i & 1][wc][2], Dq[i & 1][wc][3]}),
                    __builtin_bit_cast(bf16x8, u32x4{Xc[j][xc][0], Xc[j][xc][1], Xc[j][xc][2], Xc[j][xc][3]}),
                    acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                {
                    const int q = i == 3 ? g : 16;                                // the 16 rows of step k + 32, one per gap from the start of block 3
                    if (q < 8) NNR_WDMA_D(kn, q, dst, q);
                    else if (q < 16) NNR_WDMA_X(kn, q - 8, dst, q);
                }
#pragma unroll
                for (int o = 0; o < 48; ++o) {
                    if (o >= n_ops || (o * G) / n_ops != g) continue;
                    const bool is_d = stride == 1 || o % stride == 0;
                    const int od = stride == 1 ? o : o / stride, ox = o - o / stride - 1;   // index within the operand's own operations
                    if (is_d) {
                        const int st = od / 4, p = od % 4;                       // all pairs' fetch, then stages 1, 2, 3
                        if (i == 0) NNR_WSPLIT(lc, 0, 1, p, st, Dq[1], 0, 1, 1.f);
                        else if (i == 1) NNR_WSPLIT(lc, 0, 2, p, st, Dq[0], 0, 2, 1.f);
                        else if (i == 2) NNR_WSPLIT(lc, 0, 3, p, st, Dq[1], 0, 3, 1.f);
                        else NNR_WSPLIT(ln, 0, 0, p, st, Dq[0], 0, 0, nf);       // the NEXT step's sub-tile 0
                    } else {
                        const int cc = (NI / 2) * (i - 2) + ox / 16, r = ox % 16, st = r / 4, p = r % 4;
                        NNR_WSPLIT(lnx, 8, cc, p, st, Xn[cc], 1, -1, 0.f);
                    }
                }
            }
            // every tile stays in the accumulation registers, always: with any of them in VGPRs nothing else fits
            if constexpr (NI == 4) asm volatile("" : "+a"(acc[i][0]), "+a"(acc[i][1]), "+a"(acc[i][2]), "+a"(acc[i][3]));
            else asm volatile("" : "+a"(acc[i][0]), "+a"(acc[i][1]));
        }
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) Xc[j][t][q] = Xn[j][t][q];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the last step's prefetch writes LDS: let it finish before the next job reuses the area
#undef NNR_WDMA
#undef NNR_WDMA_D
#undef NNR_WDMA_X
#undef NNR_WOFF
#undef NNR_WSPLIT
#undef NNR_WFETCH
#undef NNR_WPACK
#undef NNR_WSPLIT_ALL

    float* slot = a.slots + (int64_t)ji * kSlotFloats;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int mr = (r & 3) + 8 * (r >> 2) + 4 * half;
            float* dst = slot + (MI * mr + i) * (32 * NI) + NI * m;
            if constexpr (NI == 4) *reinterpret_cast<f32x4*>(dst) = f32x4{acc[i][0][r], acc[i][1][r], acc[i][2][r], acc[i][3][r]};
            else *reinterpret_cast<f32x2*>(dst) = f32x2{acc[i][0][r], acc[i][1][r]};
        }
    if (jb.bias != 0) {      // (jobs without d(bias) run the BIAS = 1 instance and drop the sums: one instantiation less per shape)
        float* dst = slot + kSlotTile + half * (32 * MI) + MI * m;
#pragma unroll
        for (int i = 0; i < MI; ++i) dst[i] = bsum[i];
    }
}

// ---- the four 4 x 4 tiles of a D x D layer as ONE workgroup job (round 4): every operand value is split ONCE per workgroup ---------
// wgrad_job_split makes the terms of its two operand halves on its own: in a workgroup that owns the four tiles (a, b) of one layer over
// one sample range -- what the host plan gives every class-A workgroup -- each half is split twice, and the split is what these waves
// wait for (4.8 non-MFMA instructions per MFMA, one wave per SIMD hides about five).  Here wave (a, b) splits only HALF of each of its
// operands -- of the step's 16 samples the pairs 2 b, 2 b + 1 of gradient half a and the pairs 2 a, 2 a + 1 of activation half b, all four
// interleaved sub-tiles -- and the halves are exchanged through LDS:
//   staging   per wave 2 x 8 KiB: the 4 + 4 rows (of 2 x 4 samples) it splits, by LDS-DMA two steps ahead          64 KiB
//   exchange  2 buffers x 4 regions (gradient half 0 / 1, activation half 0 / 1) x [sub-tile 4][term 3][lane 64][16 B]   96 KiB
//             a lane's 16 bytes = its packed pairs 0 .. 3 of that sub-tile and term = the MFMA operand; the owner of pairs 2 q, 2 q + 1
//             writes bytes [8 q, 8 q + 8)
// Step k (16 samples), exchange buffer e = step parity: blocks 0 - 2 (72 MFMAs on the terms of step k read from buffer e) carry the split of
// step k + 1's rows and the writes of its terms into buffer 1 - e; ONE barrier; block 3 carries the reads of step k + 1's activation terms.
// Buffer 1 - e was last read (terms of step k - 1) before the previous step's barrier.  Per step and wave: 16 pair splits instead of 32,
// 24 + 24 LDS exchanges, 8 DMA rows instead of 16: ~2.3 non-MFMA instructions per MFMA.  The products, their order and the accumulators are
// those of wgrad_job_split (the weight gradients are bit-identical); d(bias) is summed by each wave over the samples it splits (the two
// tiles of a row block still hold one share each).  Every wave of the workgroup must run the same steps: the plan marks these jobs
// (WgradJob::reserved = 1) and gives the four waves of a class-A workgroup the same sample ranges.
constexpr int kCoopStageF4 = 2 * 8 * 64;                 // f32x4 per wave: 2 buffers x (4 gradient + 4 activation rows) x 64 lanes
constexpr int kCoopRegionF4 = 4 * 3 * 64;                // f32x4 per exchange region
constexpr int kCoopXchF4 = kWavesPerBlock * kCoopStageF4;   // the exchange buffers start behind the four staging areas
constexpr int kCoopF4 = kCoopXchF4 + 2 * 4 * kCoopRegionF4; // 160 KiB

template <bool DTILE, bool XTILE>
__device__ __forceinline__ void wgrad_group_split(const WgradJob& jb, const WgradArgs& a, int lane, int ji, f32x4* lds_all, int wave) {
    constexpr int MI = 4, NI = 4;
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    const int half = lane >> 5, m = lane & 31;
    const int ta = __builtin_amdgcn_readfirstlane(jb.d_col0 >> 7), tb = __builtin_amdgcn_readfirstlane(jb.x_col0 >> 7);   // this wave's tile
    const int dp = a.plane_pitch[jb.d_plane], xp = a.plane_pitch[jb.x_plane];
    f32x4* const stage = lds_all + wave * kCoopStageF4;
    // ---- DMA sources (all 128 columns of either half are valid in these units: wgrad_units) ----
    const char* const dg = reinterpret_cast<const char*>(a.ws + a.plane_off[jb.d_plane] + (DTILE ? (jb.d_col0 >> 3) * 256 : jb.d_col0));
    const char* const xg = reinterpret_cast<const char*>(a.ws + a.plane_off[jb.x_plane] + (XTILE ? (jb.x_col0 >> 3) * 256 : jb.x_col0));
    const int tlane = ((lane & 15) >> 1) * 1024 + (lane & 1) * 512 + (lane >> 4) * 16;
    const int dlane = DTILE ? tlane : 4 * (MI * m + 8 * half * dp);
    const int xlane = XTILE ? tlane : 4 * (NI * m + 8 * half * xp);
    const int64_t d_chunk_bytes = 128 * (int64_t)dp, x_chunk_bytes = 128 * (int64_t)xp;
    // gradient row r (0..3) of the step at sample KK -> staged row r; tile-major: the four DMA instructions that hold the samples 4 tb + 0..3 and
    // 8 + 4 tb + 0..3 of the step (instruction 2 (tb + 2 (r >> 1)) + (r & 1) of wgrad_job_split's eight); row-major: samples KK + 8 h + 4 tb + r
    auto dma_d = [&](int64_t KK, int r, f32x4* dst) __attribute__((always_inline)) {
        if constexpr (DTILE)
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(dg + (KK >> 5) * d_chunk_bytes + (KK & 31) * 16 + (r & 1) * 8192 + (tb + 2 * (r >> 1)) * 64 + dlane),
                                             (lds_ptr_t)(dst + r * 64), 16, 0, 0);
        else
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(dg + (KK + 4 * tb + r) * (int64_t)dp * 4 + dlane), (lds_ptr_t)(dst + r * 64), 16, 0, 0);
    };
    auto dma_x = [&](int64_t KK, int r, f32x4* dst) __attribute__((always_inline)) {      // samples KK + 8 h + 4 ta + r -> staged row 4 + r
        if constexpr (XTILE)
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(xg + (KK >> 5) * x_chunk_bytes + (KK & 31) * 16 + (r & 1) * 8192 + (ta + 2 * (r >> 1)) * 64 + xlane),
                                             (lds_ptr_t)(dst + (4 + r) * 64), 16, 0, 0);
        else
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(xg + (KK + 4 * ta + r) * (int64_t)xp * 4 + xlane), (lds_ptr_t)(dst + (4 + r) * 64), 16, 0, 0);
    };
    // ---- staged values of this lane: pair pl (0, 1) of the wave's four samples per half, component C ----
    const float* const sf = reinterpret_cast<const float*>(stage);
    const float* const lrd = DTILE ? sf + 4 * (128 * half + 64 * (m >> 4) + (m & 15)) : sf + 4 * lane;
    const float* const lrx = XTILE ? sf + 4 * (128 * half + 64 * (m >> 4) + (m & 15)) + 1024 : sf + 4 * lane;
    auto off_d = [](int pl, int C, int second) { return DTILE ? 128 * pl + C + 64 * second : 256 * (2 * pl + second) + C; };
    auto off_x = [](int pl, int C, int second) { return XTILE ? 128 * pl + C + 64 * second : 256 * (4 + 2 * pl + second) + C; };
    // ---- exchange addresses (bytes from lds_all) ----
    char* const xch = reinterpret_cast<char*>(lds_all + kCoopXchF4);
    constexpr int kBufBytes = 4 * kCoopRegionF4 * 16, kRegBytes = kCoopRegionF4 * 16;
    char* const rd_d = xch + ta * kRegBytes + lane * 16;                 // + buffer * kBufBytes + ((sub-tile * 3 + term) * 64) * 16
    char* const rd_x = xch + (2 + tb) * kRegBytes + lane * 16;
    char* const wr_d = rd_d + 8 * tb;                                      // this wave's pairs 2 tb, 2 tb + 1 of the gradient half
    char* const wr_x = rd_x + 8 * ta;

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float bsum[MI] = {0.f, 0.f, 0.f, 0.f};
    uint32_t Xc[NI][3][4], Dq[2][3][4];     // [sub-tile][term: 0 = l, 1 = m, 2 = h][pair]
    uint32_t T[4][3];                                     // the terms of the batch of four pairs being split: [pair in batch][term]
    f32x2 rr[4];

    auto pack = [](f32x2 v) { return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2)); };
    // pair q (0..15) = operand q >> 3 (0 gradient, 1 activation), component (q >> 1) & 3, local pair q & 1; stage 0 fetch, 1 h (+ d(bias)), 2 m, 3 l
    auto split_op = [&](int q, int st, const float* bd, const float* bx, float nf) __attribute__((always_inline)) {
        const int op = q >> 3, C = (q >> 1) & 3, pl = q & 1, s = q & 3;
        if (st == 0) {
            rr[s] = op == 0 ? f32x2{bd[off_d(pl, C, 0)], bd[off_d(pl, C, 1)]} : f32x2{bx[off_x(pl, C, 0)], bx[off_x(pl, C, 1)]};
        } else if (st == 1) {
            if (op == 0) bsum[C] += nf * (rr[s][0] + rr[s][1]);
            T[s][2] = pack(rr[s]);
            rr[s] = pair_residual(rr[s], T[s][2]);
        } else if (st == 2) {
            T[s][1] = pack(rr[s]);
            rr[s] = pair_residual(rr[s], T[s][1]);
        } else {
            T[s][0] = pack(rr[s]);
        }
    };
    // the three terms of component C of operand op (pairs (op, C, 0), (op, C, 1) = batch slots 2 (C & 1), 2 (C & 1) + 1) into exchange buffer eb
    auto write_op = [&](int op, int C, int t, int eb) __attribute__((always_inline)) {
        char* const p = (op == 0 ? wr_d : wr_x) + eb * kBufBytes + ((C * 3 + t) * 64) * 16;
        *reinterpret_cast<u32x2*>(p) = u32x2{T[2 * (C & 1)][t], T[2 * (C & 1) + 1][t]};
    };
    auto read_terms = [&](uint32_t (&dst)[3][4], const char* base, int sub, int eb) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(base + eb * kBufBytes + ((sub * 3 + t) * 64) * 16);
            dst[t][0] = v[0]; dst[t][1] = v[1]; dst[t][2] = v[2]; dst[t][3] = v[3];
        }
    };
    // the 88 split / write operations of a step in issue order: per batch of four pairs (two components of one operand) 4 fetches, 4 x h,
    // 4 x m, 4 x l, then the six writes of the two components
    auto coop_op = [&](int n, const float* bd, const float* bx, float nf, int eb) __attribute__((always_inline)) {
        const int B = n / 22, o = n % 22;
        if (o < 16) split_op(4 * B + (o & 3), o >> 2, bd, bx, nf);
        else write_op(B >> 1, 2 * (B & 1) + (o - 16) / 3, (o - 16) % 3, eb);
    };

    // ---- prologue: nobody may still read what this job is about to overwrite (the previous job's last exchange buffer, its staging) ----
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        dma_d(jb.k0, r, stage);
        dma_x(jb.k0, r, stage);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    {
        const int64_t k1st = jb.k0 + 16 < jb.k1 ? jb.k0 + 16 : jb.k0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            dma_d(k1st, r, stage + 512);
            dma_x(k1st, r, stage + 512);
        }
    }
#pragma unroll
    for (int n = 0; n < 88; ++n) coop_op(n, lrd, lrx, 1.f, 0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int j = 0; j < NI; ++j) read_terms(Xc[j], rd_x, j, 0);
    read_terms(Dq[0], rd_d, 0, 0);

    for (int64_t k = jb.k0; k < jb.k1; k += 16) {
        const int e = (int)(((k - jb.k0) >> 4) & 1);              // exchange buffer of this step's terms = staging buffer of this step's rows
        const bool more = k + 16 < jb.k1;
        const int64_t kn = k + 32 < jb.k1 ? k + 32 : k;           // rows to request (past the end: rows of the own range again, never used)
        const float nf = more ? 1.f : 0.f;
        const float* const bd = lrd + 2048 * (1 - e);             // the NEXT step's staged rows (requested a whole step ago)
        const float* const bx = lrx + 2048 * (1 - e);
        f32x4* const dst = stage + 512 * e;                       // this step's rows were split during the previous step: free
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the next step's rows have landed
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            if (i == 3) {       // every wave's terms of the next step are in buffer 1 - e, every wave's reads of this step's gradient terms are done
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
#pragma unroll
            for (int g = 0; g < 24; ++g) {
                const int j = g / 6, t = g % 6;
                const int wc = t == 0 ? 0 : (t < 3 ? 1 : 2), xc = t == 0 ? 2 : (t == 1 ? 1 : (t == 2 ? 2 : t - 3));
                __builtin_amdgcn_sched_barrier(0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                    __builtin_bit_cast(bf16x8, u32x4{Dq[i & 1][wc][0], Dq[i & 1][wc][1], Dq[i & 1][wc][2], Dq[i & 1][wc][3]}),
                    __builtin_bit_cast(bf16x8, u32x4{Xc[j][xc][0], Xc[j][xc][1], Xc[j][xc][2], Xc[j][xc][3]}),
                    acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (i < 3) {
                    const int gg = 24 * i + g;                     // gap 0..71 of the step's first three blocks
                    if (g == 1) read_terms(Dq[(i + 1) & 1], rd_d, i + 1, e);      // the gradient terms of the next block
                    if (i == 0 && g >= 4 && g < 12) {              // the rows of the step after next
                        if (g < 8) dma_d(kn, g - 4, dst);
                        else dma_x(kn, g - 8, dst);
                    }
#pragma unroll
                    for (int n = 0; n < 88; ++n)
                        if ((n * 72) / 88 == gg) coop_op(n, bd, bx, nf, 1 - e);
                } else {
                    // block 3 (behind the barrier): activation sub-tile j's last MFMA of the step is gap 6 j + 5 -- its registers take the
                    // next step's terms right behind it (no second register set, no 48 moves per step); the first gradient sub-tile last
                    if (t == 5) read_terms(Xc[j], rd_x, j, 1 - e);
                    if (g == 20) read_terms(Dq[0], rd_d, 0, 1 - e);         // (Dq[0] was last used by block 2)
                }
            }
            asm volatile("" : "+a"(acc[i][0]), "+a"(acc[i][1]), "+a"(acc[i][2]), "+a"(acc[i][3]));
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // the last prefetch writes LDS: let it finish before the area is reused

    float* slot = a.slots + (int64_t)ji * kSlotFloats;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int mr = (r & 3) + 8 * (r >> 2) + 4 * half;
            float* dst = slot + (MI * mr + i) * (32 * NI) + NI * m;
            *reinterpret_cast<f32x4*>(dst) = f32x4{acc[i][0][r], acc[i][1][r], acc[i][2][r], acc[i][3][r]};
        }
    if (jb.bias != 0) {
        float* dst = slot + kSlotTile + half * (32 * MI) + MI * m;
#pragma unroll
        for (int i = 0; i < MI; ++i) dst[i] = bsum[i];
    }
}

// ---- the same workgroup job with every product as THREE fp16 MFMA terms of two-term operands (NNR_F_SPLIT2, round 6) -------------------------
// d W = sum over ALL samples of Dlt x X: what matters for a term's precision is its absolute error against the SUM, so ONE power-of-two scale per
// operand plane does (the plane's largest magnitude -- tracked by the kernels that wrote the plane, WgradArgs::plane_max -- to [2^13, 2^14)).  A value
// v s has the terms h = fp16(v s) and m' = fp16((v s - h) 2^11): the residual carried at 2^11 as in nnr_split2.h, so that it stays a normal fp16
// number for values down to 2^-28 of the plane's largest (a whole ROW of d W may belong to a unit whose gradients are that small: with the
// plain residual its entries lost relative precision -- tests/test_gpu_layer_local.py saw it); its partner in the product is the OTHER operand's
// h 2^-11, made from h in registers (four packed multiplies per fragment, exact).  Products: m'_d (h_x 2^-11) + (h_d 2^-11) m'_x + h_d h_x, 2^-22
// relative each.
// Against wgrad_group_split: 48 MFMAs per step and wave instead of 96, two exchanged terms per value instead of three (128 KiB of LDS instead
// of 160), a two-stage split.  Staging, exchange protocol, barrier, DMA distance, d(bias) (from the unscaled fp32 values) and the slot format are
// wgrad_group_split's; the tile leaves in the terms' units, the reduction kernel multiplies the sum by 1 / (s_d s_x).
constexpr int kCoop2RegionF4 = 4 * 2 * 64;               // f32x4 per exchange region: [sub-tile 4][term 2][lane 64]
constexpr int kCoop2F4 = kCoopXchF4 + 2 * 4 * kCoop2RegionF4;

// the power of two that puts a plane's largest magnitude (bits of a non-negative float) into [2^13, 2^14); 1 for an empty plane
__device__ __forceinline__ float plane_scale(float mx) {
    const int eb = (int)((__float_as_uint(mx) >> 23) & 255u);
    int es = 267 - eb;
    es = es > 227 ? 227 : (es < 27 ? 27 : es);
    return mx > 0.f ? __uint_as_float((uint32_t)es << 23) : 1.f;
}

template <bool DTILE, bool XTILE>
__device__ __forceinline__ void wgrad_group_split2(const WgradJob& jb, const WgradArgs& a, int lane, int ji, f32x4* lds_all, int wave) {
    constexpr int MI = 4, NI = 4;
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
    typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
    const int half = lane >> 5, m = lane & 31;
    const int ta = __builtin_amdgcn_readfirstlane(jb.d_col0 >> 7), tb = __builtin_amdgcn_readfirstlane(jb.x_col0 >> 7);   // this wave's tile
    const int dp = a.plane_pitch[jb.d_plane], xp = a.plane_pitch[jb.x_plane];
    f32x4* const stage = lds_all + wave * kCoopStageF4;
    // the operands' scales: gradient planes P_DH1 + l at plane_max[8 + l] (P_DG at [16]), activation planes P_XH1 + l at [l]
    const float sd = plane_scale(a.plane_max[jb.d_plane == P_DG ? 16 : 8 + (jb.d_plane - P_DH1)]);
    const float sx = plane_scale(a.plane_max[jb.x_plane - P_XH1]);
    const char* const dg = reinterpret_cast<const char*>(a.ws + a.plane_off[jb.d_plane] + (DTILE ? (jb.d_col0 >> 3) * 256 : jb.d_col0));
    const char* const xg = reinterpret_cast<const char*>(a.ws + a.plane_off[jb.x_plane] + (XTILE ? (jb.x_col0 >> 3) * 256 : jb.x_col0));
    const int tlane = ((lane & 15) >> 1) * 1024 + (lane & 1) * 512 + (lane >> 4) * 16;
    const int dlane = DTILE ? tlane : 4 * (MI * m + 8 * half * dp);
    const int xlane = XTILE ? tlane : 4 * (NI * m + 8 * half * xp);
    const int64_t d_chunk_bytes = 128 * (int64_t)dp, x_chunk_bytes = 128 * (int64_t)xp;
    auto dma_d = [&](int64_t KK, int r, f32x4* dst) __attribute__((always_inline)) {
        if constexpr (DTILE)
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(dg + (KK >> 5) * d_chunk_bytes + (KK & 31) * 16 + (r & 1) * 8192 + (tb + 2 * (r >> 1)) * 64 + dlane),
                                             (lds_ptr_t)(dst + r * 64), 16, 0, 0);
        else
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(dg + (KK + 4 * tb + r) * (int64_t)dp * 4 + dlane), (lds_ptr_t)(dst + r * 64), 16, 0, 0);
    };
    auto dma_x = [&](int64_t KK, int r, f32x4* dst) __attribute__((always_inline)) {
        if constexpr (XTILE)
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(xg + (KK >> 5) * x_chunk_bytes + (KK & 31) * 16 + (r & 1) * 8192 + (ta + 2 * (r >> 1)) * 64 + xlane),
                                             (lds_ptr_t)(dst + (4 + r) * 64), 16, 0, 0);
        else
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(xg + (KK + 4 * ta + r) * (int64_t)xp * 4 + xlane), (lds_ptr_t)(dst + (4 + r) * 64), 16, 0, 0);
    };
    const float* const sf = reinterpret_cast<const float*>(stage);
    const float* const lrd = DTILE ? sf + 4 * (128 * half + 64 * (m >> 4) + (m & 15)) : sf + 4 * lane;
    const float* const lrx = XTILE ? sf + 4 * (128 * half + 64 * (m >> 4) + (m & 15)) + 1024 : sf + 4 * lane;
    auto off_d = [](int pl, int C, int second) { return DTILE ? 128 * pl + C + 64 * second : 256 * (2 * pl + second) + C; };
    auto off_x = [](int pl, int C, int second) { return XTILE ? 128 * pl + C + 64 * second : 256 * (4 + 2 * pl + second) + C; };
    char* const xch = reinterpret_cast<char*>(lds_all + kCoopXchF4);
    constexpr int kBufBytes = 4 * kCoop2RegionF4 * 16, kRegBytes = kCoop2RegionF4 * 16;
    char* const rd_d = xch + ta * kRegBytes + lane * 16;                 // + buffer * kBufBytes + ((sub-tile * 2 + term) * 64) * 16
    char* const rd_x = xch + (2 + tb) * kRegBytes + lane * 16;
    char* const wr_d = rd_d + 8 * tb;                                      // this wave's pairs 2 tb, 2 tb + 1 of the gradient half
    char* const wr_x = rd_x + 8 * ta;

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float bsum[MI] = {0.f, 0.f, 0.f, 0.f};
    uint32_t Xc[NI][3][4], Dq[2][3][4];     // [sub-tile][term: 0 = m' (the residual at 2^11), 1 = h, 2 = h 2^-11 (made here from h)][pair]
    auto down11 = [](uint32_t (&t)[3][4]) __attribute__((always_inline)) {      // (as ONE 8-wide multiply: see frag_down11, nnr_split2.h)
        const f16x8 v = __builtin_bit_cast(f16x8, u32x4{t[1][0], t[1][1], t[1][2], t[1][3]}) * (_Float16)0.00048828125f;
        const u32x4 u = __builtin_bit_cast(u32x4, v);
        t[2][0] = u[0]; t[2][1] = u[1]; t[2][2] = u[2]; t[2][3] = u[3];
    };
    uint32_t T[4][2];                                     // the terms of the batch of four pairs being split: [pair in batch][term]
    f32x2 rr[4];

    // pair q (0..15) = operand q >> 3 (0 gradient, 1 activation), component (q >> 1) & 3, local pair q & 1; stage 0 fetch, 1 scale + h (+ d(bias)), 2 m
    auto split_op = [&](int q, int st, const float* bd, const float* bx, float nf) __attribute__((always_inline)) {
        const int op = q >> 3, C = (q >> 1) & 3, pl = q & 1, s = q & 3;
        if (st == 0) {
            rr[s] = op == 0 ? f32x2{bd[off_d(pl, C, 0)], bd[off_d(pl, C, 1)]} : f32x2{bx[off_x(pl, C, 0)], bx[off_x(pl, C, 1)]};
        } else if (st == 1) {
            if (op == 0) bsum[C] += nf * (rr[s][0] + rr[s][1]);
            rr[s] = rr[s] * (op == 0 ? sd : sx);
            const f16x2 hh = __builtin_convertvector(rr[s], f16x2);
            T[s][1] = __builtin_bit_cast(uint32_t, hh);
            rr[s] = (rr[s] - __builtin_convertvector(hh, f32x2)) * 2048.f;
        } else {
            T[s][0] = __builtin_bit_cast(uint32_t, __builtin_convertvector(rr[s], f16x2));
        }
    };
    auto write_op = [&](int op, int C, int t, int eb) __attribute__((always_inline)) {
        char* const p = (op == 0 ? wr_d : wr_x) + eb * kBufBytes + ((C * 2 + t) * 64) * 16;
        *reinterpret_cast<u32x2*>(p) = u32x2{T[2 * (C & 1)][t], T[2 * (C & 1) + 1][t]};
    };
    auto read_terms = [&](uint32_t (&dst)[3][4], const char* base, int sub, int eb) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(base + eb * kBufBytes + ((sub * 2 + t) * 64) * 16);
            dst[t][0] = v[0]; dst[t][1] = v[1]; dst[t][2] = v[2]; dst[t][3] = v[3];
        }
    };      // (+ down11(dst) a few gaps later: right behind the loads it would wait out the LDS latency inside the MFMA stream)
    // the 64 split / write operations of a step in issue order: per batch of four pairs (two components of one operand) 4 fetches, 4 x h, 4 x m,
    // then the four writes of the two components
    auto coop_op = [&](int n, const float* bd, const float* bx, float nf, int eb) __attribute__((always_inline)) {
        const int B = n / 16, o = n % 16;
        if (o < 12) split_op(4 * B + (o & 3), o >> 2, bd, bx, nf);
        else write_op(B >> 1, 2 * (B & 1) + (o - 12) / 2, (o - 12) % 2, eb);
    };

    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // nobody may still read what this job is about to overwrite
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        dma_d(jb.k0, r, stage);
        dma_x(jb.k0, r, stage);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    {
        const int64_t k1st = jb.k0 + 16 < jb.k1 ? jb.k0 + 16 : jb.k0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            dma_d(k1st, r, stage + 512);
            dma_x(k1st, r, stage + 512);
        }
    }
#pragma unroll
    for (int n = 0; n < 64; ++n) coop_op(n, lrd, lrx, 1.f, 0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int j = 0; j < NI; ++j) { read_terms(Xc[j], rd_x, j, 0); down11(Xc[j]); }
    read_terms(Dq[0], rd_d, 0, 0);
    down11(Dq[0]);

    for (int64_t k = jb.k0; k < jb.k1; k += 16) {
        const int e = (int)(((k - jb.k0) >> 4) & 1);              // exchange buffer of this step's terms = staging buffer of this step's rows
        const bool more = k + 16 < jb.k1;
        const int64_t kn = k + 32 < jb.k1 ? k + 32 : k;           // rows to request (past the end: rows of the own range again, never used)
        const float nf = more ? 1.f : 0.f;
        const float* const bd = lrd + 2048 * (1 - e);             // the NEXT step's staged rows (requested a whole step ago)
        const float* const bx = lrx + 2048 * (1 - e);
        f32x4* const dst = stage + 512 * e;                       // this step's rows were split during the previous step: free
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the next step's rows have landed
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            if (i == 3) {       // every wave's terms of the next step are in buffer 1 - e, every wave's reads of this step's gradient terms are done
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
#pragma unroll
            for (int g = 0; g < 12; ++g) {
                // (the shifted-down h term of an operand is made a few gaps behind its loads and before its first use: X sub-tile 3's -- read in the
                // last gap of the previous step -- in gap 1 of block 0, first used by MFMA 9)
                if (i == 0 && g == 1 && k != jb.k0) down11(Xc[3]);
                const int j = g / 3, t = g % 3;
                const int wc = t == 0 ? 0 : (t == 1 ? 2 : 1), xc = t == 0 ? 2 : (t == 1 ? 0 : 1);      // (gradient term, activation term): (m', h 2^-11) (h 2^-11, m') (h, h) -- small products first
                __builtin_amdgcn_sched_barrier(0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
                    __builtin_bit_cast(f16x8, u32x4{Dq[i & 1][wc][0], Dq[i & 1][wc][1], Dq[i & 1][wc][2], Dq[i & 1][wc][3]}),
                    __builtin_bit_cast(f16x8, u32x4{Xc[j][xc][0], Xc[j][xc][1], Xc[j][xc][2], Xc[j][xc][3]}),
                    acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (i < 3) {
                    const int gg = 12 * i + g;                     // gap 0..35 of the step's first three blocks
                    if (g == 1) read_terms(Dq[(i + 1) & 1], rd_d, i + 1, e);      // the gradient terms of the next block
                    if (g == 7) down11(Dq[(i + 1) & 1]);
                    if (i == 0 && g >= 4 && g < 12) {              // the rows of the step after next
                        if (g < 8) dma_d(kn, g - 4, dst);
                        else dma_x(kn, g - 8, dst);
                    }
#pragma unroll
                    for (int n = 0; n < 64; ++n)
                        if ((n * 36) / 64 == gg) coop_op(n, bd, bx, nf, 1 - e);
                } else {
                    // block 3 (behind the barrier): activation sub-tile j's last MFMA of the step is gap 3 j + 2 -- its registers take the next step's
                    // terms right behind it; the first gradient sub-tile last
                    if (t == 2) read_terms(Xc[j], rd_x, j, 1 - e);
                    if (t == 2 && j > 0) down11(Xc[j - 1]);                   // (sub-tile j - 1's terms were read three gaps ago)
                    if (g == 3) read_terms(Dq[0], rd_d, 0, 1 - e);          // (Dq[0] was last used by block 2)
                    if (g == 9) down11(Dq[0]);
                }
            }
            asm volatile("" : "+a"(acc[i][0]), "+a"(acc[i][1]), "+a"(acc[i][2]), "+a"(acc[i][3]));
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // the last prefetch writes LDS: let it finish before the area is reused

    // the tile leaves in the TERMS' units (times s_d s_x): the reduction kernel, which adds the splits of a tile anyway, multiplies the sum by
    // 1 / (s_d s_x) (wgrad_reduce_kernel) -- multiplied here, hipcc read the whole tile out of the accumulators first: 256 VGPRs and scratch
    float* slot = a.slots + (int64_t)ji * kSlotFloats;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int mr = (r & 3) + 8 * (r >> 2) + 4 * half;
            float* dst = slot + (MI * mr + i) * (32 * NI) + NI * m;
            *reinterpret_cast<f32x4*>(dst) = f32x4{acc[i][0][r], acc[i][1][r], acc[i][2][r], acc[i][3][r]};
        }
    if (jb.bias != 0) {
        float* dst = slot + kSlotTile + half * (32 * MI) + MI * m;
#pragma unroll
        for (int i = 0; i < MI; ++i) dst[i] = bsum[i];
    }
}

#if defined(NNR_TIMELINE) && !defined(NNR_WGRAD_F16_TU)
extern "C" int nnr_timeline_wgrad(unsigned long long* host32) {
    return (int)hipMemcpyFromSymbol(host32, HIP_SYMBOL(tl_wgrad), 32 * sizeof(unsigned long long));
}
#endif

// ---- the 128 x 64 tiles against the position encoding as a PRIVATE two-term job (NNR_F_SPLIT2, round 6) -----------------------------------------
// Hidden 1 and the encoding columns of the skip layer: 33 of the 66 cost units the narrow tiles had on fp32 MFMAs (a 32x32x2 fp32 MFMA is 64 cycles:
// 4096 cycles per 16 samples of such a tile), i.e. half of what kept 30 % of the workgroups busy with 9 % of the MACs.  Here a wave takes the tile
// with the arithmetic of wgrad_group_split2 -- per-plane power-of-two scales (the encoding plane's largest magnitude at plane_max[17], written by the
// forward), v s = h + 2^-11 m', products m'_d (h_x 2^-11) + (h_d 2^-11) m'_x + h_d h_x -- but on its own: the gradient half belongs to this tile alone
// and the encoding operand is 8 pairs per step, nothing worth an exchange.  24 MFMAs of 32 cycles per step against 24 pair splits: the job is bound
// by the splits (~7 instructions per pair), and written plainly for that -- per step: wait for the rows (LDS-DMA, three buffers, two steps ahead),
// request the rows of the step after next, split everything, multiply.  Both planes are tile-major (nnr_layout.h); the gradient rows come in as in
// wgrad_job_split (eight instructions of 64-byte runs), the encoding rows as four: instruction j, lane i = quad (i & 15) of the 16, sample 4 j + (i >> 4).
// Slot format, flush and d(bias) (from the unscaled fp32 values, this half-wave's samples) are wgrad_job's for MI = 4, NI = 2; the tile leaves in
// the terms' units and the reduction kernel multiplies by 1 / (s_d s_x) (WgradJob::reserved = 2).
constexpr int kEncRows = 12;                                   // staged 16-byte rows per lane and step: 8 gradient + 4 encoding
constexpr int kEncStageF4 = 3 * kEncRows * 64;                 // f32x4 per wave: three buffers of 12 KiB
constexpr int kPlaneMaxEnc = 17;                               // WgradArgs::plane_max: the position-encoding plane's largest magnitude

__device__ __forceinline__ void wgrad_job_enc2(const WgradJob& jb, const WgradArgs& a, int lane, int ji, f32x4* stage) {
    constexpr int MI = 4, NI = 2;
    typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
    typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
    const int half = lane >> 5, m = lane & 31;
    const int dp = a.plane_pitch[jb.d_plane], xp = a.plane_pitch[jb.x_plane];
    const float sd = plane_scale(a.plane_max[8 + (jb.d_plane - P_DH1)]);
    const float sx = plane_scale(a.plane_max[kPlaneMaxEnc]);
    const char* const dg = reinterpret_cast<const char*>(a.ws + a.plane_off[jb.d_plane] + (jb.d_col0 >> 3) * 256);
    const char* const xg = reinterpret_cast<const char*>(a.ws + a.plane_off[jb.x_plane] + (jb.x_col0 >> 3) * 256);
    const int tlane = ((lane & 15) >> 1) * 1024 + (lane & 1) * 512 + (lane >> 4) * 16;      // quad (lane & 15): octet, half-octet; sample (lane >> 4)
    const int64_t d_chunk_bytes = 128 * (int64_t)dp, x_chunk_bytes = 128 * (int64_t)xp;
    auto dma_step = [&](int64_t KK, f32x4* dst) __attribute__((always_inline)) {
        const char* const dk = dg + (KK >> 5) * d_chunk_bytes + (KK & 31) * 16 + tlane;
        const char* const xk = xg + (KK >> 5) * x_chunk_bytes + (KK & 31) * 16 + tlane;
#pragma unroll
        for (int S = 0; S < 8; ++S)
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(dk + (S & 1) * 8192 + (S >> 1) * 64), (lds_ptr_t)(dst + S * 64), 16, 0, 0);
#pragma unroll
        for (int S = 0; S < 4; ++S)
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(xk + S * 64), (lds_ptr_t)(dst + (8 + S) * 64), 16, 0, 0);
    };
    // this lane's staged values (floats from the buffer's start): gradient column 4 m + C, encoding column 2 m + J; pair P = samples 8 half + 2 P (+ 1)
    const int base_d = 4 * (256 * half + 64 * (m >> 4) + (m & 15));
    const int base_x = 4 * (8 * 64) + 512 * half + 4 * (m >> 1) + 2 * (m & 1);
    auto off_d = [](int P, int C, int second) { return 512 * (P >> 1) + 128 * (P & 1) + 64 * second + C; };
    auto off_x = [](int P, int J, int second) { return 256 * (P >> 1) + 128 * (P & 1) + 64 * second + J; };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float bsum[MI] = {0.f, 0.f, 0.f, 0.f};
    auto down11 = [](u32x4 h) __attribute__((always_inline)) {      // (ONE 8-wide multiply: see frag_down11, nnr_split2.h)
        return __builtin_bit_cast(u32x4, __builtin_bit_cast(f16x8, h) * (_Float16)0.00048828125f);
    };
    // the two terms of a pair of the scaled operand
    auto split = [](f32x2 v, float sc, uint32_t& h, uint32_t& mm) __attribute__((always_inline)) {
        v = v * sc;
        const f16x2 hh = __builtin_convertvector(v, f16x2);
        h = __builtin_bit_cast(uint32_t, hh);
        mm = __builtin_bit_cast(uint32_t, __builtin_convertvector((v - __builtin_convertvector(hh, f32x2)) * 2048.f, f16x2));
    };

    const int64_t n_steps = (jb.k1 - jb.k0) >> 4;
    auto step_at = [&](int64_t t) { return jb.k0 + 16 * (t < n_steps ? t : n_steps - 1); };      // past the end: the last step's rows again (never used)
    // A lane's four gradient components of a sample are 16 contiguous bytes and consecutive lanes sit 16 bytes apart: ONE conflict-free ds_read_b128
    // per (pair, sample) instead of four 4-byte reads that collide four ways; its two encoding columns are 8 contiguous bytes (ds_read_b64).
    f32x4 dv[8];                      // raw gradient values of the step: [2 P + second][component]
    f32x2 ev[8];                      // raw encoding values: [2 P + second][column]
    u32x4 Eh[NI], Em[NI], Es[NI];     // the encoding operand's terms of the step being multiplied
    u32x4 Dh, Dm, Ds, Nh, Nm;         // gradient sub-tile i's terms; sub-tile i + 1's in the making
    auto load_rows = [&](int b) __attribute__((always_inline)) {
        const float* const sf = reinterpret_cast<const float*>(stage + b * (kEncRows * 64));
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            dv[q] = *reinterpret_cast<const f32x4*>(sf + base_d + off_d(q >> 1, 0, q & 1));
            ev[q] = *reinterpret_cast<const f32x2*>(sf + base_x + off_x(q >> 1, 0, q & 1));
        }
    };
    auto split_e = [&](int j, int P) __attribute__((always_inline)) {
        uint32_t h, mm;
        split(f32x2{ev[2 * P][j], ev[2 * P + 1][j]}, sx, h, mm);
        Eh[j][P] = h; Em[j][P] = mm;
    };
    auto split_d = [&](int i, int P, u32x4& H, u32x4& M, float wb = 1.f) __attribute__((always_inline)) {
        const f32x2 v = {dv[2 * P][i], dv[2 * P + 1][i]};
        bsum[i] += wb * (v[0] + v[1]);
        uint32_t h, mm;
        split(v, sd, h, mm);
        H[P] = h; M[P] = mm;
    };
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // (the wave's own previous job: its staging reads, its flush)
    dma_step(step_at(0), stage);
    dma_step(step_at(1), stage + kEncRows * 64);
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");                // step 0's rows
    load_rows(0);
#pragma unroll
    for (int j = 0; j < NI; ++j) {
#pragma unroll
        for (int P = 0; P < 4; ++P) split_e(j, P);
        Es[j] = down11(Eh[j]);
    }
#pragma unroll
    for (int P = 0; P < 4; ++P) split_d(0, P, Dh, Dm);
    Ds = down11(Dh);
    // Per step: block i = the 6 MFMAs of gradient sub-tile i; in their gaps the terms of sub-tile i + 1 (one pair per gap, then the 2^-11 copy) --
    // and under block 3 the NEXT step: its rows are waited for (they were requested a whole step ago), read, the encoding operand and the first
    // gradient sub-tile split.  The encoding terms being multiplied must survive block 3: the next step's go to a second set (En*).
    for (int64_t t = 0; t < n_steps; ++t) {
        const int b = (int)(t % 3);
        dma_step(step_at(t + 2), stage + ((b + 2) % 3) * (kEncRows * 64));      // (that buffer's last reader was step t - 1: its values are long in registers)
        u32x4 Enh[NI], Enm[NI];
#pragma unroll
        for (int i = 0; i < MI; ++i) {
#pragma unroll
            for (int g = 0; g < 6; ++g) {
                const int j = g / 3, tt = g % 3;
                __builtin_amdgcn_sched_barrier(0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, tt == 0 ? Dm : (tt == 1 ? Ds : Dh)),
                                                                 __builtin_bit_cast(f16x8, tt == 0 ? Es[j] : (tt == 1 ? Em[j] : Eh[j])), acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (i < 3) {
                    if (g < 4) split_d(i + 1, g, Nh, Nm);
                } else {
                    if (g == 0) {
                        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");      // everything but this step's request has landed: the next step's rows
                        load_rows((b + 1) % 3);
                    } else if (g < 5) {      // the next step's encoding terms: two pairs per gap
                        const int q = 2 * (g - 1);
                        uint32_t h, mm;
                        split(f32x2{ev[2 * (q & 3)][q >> 2], ev[2 * (q & 3) + 1][q >> 2]}, sx, h, mm);
                        Enh[q >> 2][q & 3] = h; Enm[q >> 2][q & 3] = mm;
                        split(f32x2{ev[2 * ((q + 1) & 3)][(q + 1) >> 2], ev[2 * ((q + 1) & 3) + 1][(q + 1) >> 2]}, sx, h, mm);
                        Enh[(q + 1) >> 2][(q + 1) & 3] = h; Enm[(q + 1) >> 2][(q + 1) & 3] = mm;
                    }
                }
            }
            asm volatile("" : "+a"(acc[i][0]), "+a"(acc[i][1]));
            if (i < 3) {
                Dh = Nh; Dm = Nm;
                Ds = down11(Dh);
            }
        }
        // behind the last MFMA of the step: the next step's first gradient sub-tile, and its encoding terms take over
        const float keep = t + 1 < n_steps ? 1.f : 0.f;      // (past the end the rows are the last step's again: keep them out of d(bias))
#pragma unroll
        for (int P = 0; P < 4; ++P) split_d(0, P, Dh, Dm, keep);
        Ds = down11(Dh);
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            Eh[j] = Enh[j]; Em[j] = Enm[j];
            Es[j] = down11(Eh[j]);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the prefetches past the end write LDS: let them finish before the area is reused

    float* slot = a.slots + (int64_t)ji * kSlotFloats;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int mr = (r & 3) + 8 * (r >> 2) + 4 * half;
            float* dst = slot + (MI * mr + i) * (32 * NI) + NI * m;
            *reinterpret_cast<f32x2*>(dst) = f32x2{acc[i][0][r], acc[i][1][r]};
        }
    if (jb.bias != 0) {
        float* dst = slot + kSlotTile + half * (32 * MI) + MI * m;
#pragma unroll
        for (int i = 0; i < MI; ++i) dst[i] = bsum[i];
    }
}

template <bool SPLIT, bool F16 = false>     // SPLIT: the 4 x 4 tiles with three-term products (wgrad_job_split); the narrow tiles stay on fp32 MFMAs
                                            // F16 (NNR_F_SPLIT2): the workgroup jobs with three fp16 terms per product (wgrad_group_split2) -- an
                                            // instantiation of its own: both workgroup jobs in one kernel cost 188 bytes of scratch per lane
__global__ __launch_bounds__(256, 1) void wgrad_kernel(WgradArgs a) {
    constexpr int kLdsBase = kCoopF4 > kWavesPerBlock * kStageF4 ? kCoopF4 : kWavesPerBlock * kStageF4;
    __shared__ __attribute__((aligned(16))) f32x4 stage_all[SPLIT ? (F16 && kWavesPerBlock * kEncStageF4 > kLdsBase ? kWavesPerBlock * kEncStageF4 : kLdsBase) : 1];
    const int lane = threadIdx.x & 63;
    const int wslot = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    // The plan blob is the caller's (a device buffer built by nnr_plan_build); the launch's counts come from a fresh host plan of the same
    // configuration.  The blob's trailer -- magic, jobs, waves, tiles -- must say the same, or the blob belongs to another ABI version, shape
    // or plan setting: trap (the launch fails loudly) instead of walking the job table with foreign indices.
    {
        const int32_t* const tr = a.heads + a.n_heads;
        if (tr[0] != kPlanMagic || tr[1] != a.n_jobs || tr[2] != a.n_waves || tr[3] != a.n_heads) __builtin_trap();
    }
    const int j0 = a.wave_first[wslot], j1 = a.wave_first[wslot + 1];
    // The gradients are OVERWRITTEN, not accumulated into: the weight tiles by the reduction kernel (every weight belongs to exactly one
    // tile), the bias rows by up to two atomic shares onto the zeros written here -- one workgroup's worth of stores instead of a
    // zero-fill launch over all 2.4 MB of gradients before this kernel.
    if (blockIdx.x == 0) {
        for (int l = 0; l < 13; ++l)
            for (int r = threadIdx.x; r < a.bias_rows[l]; r += 256) a.gb[l][r] = 0.f;
    }
#ifdef NNR_TIMELINE
    if (lane == 0 && wslot < 2048) tl_wgrad_all[2 * wslot] = __builtin_amdgcn_s_memtime();
#endif
    for (int ji = j0; ji < j1; ++ji) {   // 1 job per wave, 2 where the wave's span of the schedule crosses a tile boundary
        const WgradJob jb = a.jobs[ji];
        // bias reduction (MI VALU adds per k-step inside the MFMA stream) only in the jobs that own it
        const int key = __builtin_amdgcn_readfirstlane(jb.MI * 8 + jb.NI + 64 * jb.bias);
#define NNR_WGRAD_RUN(MI_, NI_, B_) wgrad_job<MI_, NI_, B_>(jb, a, lane, ji)
#define NNR_WGRAD_CASE(MI_, NI_)                                 \
    case MI_ * 8 + NI_: NNR_WGRAD_RUN(MI_, NI_, 0); break;        \
    case MI_ * 8 + NI_ + 64: NNR_WGRAD_RUN(MI_, NI_, 1); break;
        if constexpr (SPLIT) {
            // (the 128 x 64 tiles against the position encoding -- wgrad_job_split<.., 2> -- were measured on this path too: their VALU work
            // per MFMA is 1.6 times the 4 x 4 tile's and the kernel got SLOWER, 1.15 -> 1.29 ms at the best plan weight; they stay on fp32 MFMAs)
            if (__builtin_amdgcn_readfirstlane(jb.reserved) == 1) {      // a class-A workgroup: the layer's four tiles share the split (all four waves are here)
                if constexpr (F16)      // NNR_F_SPLIT2: three fp16 terms per product, scaled by the planes' maxima
                    wgrad_group_split2<kTileGradPlanes, kTileActPlanes>(jb, a, lane, ji, stage_all, __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)));
                else
                    wgrad_group_split<kTileGradPlanes, kTileActPlanes>(jb, a, lane, ji, stage_all, __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)));
                continue;
            }
            if constexpr (F16) {
                if (__builtin_amdgcn_readfirstlane(jb.reserved) == 2) {      // a 128 x 64 tile against the position encoding: private two-term job
                    wgrad_job_enc2(jb, a, lane, ji, stage_all + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) * kEncStageF4);
                    continue;
                }
            }
            if (__builtin_amdgcn_readfirstlane(jb.MI * 8 + jb.NI) == 4 * 8 + 4) {
                f32x4* const stage = stage_all + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) * kStageF4;
                // (every gradient plane of the three-term mode is tile-major: WsLayout::tiled -- the row-major instantiation is not built)
                switch (__builtin_amdgcn_readfirstlane(jb.bias)) {
                    case 2: wgrad_job_split<2, 4, kTileGradPlanes, kTileActPlanes>(jb, a, lane, ji, stage); break;
                    case 3: wgrad_job_split<3, 4, kTileGradPlanes, kTileActPlanes>(jb, a, lane, ji, stage); break;
                    default: wgrad_job_split<1, 4, kTileGradPlanes, kTileActPlanes>(jb, a, lane, ji, stage); break;     // 1, and 0 (sums dropped)
                }
                continue;
            }
        }
        switch (key) {
            case 4 * 8 + 4 + 128: NNR_WGRAD_RUN(4, 4, 2); break;
            case 4 * 8 + 4 + 192: NNR_WGRAD_RUN(4, 4, 3); break;
            NNR_WGRAD_CASE(4, 4)
            NNR_WGRAD_CASE(4, 2)
            NNR_WGRAD_CASE(4, 1)
            NNR_WGRAD_CASE(2, 4)
            NNR_WGRAD_CASE(2, 1)
            NNR_WGRAD_CASE(1, 4)
            NNR_WGRAD_CASE(1, 2)
            default: break;
        }
#undef NNR_WGRAD_CASE
#undef NNR_WGRAD_RUN
    }
#ifdef NNR_TIMELINE
    if (lane == 0 && wslot < 2048) tl_wgrad_all[2 * wslot + 1] = __builtin_amdgcn_s_memtime();
#endif
}

// The fp16-term instantiation lives in a translation unit of its own (this file compiled with -DNNR_WGRAD_F16_TU, csrc/build.py): the three
// instantiations of wgrad_kernel in one unit took hipcc eleven minutes.
hipError_t launch_wgrad_f16_main(const WgradArgs& a, hipStream_t st);
#ifdef NNR_WGRAD_F16_TU
hipError_t launch_wgrad_f16_main(const WgradArgs& a, hipStream_t st) {
    hipLaunchKernelGGL((wgrad_kernel<true, true>), dim3(a.n_waves / 4), dim3(256), 0, st, a);
    return hipGetLastError();
}
#else
// dW[tile] = sum over the tile's splits of their partial slots (every weight belongs to exactly one tile: a plain store, the caller's
// buffer needs no zero-fill); d(bias) += its share onto the zeros the main kernel wrote (at most two shares per row: a + b == b + a,
// the result does not depend on their order).  Blocks of jobs that are not split 0 of their tile exit at once; split 0 walks the chain.
constexpr int kMaxChain = 256;   // splits of one tile (the plan builder cuts a tile's sample range into far fewer)
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(WgradArgs a) {
    const int ji = a.heads[blockIdx.x >> 4];   // 16 blocks x 256 threads x one float4 = a 128 x 128 tile; one group of 16 per TILE (round 4:
                                               // launched per job -- 17 000 workgroups of which 16 in 17 returned at once -- this took 24 us)
    const WgradJob jb = a.jobs[ji];
    // the tile's chain of splits, walked ONCE per block into LDS: with every thread walking it, each partial slot was fetched behind a
    // dependent load of the job table (two serialised L2 round trips per split); now the slot loads of all splits are independent
    // (27.8 -> 24.5 us at 1024 x 192)
    __shared__ int chain[kMaxChain];
    __shared__ int n_chain;
    if (threadIdx.x == 0) {
        int n = 0, j = ji;
        for (; j >= 0 && n < kMaxChain; j = a.jobs[j].next_split) chain[n++] = j - ji;
        if (j >= 0) __builtin_trap();      // a longer chain than the plan builder can produce: never sum a truncated one
        n_chain = n;
    }
    __syncthreads();
    const int n = n_chain;
    const int t = (blockIdx.x & 15) * 256 + threadIdx.x;
    const int pitch = 32 * jb.NI, f4_per_row = 8 * jb.NI;
    const int row = t / f4_per_row, c0 = 4 * (t - row * f4_per_row);
    if (row < 32 * jb.MI && row < jb.d_valid && jb.row0 + row < jb.rows_real) {
        f32x4 sum = {0.f, 0.f, 0.f, 0.f};
        const float* src = a.slots + (int64_t)ji * kSlotFloats + row * pitch + c0;
        int s = 0;
        for (; s + 8 <= n; s += 8) {      // eight slots in flight, added in chain (= sample) order
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4*>(src + (int64_t)chain[s + u] * kSlotFloats);
#pragma unroll
            for (int u = 0; u < 8; ++u) sum += v[u];
        }
        for (; s < n; ++s) sum += *reinterpret_cast<const f32x4*>(src + (int64_t)chain[s] * kSlotFloats);
        if (a.plane_max != nullptr && jb.reserved == 2) {      // ... and the private two-term jobs against the position encoding (wgrad_job_enc2)
            const float sd = plane_scale(a.plane_max[8 + (jb.d_plane - P_DH1)]), sx = plane_scale(a.plane_max[kPlaneMaxEnc]);
            sum = (sum * (1.f / sd)) * (1.f / sx);
        }
        if (a.plane_max != nullptr && jb.reserved == 1) {      // NNR_F_SPLIT2: the workgroup jobs' tiles carry their operands' scales (wgrad_group_split2): exact powers of two
            const float sd = plane_scale(a.plane_max[jb.d_plane == P_DG ? 16 : 8 + (jb.d_plane - P_DH1)]), sx = plane_scale(a.plane_max[jb.x_plane - P_XH1]);
            sum = (sum * (1.f / sd)) * (1.f / sx);      // (two exact steps: s_d s_x itself may leave the float range when both planes are tiny)
        }
        float* dst = a.gw[jb.layer] + (int64_t)(jb.row0 + row) * jb.ldw + jb.wcol0 + c0;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (c0 + e < jb.x_valid && jb.wcol0 + c0 + e < jb.cols_real) dst[e] = sum[e];
    }
    if (jb.bias && (blockIdx.x & 15) == 0 && (int)threadIdx.x < 32 * jb.MI) {
        const int r = threadIdx.x;
        if (r < jb.d_valid && jb.row0 + r < jb.rows_real) {
            float sum = 0.f;
            const float* src = a.slots + (int64_t)ji * kSlotFloats + kSlotTile + r;
            for (int s = 0; s < n; ++s) {
                const float* p = src + (int64_t)chain[s] * kSlotFloats;
                sum += p[0] + p[32 * jb.MI];
            }
            atomicAdd(a.gb[jb.layer] + jb.row0 + r, sum);   // two tiles of a row block may each hold a share
        }
    }
}

// Un-merge (nnr_layout.h): from dW' (D/2 x D) and db' of the merged matrix W' = Wg1 Wf, b' = Wg1 bf + bg, with Wg1 = Wg[:, :D]:
//   dWf = Wg1^T dW'      dWg[:, :D] = dW' Wf^T + db' bf^T      dbf = Wg1^T db'      dbg = db'      (overwritten, like every gradient)
// Two (D x D/2 x D) products per step, 0.03 % of the MFMA work of the pass: plain VALU dot products, one output per thread.
template <int D, int BF16>   // BF16 = the MODE of Layout<D, MODE>: where the merge area of the packed buffer sits
__global__ __launch_bounds__(256) void wgrad_unmerge_kernel(WgradArgs a) {
    using L = Layout<D, BF16>;
    constexpr int Dh = L::Dh, ldg = D + kDirReal;
    const float* Wf = a.packed + L::copy_wf_off;    // [D][D]
    const float* Wg1 = a.packed + L::copy_wg_off;   // [Dh][D]
    const float* bf = a.packed + L::copy_bf_off;
    const float* dWm = a.gw[kMergedLayer];          // [Dh][D]
    const float* dbm = a.gb[kMergedLayer];
    const int gid = blockIdx.x * 256 + threadIdx.x;
    if (gid < D * D) {                               // dWf[j][k] = sum_m Wg1[m][j] dW'[m][k]
        const int j = gid / D, k = gid - j * D;
        float acc = 0.f;
#pragma unroll 16   // latency-bound dot products: keep 16 pairs of loads in flight (same fma chain)
        for (int m = 0; m < Dh; ++m) acc = fmaf(Wg1[m * D + j], dWm[m * D + k], acc);
        a.gw[9][gid] = acc;
    } else if (gid < D * D + Dh * D) {               // dWg[m][j] = sum_k dW'[m][k] Wf[j][k] + db'[m] bf[j]
        const int t = gid - D * D, m = t / D, j = t - m * D;
        float acc = dbm[m] * bf[j];
#pragma unroll 16
        for (int k = 0; k < D; ++k) acc = fmaf(dWm[m * D + k], Wf[j * D + k], acc);
        a.gw[10][m * ldg + j] = acc;
    } else if (gid < D * D + Dh * D + D) {           // dbf[j] = sum_m Wg1[m][j] db'[m]
        const int j = gid - D * D - Dh * D;
        float acc = 0.f;
        for (int m = 0; m < Dh; ++m) acc = fmaf(Wg1[m * D + j], dbm[m], acc);
        a.gb[9][j] = acc;
    } else if (gid < D * D + Dh * D + D + Dh) {      // dbg = db'
        const int m = gid - D * D - Dh * D - D;
        a.gb[10][m] = dbm[m];
    }
}

hipError_t launch_wgrad(const WgradArgs& a, hipStream_t st) {
    hipError_t e;
    prof_before(PROF_WGRAD, st);
    static const bool split_off = std::getenv("NNR_WGRAD_FP32") != nullptr;      // experiments: fp32 MFMAs in the weight gradient of the three-term mode
    static const bool f16_off = std::getenv("NNR_WGRAD_BF16_TERMS") != nullptr;      // experiments / A-B: the six-term workgroup jobs in the two-term mode
    WgradArgs b = a;
    if (!(a.bf16 == 3 && a.plane_max != nullptr && !split_off && !f16_off)) b.plane_max = nullptr;      // (non-null = "the workgroup jobs' tiles carry scales": the reduction undoes them)
    if (b.plane_max != nullptr) (void)launch_wgrad_f16_main(b, st);
    else if (a.bf16 >= 2 && !split_off) hipLaunchKernelGGL(wgrad_kernel<true>, dim3(a.n_waves / 4), dim3(256), 0, st, b);
    else hipLaunchKernelGGL(wgrad_kernel<false>, dim3(a.n_waves / 4), dim3(256), 0, st, b);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(a.n_heads * 16), dim3(256), 0, st, b);
    e = hipGetLastError();
    if (e == hipSuccess) e = launch_wgrad_unmerge(a, st);
    prof_after(PROF_WGRAD, st);      // the bracket covers the whole stage: main kernel + slot reduction + un-merge
    return e;
}

hipError_t launch_wgrad_unmerge(const WgradArgs& a, hipStream_t st) {
    const int threads = a.D * a.D + (a.D / 2) * a.D + a.D + a.D / 2;
    const dim3 grid((threads + 255) / 256), block(256);
    if (a.D == 256) {
        if (a.bf16 == 3) hipLaunchKernelGGL((wgrad_unmerge_kernel<256, 3>), grid, block, 0, st, a);
        else if (a.bf16 == 2) hipLaunchKernelGGL((wgrad_unmerge_kernel<256, 2>), grid, block, 0, st, a);
        else if (a.bf16 == 1) hipLaunchKernelGGL((wgrad_unmerge_kernel<256, 1>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((wgrad_unmerge_kernel<256, 0>), grid, block, 0, st, a);
    } else {
        if (a.bf16 == 3) hipLaunchKernelGGL((wgrad_unmerge_kernel<128, 3>), grid, block, 0, st, a);
        else if (a.bf16 == 2) hipLaunchKernelGGL((wgrad_unmerge_kernel<128, 2>), grid, block, 0, st, a);
        else if (a.bf16 == 1) hipLaunchKernelGGL((wgrad_unmerge_kernel<128, 1>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((wgrad_unmerge_kernel<128, 0>), grid, block, 0, st, a);
    }
    return hipGetLastError();
}

#endif      // NNR_WGRAD_F16_TU

}  // namespace nnr
