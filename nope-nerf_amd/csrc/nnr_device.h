// nnr_device.h -- device-side building blocks shared by the fused MLP kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include "nnr_layout.h"

namespace nnr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// D(32x32) += A(32x2) * B(2x32), exact fp32 (v_mfma_f32_32x32x2_f32): lane l supplies A[l&31][l>>5] and B[l>>5][l&31].
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// ---- profiling builds only.  -DNNR_ABLATE=<bits> switches parts of the MLP kernels off to price them (results are NOT valid; the product
// build has it 0): 1 = no side work between the MFMAs (epilogue units, splits, stash stores), 2 = no LDS-DMA of weight panels, 4 = no
// waits / barriers on the panel pipe, 8 = no stash stores.  The other experiment switches of rounds 2-6 (rejected variants and their A/B
// macros) are out of the sources: tools/experiments/README.md lists them with the commit that last carried each.
#ifndef NNR_ABLATE
#define NNR_ABLATE 0
#endif
// -DNNR_TIMELINE: shader-clock stamps of one mid-grid wave, read back with nnr_timeline_*
#ifdef NNR_TIMELINE
#define NNR_TL_DECL(name) __device__ unsigned long long name[32];
#define NNR_STAMP(name, i)                                                                      \
    do {                                                                                        \
        if ((blockIdx.x == 700 || (gridDim.x < 700 && blockIdx.x == 700 / 8)) && threadIdx.x == 0) name[i] = __builtin_amdgcn_s_memtime();     \
    } while (0)
#else
#define NNR_TL_DECL(name)
#define NNR_STAMP(name, i)
#endif

// ---- weight panels through LDS ---------------------------------------------------------------------------------------
// The four waves of a workgroup consume the same packed-weight stream, one 32 KiB panel (32 fragments = 128 MFMAs per
// wave = 8192 cycles) at a time.  Panels are DMA'd global -> LDS (global_load_lds_dwordx4: 1 KiB per wave-instruction,
// lane-linear, no VGPR round trip) into a ring of three buffers, two panels ahead; MFMA issue then depends only on
// ds_read_b128 (lgkmcnt), never on vmcnt -- which on CDNA4 also counts the stash *stores* interleaved into the stream and
// would otherwise stall every k-group behind HBM write latency.  Per panel: one counted s_waitcnt + one s_barrier.
constexpr int kNBuf = 3;
constexpr int kPanelF4 = kPanelFrags * 64;  // float4 elements per panel

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

// W = waves of the workgroup that share the stream; each copies PW = 32 / W of a panel's 32 fragments (8 with the usual 4 waves).
// F = fragment slots (KiB) per panel: 32, or 24 in the three-term mode (nnr_layout.h, MODE 2).
// STASH_TILE_: see kStashTile below.
template <int W, int F_ = kPanelFrags, bool STASH_TILE_ = false>
struct PanelPipeT {
    static constexpr int F = F_, F4 = F_ * 64;   // fragment slots / float4 elements per panel
    static constexpr int PW = F / W;             // DMA pieces (1 KiB) per wave and panel
    static_assert(F % W == 0 && PW >= 2 && PW <= 8 && PW % 2 == 0, "piece() addresses the pieces as immediates -PW/2 .. PW/2-1 around the middle one");
    const f32x4* src;  // stream base in global memory offset by this wave's fragment slice (wave-uniform: SGPRs)
    f32x4* lds;        // base of the three panel buffers in LDS
    int wave, lane;    // wave is wave-uniform (readfirstlane'd by the caller)
    int n_panels;      // panels in the stream
    // A workgroup may run the stream several times in a row (one pass per 32-sample chunk of its rays, see mlp_fwd_kernel): the
    // stream then WRAPS -- while the last panels of one pass are consumed, the first panels of the next pass are already on their
    // way, so only the first pass of a workgroup waits for weights.  A panel index p >= n_panels names panel p - n_panels of the
    // next pass; `phase` = (panels of the earlier passes) % kNBuf keeps the ring position running across passes.
    int phase = 0;
    bool more = false;   // another pass follows this one
    // three-term kernels (nnr_split.h): stash stores the part BEFORE the next one certainly issued after the last DMA piece of that
    // part's first panel (set by the kernel right before the call: a literal, folded after inlining; see enter<EXTRA>)
    int part_pre = 0;
    // three-term training forward (nnr_split.h): while gates_on, the part that splits a hidden vector into its bf16 terms also leaves that
    // vector's ReLU gate bits (register r -> bit r & 31 of gw[r >> 5]) -- the packed h term is zero exactly where the activation is
    bool gates_on = false;
    mutable uint32_t gw[4] = {0, 0, 0, 0};
    // three-term input-gradient kernel (nnr_split.h): the stash planes are tile-major fp32 (nnr_layout.h, tile32_index) -- the part's
    // `stash` argument is then the block of (this wave's chunk, first octet of the part's input) + 16 bytes per lane, every store one
    // contiguous non-temporal 1 KiB block.  A property of the pipe's TYPE: as a run-time member the two store flavours sat in the two arms
    // of a branch, the branch folded -- and the arms were merged first, which dropped the non-temporal hint from the surviving store.
    static constexpr bool kStashTile = STASH_TILE_;

    __device__ __forceinline__ int buffer(int p) const {   // ring slot of panel p of the current pass (p compile-time in the callers)
        const int b = p % kNBuf + phase;
        return b >= kNBuf ? b - kNBuf : b;
    }
    __device__ __forceinline__ void next_pass(bool more_after) {
        phase = (phase + n_panels % kNBuf) % kNBuf;
        more = more_after;
    }

    // This wave copies fragments [8*wave, 8*wave+8) of panel p into its ring slot as 8 DMA "pieces" of 1 KiB, always
    // exactly 8 per panel -- the counted wait below relies on it.  With a uniform base the address is
    // SGPR base + lane*16 and the LDS destination (M0) is scalar: no VALU work per piece.
    __device__ __forceinline__ void piece(int p, int i) const {
#if NNR_ABLATE & 2
        return;
#endif
        // One address pair per PANEL (global: VGPRs, LDS: M0), both pointing at piece 4; the piece is selected by the
        // instruction's signed 13-bit immediate, which offsets the global and the LDS address alike -- so a piece costs
        // one VMEM issue and no address arithmetic.
        const int ps = p < n_panels ? p : p - n_panels;
        const f32x4* g = src + (int64_t)ps * F4 + (PW / 2) * 64 + lane;
        f32x4* l = lds + buffer(p) * F4 + wave * (PW * 64) + (PW / 2) * 64;
        switch (i - PW / 2) {   // the offset operand must be a literal
            case -4: __builtin_amdgcn_global_load_lds((glb_ptr_t)g, (lds_ptr_t)l, 16, -4096, 0); break;
            case -3: __builtin_amdgcn_global_load_lds((glb_ptr_t)g, (lds_ptr_t)l, 16, -3072, 0); break;
            case -2: __builtin_amdgcn_global_load_lds((glb_ptr_t)g, (lds_ptr_t)l, 16, -2048, 0); break;
            case -1: __builtin_amdgcn_global_load_lds((glb_ptr_t)g, (lds_ptr_t)l, 16, -1024, 0); break;
            case 0: __builtin_amdgcn_global_load_lds((glb_ptr_t)g, (lds_ptr_t)l, 16, 0, 0); break;
            case 1: __builtin_amdgcn_global_load_lds((glb_ptr_t)g, (lds_ptr_t)l, 16, 1024, 0); break;
            case 2: __builtin_amdgcn_global_load_lds((glb_ptr_t)g, (lds_ptr_t)l, 16, 2048, 0); break;
            default: __builtin_amdgcn_global_load_lds((glb_ptr_t)g, (lds_ptr_t)l, 16, 3072, 0); break;
        }
    }
    // pieces [first, first+count) of panel p (nothing past the end of the LAST pass's stream; p is wave-uniform)
    __device__ __forceinline__ void pieces(int p, int first, int count) const {
        if (p < n_panels || more) {
#pragma unroll
            for (int i = first; i < first + count && i < PW; ++i) piece(p, i);
        }
    }
    // Make panel p readable and release the buffer of panel p-1 (which the pieces of panel p+2 overwrite; the caller
    // issues them AFTER this call, spread over the k-groups of panel p -- see gemm_part).
    //   vmcnt(8): everything older than the 8 youngest VMEM ops of this wave is complete.  The 8 pieces of panel p+1 were
    //   issued after those of panel p, so panel p has landed (any stores issued since only make the wait more conservative).
    //   The barrier then tells every wave that (a) all four slices of panel p are in LDS and (b) everybody is done reading
    //   panel p-1.
    // EXTRA: VMEM ops (stash stores) this wave certainly issued, in addition to the 8 pieces of panel p+1, after the last
    // piece of panel p -- they may stay in flight too (a store's completion is the L2's write acknowledgement; waiting for
    // recent ones here stalls the matrix pipe behind HBM write latency).
    template <int EXTRA = 0>
    __device__ __forceinline__ void enter(int p) const {
#if NNR_ABLATE & 4
        return;
#endif
        static_assert(PW + EXTRA < 64, "vmcnt is a 6-bit field");
        // lgkmcnt(0): this wave's ds_reads of panel p-1 have returned before it reports "done reading" at the barrier
        if (p + 1 < n_panels || more)
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(PW + EXTRA) : "memory");
        else
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    __device__ __forceinline__ void start() const {  // prologue of a workgroup's FIRST pass: two panels in flight
        pieces(0, 0, PW);
        pieces(1, 0, PW);
    }
};
using PanelPipe = PanelPipeT<kWavesPerBlock>;

// One k-group of A fragments (MT x 16 bytes per lane) in registers.
template <int MT>
struct Frags { f32x4 v[MT]; };

// k-groups of a part that live in its panel number pi, and DMA pieces issued per k-group slot of that panel
template <int KT, int MT>
__device__ __forceinline__ constexpr int panel_groups(int pi) {
    return (4 * KT - pi * part_gp(MT)) < part_gp(MT) ? (4 * KT - pi * part_gp(MT)) : part_gp(MT);
}
template <int KT, int MT>
__device__ __forceinline__ constexpr int panel_ppk(int pi) { return (8 + panel_groups<KT, MT>(pi) - 1) / panel_groups<KT, MT>(pi); }

// Enter the first panel of a layer part (KT x MT tiles), start the DMA of the panel two ahead and fetch the part's first
// k-group.
template <int KT, int MT>
__device__ __forceinline__ Frags<MT> gemm_open(const PanelPipe& pipe, int p0) {
    pipe.enter(p0);
    pipe.pieces(p0 + 2, 0, panel_ppk<KT, MT>(0));
    const f32x4* buf = pipe.lds + pipe.buffer(p0) * kPanelF4 + pipe.lane;
    Frags<MT> f;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) f.v[mt] = buf[mt * 64];
    return f;
}

// acc[mt] += A_part[32*mt.., :] * in   for one layer part whose packed panels start at stream panel p0.
//   in    : 16*KT registers in fragment layout (this wave's 32 samples)
//   stash : optional (sample, feature) row-major destination of `in` (row of this lane's sample, + 4*half): the four
//           registers consumed by k-group g are features 8g+4h..+3, i.e. one 16-byte store per k-group, issued *inside*
//           the MFMA stream.  Stashing a layer's input here -- instead of its output in an epilogue burst -- spreads the
//           10 KB/sample of training stash evenly over the kernel.
//   side  : optional VALU work hidden under this part's MFMAs: NSIDE units side(0..NSIDE-1), PPG per k-group, starting
//           at k-group SHIFT.  The MLP kernels use it to finish the PREVIOUS pass's accumulators (bias is already in
//           them; ReLU + sign bit, or the ReLU' select) one register pair per unit.  Two patterns:
//             SHIFT = 0: the units write registers of `in` that this part reads later (pair u -> registers >= 2u of the
//                        upper half, first needed at k-group >= 2*KT) -- "finish the other half while starting";
//             SHIFT = 1: the units OVERWRITE `in` behind the read pointer (k-group g rewrites registers 4(g-1)..4g-1,
//                        last read by k-group g-1) -- the new half-A vector replaces the old input in place.
//           Units that do not fit ((G - SHIFT) * PPG < NSIDE) run after the last k-group.
// Software pipeline, pinned with sched_barrier(0) (left alone, hipcc sinks every ds_read to just before its first use
// and then waits lgkmcnt(0) with the matrix pipe idle, and lumps the VALU / VMEM work where it stalls MFMA issue).
// A k-group is 4*MT MFMAs; with one wave per SIMD the wave has ~16 issue slots per 64-cycle fp32 MFMA, but a VMEM
// instruction that touches 32 lines or a handful of VALU ops each take most of one such gap -- so the non-MFMA work of a
// k-group is cut into MT + 4 "fillers" placed in DIFFERENT gaps, evenly spread: the MT LDS reads of the next k-group
// (across a panel boundary too: the panel switch -- wait + barrier -- sits in front of the first read), the stash store,
// the first half of the side units, the DMA piece(s), the second half of the side units.
// Pin the accumulators at a program point.  The MFMA builtin is a pure function to LLVM, so nothing ties it to the
// side-effecting skeleton (sched_barrier, stores, DMA) around it: IR passes may -- and for some passes do -- sink all MFMAs of
// a part below that skeleton, leaving the reads of a whole panel live at once (hundreds of spills).  An empty volatile
// asm that "modifies" the accumulators (in AGPRs) orders the MFMA chain against the other side effects at no run-time cost.
template <int MT, int NACC>
__device__ __forceinline__ void pin_acc(f32x16 (&acc)[NACC]) {
    if constexpr (MT == 1) asm volatile("" : "+a"(acc[0]));
    else if constexpr (MT == 2) asm volatile("" : "+a"(acc[0]), "+a"(acc[1]));
    else if constexpr (MT == 4) asm volatile("" : "+a"(acc[0]), "+a"(acc[1]), "+a"(acc[2]), "+a"(acc[3]));
    else static_assert(MT == 1 || MT == 2 || MT == 4, "unsupported tile count");
}

struct NoSide {
    __device__ __forceinline__ void operator()(int) const {}
};

template <int KT, int MT, bool STASH, int NSIDE_, int PPG, int SHIFT, class Side, int NACC, int NIN>
__device__ __forceinline__ void gemm_part(f32x16 (&acc)[NACC], const float (&in)[NIN], const PanelPipe& pipe, int p0,
                                          float* stash, const Side& side) {
#if NNR_ABLATE & 1
    constexpr int NSIDE = 0;   // profiling build only
#else
    constexpr int NSIDE = NSIDE_;
#endif
    static_assert(MT <= NACC && 16 * KT <= NIN, "tile counts exceed the register arrays");
    constexpr int G = 4 * KT, GP = part_gp(MT);
    constexpr int NM = 4 * MT;   // MFMAs (= gaps) per k-group
    constexpr int NF = MT + 4;   // fillers per k-group; filler f sits in the gap after MFMA number f * NM / NF
    Frags<MT> cur = gemm_open<KT, MT>(pipe, p0);
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const int u0 = (g - SHIFT) * PPG;                 // first side unit of this k-group
        const int um = u0 + (PPG + 1) / 2, u1 = u0 + PPG;  // split between two gaps
        Frags<MT> nxt;
#pragma unroll
        for (int j = 0; j < NM; ++j) {
            __builtin_amdgcn_sched_barrier(0);
            acc[j % MT] = mfma32(cur.v[j % MT][j / MT], in[4 * g + j / MT], acc[j % MT]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                if (f * NM / NF != j) continue;
                if (f < MT) {   // LDS read of fragment f of the next k-group
                    if (g + 1 < G) {
                        const int pn = p0 + (g + 1) / GP;
                        // panel switch inside the part: the stores of this panel's earlier k-groups are younger than
                        // the pieces waited for
                        if (f == 0 && (g + 1) % GP == 0) pipe.template enter<STASH ? GP - 1 : 0>(pn);
                        const f32x4* buf = pipe.lds + pipe.buffer(pn) * kPanelF4 + pipe.lane;
                        nxt.v[f] = buf[(((g + 1) % GP) * MT + f) * 64];
                    }
                } else if (f == MT) {
#if !(NNR_ABLATE & 8)
                    if constexpr (STASH)
                        *reinterpret_cast<f32x4*>(stash + 8 * g) = f32x4{in[4 * g], in[4 * g + 1], in[4 * g + 2], in[4 * g + 3]};
#endif
                } else if (f == MT + 1) {
                    if constexpr (NSIDE > 0) {
#pragma unroll
                        for (int u = u0; u < um; ++u)
                            if (u >= 0 && u < NSIDE) side(u);
                    }
                } else if (f == MT + 2) {
                    // DMA pieces of the panel two ahead, spread over the k-groups of the current panel: slot 0 right
                    // after the panel switch (which happens in the LAST k-group of the previous panel, or in
                    // gemm_open), slots 1.. in the following k-groups
                    const int pi = g / GP, gi = g % GP;
                    const int n_in = (G - pi * GP) < GP ? (G - pi * GP) : GP;
                    if (gi == n_in - 1) {
                        if (g + 1 < G) {   // this k-group entered panel pi+1 above
                            const int n_nx = (G - (pi + 1) * GP) < GP ? (G - (pi + 1) * GP) : GP;
                            pipe.pieces(p0 + pi + 3, 0, (8 + n_nx - 1) / n_nx);
                        }
                    } else {
                        const int ppk = (8 + n_in - 1) / n_in;
                        pipe.pieces(p0 + pi + 2, (gi + 1) * ppk, ppk);
                    }
                } else {
                    if constexpr (NSIDE > 0) {
#pragma unroll
                        for (int u = um; u < u1; ++u)
                            if (u >= 0 && u < NSIDE) side(u);
                    }
                }
            }
        }
        pin_acc<MT>(acc);
        __builtin_amdgcn_sched_barrier(0);
        if (g + 1 < G) cur = nxt;
    }
    if constexpr (NSIDE > 0) {
#pragma unroll
        for (int u = (G - SHIFT) * PPG; u < NSIDE; ++u)
            if (u >= 0) side(u);
    }
}

template <int KT, int MT, bool STASH = false, int NACC, int NIN>
__device__ __forceinline__ void gemm_part(f32x16 (&acc)[NACC], const float (&in)[NIN], const PanelPipe& pipe, int p0,
                                          float* stash = nullptr) {
    gemm_part<KT, MT, STASH, 0, 1, 0>(acc, in, pipe, p0, stash, NoSide{});
}

// ---- bf16 training mode: tile-major planes (the kernels themselves: nnr_mlp_bf16.h) ------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int kBlockBf16 = 512;   // bf16 elements per 1 KiB block of a tile-major plane (nnr_layout.h)

// Where a lane starts writing its sample's part of a TILE-MAJOR bf16 plane of `width` features (nnr_layout.h): 1 KiB blocks [chunk
// of 32 samples][group of 16 features][lane][8 bf16]; the lane's 16 bytes of group g sit at the returned pointer + 512 g elements,
// so that a row-step's stash store is one fully coalesced 1 KiB wave-store.  `row` may carry a plane index (row = plane * S_pad +
// sample; S_pad is a multiple of 32).
__device__ __forceinline__ __bf16* tile_row(float* plane, int64_t row, int width, int half) {
    const int64_t c = row & 31;
    return reinterpret_cast<__bf16*>(plane) + (row - c) * width + (32 * half + c) * 8;
}

// The same from the WAVE-UNIFORM first row of the lane's chunk (row0 = plane * S_pad + 32 * chunk, a scalar) and the lane index: a scalar
// base plus a 32-bit lane offset -- the form the compiler addresses with (SGPR pair + VGPR offset) instead of a 64-bit address per lane
__device__ __forceinline__ __bf16* tile_lane(float* plane, int64_t row0, int width, int lane) {
    return reinterpret_cast<__bf16*>(plane) + row0 * width + lane * 8;
}

template <int N>
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
}

// feature index held by register r (any tile) in half h
__device__ __forceinline__ constexpr int frag_feature(int r, int h) { return 32 * (r >> 4) + (r & 3) + 8 * ((r & 15) >> 2) + 4 * h; }

// sin(a) or cos(a) (want_cos) of one lane's argument, accurate to ~1 ulp for |a| < 2^15 -- the arguments reach 2^9 |p| ~ 5e3 rad, so the
// fast hardware sin (1e-6 ABSOLUTE error in revolutions) is not usable at 1e-4 parity, and ocml's sincosf, which is, spends ~80
// instructions per call on a Payne-Hanek path these arguments never need (the 48 calls per sample were 16-32 k cycles per wave: 5 % of
// the fp32 forward, a quarter of the bf16 one).  Two-term Cody-Waite reduction with fma (exact for the product, the remainder is < 1:
// error ~3e-8) to r in [-pi/4, pi/4], quadrant n; cos(a) = sin(a + pi/2) shifts the quadrant; then the cephes minimax polynomials
// (|error| < 1 ulp on the reduced range).  ~20 VALU instructions.
__device__ __forceinline__ float sin_or_cos(float a, bool want_cos) {
    const float n = __builtin_rintf(a * 0.63661977236758134308f);          // a / (pi/2), round to nearest even
    float r = __builtin_fmaf(-n, 1.57079637050628662109375f, a);             // pi/2 rounded to fp32 ...
    r = __builtin_fmaf(-n, -4.37113900018624283e-8f, r);                     // ... and what it leaves out
    const int q = (int)n + (want_cos ? 1 : 0);
    const float z = r * r;
    const float sp = __builtin_fmaf(__builtin_fmaf(__builtin_fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f) * z, r, r);
    const float cp = __builtin_fmaf(__builtin_fmaf(__builtin_fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f), z * z,
                                    __builtin_fmaf(-0.5f, z, 1.0f));
    const float v = (q & 1) ? cp : sp;
    return (q & 2) ? -v : v;
}

// What feature f of gamma_L is, at compile time (same 3-wide block order as the reference: model/official_nerf.py:112-118)
struct EncWhat { int coord, lvl; bool identity, is_cos, pad; };
__device__ __forceinline__ constexpr EncWhat enc_what(int f, int n_real) {
    if (f >= n_real) return {0, 0, false, false, true};
    if (f < 3) return {f, 0, true, false, false};
    const int t = f - 3, lvl = t / 6, rem = t - 6 * lvl;
    return {rem >= 3 ? rem - 3 : rem, lvl, false, rem >= 3, false};
}

// Register r of the lane's half of gamma_L(x, y, z) = [x, sin(2^0 x), cos(2^0 x), ...] in fragment layout: feature
// frag_feature(r, half) -- frag_feature(r, 0) for the lower half-wave, + 4 for the upper.  Both candidates are decoded at compile
// time; at run time the lane picks its argument (one select per field that differs) and evaluates ONE sin_or_cos.
// f >= n_real is padding (0).
__device__ __forceinline__ float enc_register(int r, int half, int n_real, float x, float y, float z) {
    const EncWhat w0 = enc_what(frag_feature(r, 0), n_real), w1 = enc_what(frag_feature(r, 1), n_real);
    const float c0 = w0.coord == 0 ? x : (w0.coord == 1 ? y : z), c1 = w1.coord == 0 ? x : (w1.coord == 1 ? y : z);
    const float v = half ? c1 : c0;
    const float a = v * (half ? (float)(1 << w1.lvl) : (float)(1 << w0.lvl));        // exact: a power of two
    const float sc = sin_or_cos(a, half ? w1.is_cos : w0.is_cos);
    const bool ident = half ? w1.identity : w0.identity, pad = half ? w1.pad : w0.pad;
    return pad ? 0.f : (ident ? v : sc);
}

// One element of gamma_L(x) for a run-time feature index (kept for callers outside the hot loop)
__device__ __forceinline__ float enc_feature(int f, int n_real, float x, float y, float z) {
    if (f >= n_real) return 0.f;
    int t = f < 3 ? f : f - 3;
    int lvl = f < 3 ? 0 : t / 6;
    int rem = t - 6 * lvl;
    int c = f < 3 ? f : (rem >= 3 ? rem - 3 : rem);
    float v = c == 0 ? x : (c == 1 ? y : z);
    if (f < 3) return v;
    return sin_or_cos(ldexpf(v, lvl), rem >= 3);
}

// d gamma_L / d x contracted with the upstream gradient `ge` of feature f, using the *stored* encoding `e`:
// d sin(a x)/dx = a cos(a x), d cos(a x)/dx = -a sin(a x), and the partner (cos for a sin feature, sin for a cos
// feature) lives 3 features away in the same vector.  Returns the contribution and the coordinate it belongs to.
// partner must be supplied by the caller (it may live in the other half-wave).
__device__ __forceinline__ void enc_feature_meta(int f, int n_real, int& coord, float& scale, int& partner) {
    if (f >= n_real) { coord = 0; scale = 0.f; partner = f; return; }
    if (f < 3) { coord = f; scale = 1.f; partner = -1; return; }
    int t = f - 3;
    int lvl = t / 6;
    int rem = t - 6 * lvl;
    bool is_cos = rem >= 3;
    coord = is_cos ? rem - 3 : rem;
    scale = is_cos ? -ldexpf(1.f, lvl) : ldexpf(1.f, lvl);
    partner = is_cos ? f - 3 : f + 3;
}

// ReLU as ONE instruction.  fmaxf(x, 0) (and every builtin hipcc folds to it) costs two: the MFMA result is canonicalised
// first.  Every VALU instruction inside the MFMA stream delays the matrix pipe by about its own issue time, so the
// instruction is spelled out.  (v_max_f32 returns the non-NaN operand, like fmaxf.)
__device__ __forceinline__ float relu1(float x) {
    float r;
    asm("v_max_f32_e32 %0, 0, %1" : "=v"(r) : "v"(x));
    return r;
}

__device__ __forceinline__ float softplus_ref(float x) {  // F.softplus, beta=1, threshold=20
    return x > 20.f ? x : log1pf(expf(x));
}
__device__ __forceinline__ float sigmoid_ref(float x) { return 1.f / (1.f + expf(-x)); }

// ---- alpha compositing pieces shared by nnr_composite.hip and the fused epilogue of the inference forward -----------------
constexpr float kEpsT = 1e-6f;  // model/rendering.py:9
// the rendering switches of nnr_cfg.flags (include/nnr.h: NNR_F_DIST_ALPHA, NNR_F_WHITE_BG, NNR_F_RELU_SIGMA; checked in nnr_api.cpp)
constexpr uint32_t kFlagDistAlpha = 1u, kFlagWhiteBg = 2u, kFlagReluSigma = 4u;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

// inclusive product scan across the 64 lanes
__device__ __forceinline__ float wave_scan_mul(float v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        float t = __shfl_up(v, d, 64);
        if (lane >= d) v *= t;
    }
    return v;
}
// density / alpha of one sample.  Returns alpha; d_alpha_d_raw receives d alpha / d sigma_raw.
__device__ __forceinline__ float sample_alpha(float raw, float delta, bool last, uint32_t flags, float& d_alpha_d_raw) {
    float sigma, dsig;
    if (flags & kFlagReluSigma) {
        sigma = fmaxf(raw, 0.f);
        dsig = raw > 0.f ? 1.f : 0.f;
    } else {
        sigma = softplus_ref(raw);
        dsig = raw > 20.f ? 1.f : sigmoid_ref(raw);
    }
    float alpha;
    if (flags & kFlagDistAlpha) {                    // rendering.py:122-128
        const float e = expf(-1.0f * sigma * delta);
        alpha = last ? 1.f : 1.f - e;                  // alpha[:, -1] = 1 *after* the exp: no gradient through it
        d_alpha_d_raw = last ? 0.f : delta * e * dsig;
    } else {                                           // official_nerf.py:82-83
        const float e = expf(-1.0f * sigma);
        alpha = 1.f - e;
        d_alpha_d_raw = e * dsig;
    }
    return alpha;
}

}  // namespace nnr
