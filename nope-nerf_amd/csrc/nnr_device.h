// nnr_device.h -- device-side building blocks shared by the fused MLP kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include "nnr_layout.h"

namespace nnr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// D(32x32) += A(32x2) * B(2x32), exact fp32 (v_mfma_f32_32x32x2_f32): lane l supplies A[l&31][l>>5] and B[l>>5][l&31].
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

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

struct PanelPipe {
    const f32x4* src;  // stream base in global memory, already offset by this lane and this wave's fragment slice
    f32x4* lds;        // base of the three panel buffers in LDS
    int wave, lane;
    int n_panels;      // panels in the stream

    // This wave copies fragments [8*wave, 8*wave+8) of panel p into buffer p % 3: 8 DMA instructions, always exactly 8 --
    // the counted wait below relies on it.
    __device__ __forceinline__ void issue(int p) const {
#ifdef NNR_ABLATE_NO_DMA
        return;
#endif
        const f32x4* g = src + (int64_t)p * kPanelF4;
        f32x4* l = lds + (p % kNBuf) * kPanelF4 + wave * (8 * 64);
#pragma unroll
        for (int i = 0; i < 8; ++i)
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(g + i * 64), (lds_ptr_t)(l + i * 64), 16, 0, 0);
    }
    // Make panel p readable and recycle the buffer of panel p-1 for panel p+2.
    //   vmcnt(8): everything older than the 8 youngest VMEM ops of this wave is complete.  The 8 DMA ops of panel p+1 were
    //   issued after those of panel p, so panel p has landed (any stores issued since only make the wait more conservative).
    //   The barrier then tells every wave that (a) all four slices of panel p are in LDS and (b) everybody is done reading
    //   panel p-1, whose buffer the DMA of panel p+2 overwrites.
    __device__ __forceinline__ void enter(int p) const {
#ifdef NNR_ABLATE_NO_SYNC
        if (p + 2 < n_panels) issue(p + 2);
        return;
#endif
        // lgkmcnt(0): this wave's ds_reads of panel p-1 have returned before it reports "done reading" at the barrier
        if (p + 1 < n_panels)
            asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
        else
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (p + 2 < n_panels) issue(p + 2);
    }
    __device__ __forceinline__ void start() const {  // prologue: two panels in flight
        issue(0);
        if (n_panels > 1) issue(1);
    }
};

// One k-group of A fragments (MT x 16 bytes per lane) in registers.
template <int MT>
struct Frags { f32x4 v[MT]; };

// Enter the first panel of a layer part and fetch its first k-group.  Called BEFORE the previous layer's epilogue so that
// the barrier, the DMA issue and the LDS latency of these reads hide under the epilogue's VALU work.
template <int MT>
__device__ __forceinline__ Frags<MT> gemm_open(const PanelPipe& pipe, int p0) {
    pipe.enter(p0);
    const f32x4* buf = pipe.lds + (p0 % kNBuf) * kPanelF4 + pipe.lane;
    Frags<MT> f;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) f.v[mt] = buf[mt * 64];
    return f;
}

// acc[mt] += A_part[32*mt.., :] * in   for one layer part whose packed panels start at stream panel p0.
//   in    : 16*KT registers in fragment layout (this wave's 32 samples)
//   cur   : the part's first k-group, from gemm_open
//   stash : optional (sample, feature) row-major destination of `in` (row of this lane's sample, + 4*half): the four
//           registers consumed by k-group g are features 8g+4h..+3, i.e. one 16-byte store per k-group, issued *inside*
//           the MFMA stream.  Stashing a layer's input here -- instead of its output in an epilogue burst -- spreads the
//           10 KB/sample of training stash evenly over the kernel.
// Software pipeline, pinned with sched_barrier(0) (left alone, hipcc sinks every ds_read to just before its first use
// and then waits lgkmcnt(0) with the matrix pipe idle): while the 4*MT MFMAs of k-group g run, the fragments of k-group
// g+1 are already on their way from LDS -- across a panel boundary too (the panel switch, i.e. wait + barrier + DMA
// issue, sits in front of those reads and is covered by the same MFMAs).
template <int KT, int MT, bool STASH = false, int NACC, int NIN>
__device__ __forceinline__ void gemm_part(f32x16 (&acc)[NACC], const float (&in)[NIN], const PanelPipe& pipe, int p0,
                                          Frags<MT> cur, float* stash = nullptr) {
    static_assert(MT <= NACC && 16 * KT <= NIN, "tile counts exceed the register arrays");
    constexpr int G = 4 * KT, GP = part_gp(MT);
#pragma unroll
    for (int g = 0; g < G; ++g) {
        // first half of this k-group's MFMAs ...
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt] = mfma32(cur.v[mt][i], in[4 * g + i], acc[mt]);
        }
        __builtin_amdgcn_sched_barrier(0);
        // ... then, with 2*MT MFMAs in the pipe behind them and 2*MT more to come, the LDS reads of the next k-group
        // (hipcc waits lgkmcnt(0) before the next group's first MFMA: placed here the reads have >= 2*MT*64 cycles to land)
        Frags<MT> nxt;
        if (g + 1 < G) {
            const int pn = p0 + (g + 1) / GP;
            if ((g + 1) % GP == 0) pipe.enter(pn);
            const f32x4* buf = pipe.lds + (pn % kNBuf) * kPanelF4 + pipe.lane;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) nxt.v[mt] = buf[(((g + 1) % GP) * MT + mt) * 64];
        }
        if constexpr (STASH)
            *reinterpret_cast<f32x4*>(stash + 8 * g) = f32x4{in[4 * g], in[4 * g + 1], in[4 * g + 2], in[4 * g + 3]};
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 2; i < 4; ++i) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt] = mfma32(cur.v[mt][i], in[4 * g + i], acc[mt]);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (g + 1 < G) cur = nxt;
    }
}

template <int N>
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
}

// feature index held by register r (any tile) in half h
__device__ __forceinline__ int frag_feature(int r, int h) { return 32 * (r >> 4) + (r & 3) + 8 * ((r & 15) >> 2) + 4 * h; }

// One element of gamma_L(x) = [x, sin(2^0 x), cos(2^0 x), ...] in the reference's 3-wide block order
// (model/official_nerf.py:112-118).  f >= 3*(2L+1) is padding (0).  Accurate sincosf (full range reduction): arguments
// reach 2^9 * |p| ~ 5e3 rad, so the fast hardware sin is not usable at 1e-4 parity.
__device__ __forceinline__ float enc_feature(int f, int n_real, float x, float y, float z) {
    if (f >= n_real) return 0.f;
    int t = f < 3 ? f : f - 3;
    int lvl = f < 3 ? 0 : t / 6;
    int rem = t - 6 * lvl;
    int c = f < 3 ? f : (rem >= 3 ? rem - 3 : rem);
    float v = c == 0 ? x : (c == 1 ? y : z);
    if (f < 3) return v;
    float s, co;
    sincosf(ldexpf(v, lvl), &s, &co);
    return rem >= 3 ? co : s;
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

__device__ __forceinline__ float softplus_ref(float x) {  // F.softplus, beta=1, threshold=20
    return x > 20.f ? x : log1pf(expf(x));
}
__device__ __forceinline__ float sigmoid_ref(float x) { return 1.f / (1.f + expf(-x)); }

}  // namespace nnr
