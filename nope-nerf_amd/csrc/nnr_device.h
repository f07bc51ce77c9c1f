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

// acc[mt] += A_part[32*mt.., :] * in   for one layer part.
//   in    : 16*KT registers in fragment layout (this wave's 32 samples)
//   frag  : packed A fragments of the part, [4*KT][MT][64] float4, already offset by +lane
//   stash : optional (sample, feature) row-major destination of `in` (row of this lane's sample, + 4*half): the four
//           registers consumed by k-group g are features 8g+4h..+3, i.e. one 16-byte store per k-group, issued *inside*
//           the MFMA stream.  Stashing a layer's input here -- instead of its output in the epilogue -- spreads the
//           10 KB/sample of training stash evenly over the kernel; in an epilogue burst every CU of the chip stores at
//           once and each wave then sits in s_waitcnt vmcnt (stores count) until HBM has drained 32 MB.
// Fragments are fetched straight from L2/L1 one k-group ahead (1 KiB coalesced per wave-load, 4 MFMAs each); all
// waves of the chip stream the same 2.4 MB so the working set is L2 resident and the 4 waves of a CU share L1 lines.
template <int KT, int MT, bool STASH = false, int NACC, int NIN>
__device__ __forceinline__ void gemm_part(f32x16 (&acc)[NACC], const float (&in)[NIN], const f32x4* __restrict__ frag,
                                          float* stash = nullptr) {
    static_assert(MT <= NACC && 16 * KT <= NIN, "tile counts exceed the register arrays");
    constexpr int G = 4 * KT;
    f32x4 cur[MT], nxt[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) cur[mt] = frag[mt * 64];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        if (g + 1 < G) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) nxt[mt] = frag[((g + 1) * MT + mt) * 64];
        }
        if constexpr (STASH) *reinterpret_cast<f32x4*>(stash + 8 * g) = f32x4{in[4 * g], in[4 * g + 1], in[4 * g + 2], in[4 * g + 3]};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt] = mfma32(cur[mt][i], in[4 * g + i], acc[mt]);
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) cur[mt] = nxt[mt];
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
