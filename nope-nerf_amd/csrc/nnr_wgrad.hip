// nnr_wgrad.hip -- weight / bias gradients of the 12 nn.Linear layers from the stashed layer inputs X and the stashed
// pre-activation gradients Dlt:   dW_l[out][in] = sum_s Dlt_l[s][out] * X_l[s][in],   db_l[out] = sum_s Dlt_l[s][out].
// Replaces autograd's `mm` wgrad calls (24 of the 36 backward GEMMs) for model/official_nerf.py:20-37.
//
// Shape of the problem: outputs are tiny (2.4 MB), the reduction dimension is every sample of the step.  So this is a
// split-K kernel: a *wave job* owns a (32*MI) x (32*NI) tile of one dW and a contiguous range of samples, keeps the
// tile in MI*NI MFMA accumulators (256 registers for 4x4) for the whole range and flushes once with float atomics.
// Both operands are read straight from the (sample, feature) row-major stashes: with interleaved sub-tiles
// (row = MI*m + i) one MI-wide and one NI-wide vector load per lane feed MI*NI MFMAs; the four jobs of a 256x256 layer
// that share a sample range sit in one workgroup so their re-reads hit L1/L2.  The job table comes from the host plan.
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

#include "nnr_device.h"
#include "nnr_kernels.h"

namespace nnr {

template <int W>
struct Vec { float v[W]; };

template <int W>
__device__ __forceinline__ Vec<W> load_vec(const float* p, bool ok) {
    Vec<W> r;
    if constexpr (W == 4) {
        f32x4 t = ok ? *reinterpret_cast<const f32x4*>(p) : f32x4{0.f, 0.f, 0.f, 0.f};
        r.v[0] = t[0]; r.v[1] = t[1]; r.v[2] = t[2]; r.v[3] = t[3];
    } else if constexpr (W == 2) {
        f32x2 t = ok ? *reinterpret_cast<const f32x2*>(p) : f32x2{0.f, 0.f};
        r.v[0] = t[0]; r.v[1] = t[1];
    } else {
        r.v[0] = ok ? *p : 0.f;
    }
    return r;
}

constexpr int kU = 4;  // k-steps (pairs of samples) per pipeline stage

template <int MI, int NI>
__device__ __forceinline__ void wgrad_job(const WgradJob& jb, const WgradArgs& a, int lane) {
    const int half = lane >> 5, m = lane & 31;
    const int dp = a.plane_pitch[jb.d_plane], xp = a.plane_pitch[jb.x_plane];
    const bool dok = MI * m < jb.d_valid, xok = NI * m < jb.x_valid;
    const float* dptr = a.ws + a.plane_off[jb.d_plane] + jb.d_col0 + MI * m + (int64_t)half * dp;
    const float* xptr = a.ws + a.plane_off[jb.x_plane] + jb.x_col0 + NI * m + (int64_t)half * xp;

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

    Vec<MI> dc[kU], dn[kU];
    Vec<NI> xc[kU], xn[kU];
    int64_t k = jb.k0;
#pragma unroll
    for (int u = 0; u < kU; ++u) {
        dc[u] = load_vec<MI>(dptr + (k + 2 * u) * dp, dok);
        xc[u] = load_vec<NI>(xptr + (k + 2 * u) * xp, xok);
    }
    for (; k < jb.k1; k += 2 * kU) {
        const int64_t kn = k + 2 * kU;
        if (kn < jb.k1) {
#pragma unroll
            for (int u = 0; u < kU; ++u) {
                dn[u] = load_vec<MI>(dptr + (kn + 2 * u) * dp, dok);
                xn[u] = load_vec<NI>(xptr + (kn + 2 * u) * xp, xok);
            }
        }
#pragma unroll
        for (int u = 0; u < kU; ++u) {
#pragma unroll
            for (int i = 0; i < MI; ++i) {
#pragma unroll
                for (int j = 0; j < NI; ++j) acc[i][j] = mfma32(dc[u].v[i], xc[u].v[j], acc[i][j]);
                bsum[i] += dc[u].v[i];
            }
        }
#pragma unroll
        for (int u = 0; u < kU; ++u) { dc[u] = dn[u]; xc[u] = xn[u]; }
    }

    // flush: D[row m'][col n] of sub-tile (i,j) -> dW[row0 + MI*m' + i][wcol0 + NI*n + j]
    float* gw = a.gw[jb.layer];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int colw = jb.wcol0 + NI * m + j;
            const bool cok = (NI * m + j < jb.x_valid) && colw < jb.cols_real;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int mr = (r & 3) + 8 * (r >> 2) + 4 * half;
                const int row = jb.row0 + MI * mr + i;
                if (cok && row < jb.rows_real) unsafeAtomicAdd(gw + (int64_t)row * jb.ldw + colw, acc[i][j][r]);
            }
        }
    if (jb.bias) {
        float* gb = a.gb[jb.layer];
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int row = jb.row0 + MI * m + i;
            if (dok && row < jb.rows_real) unsafeAtomicAdd(gb + row, bsum[i]);
        }
    }
}

__global__ __launch_bounds__(256, 1) void wgrad_kernel(WgradArgs a) {
    const int lane = threadIdx.x & 63;
    const int ji = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (ji >= a.n_jobs) return;
    const WgradJob jb = a.jobs[ji];
    if (jb.layer < 0) return;
    const int key = __builtin_amdgcn_readfirstlane(jb.MI * 8 + jb.NI);
    switch (key) {
        case 4 * 8 + 4: wgrad_job<4, 4>(jb, a, lane); break;
        case 4 * 8 + 2: wgrad_job<4, 2>(jb, a, lane); break;
        case 4 * 8 + 1: wgrad_job<4, 1>(jb, a, lane); break;
        case 2 * 8 + 4: wgrad_job<2, 4>(jb, a, lane); break;
        case 2 * 8 + 1: wgrad_job<2, 1>(jb, a, lane); break;
        case 1 * 8 + 4: wgrad_job<1, 4>(jb, a, lane); break;
        case 1 * 8 + 2: wgrad_job<1, 2>(jb, a, lane); break;
        default: break;
    }
}

hipError_t launch_wgrad(const WgradArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(wgrad_kernel, dim3((a.n_jobs + 3) / 4), dim3(256), 0, st, a);
    return hipGetLastError();
}

}  // namespace nnr
