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

// Unconditional vector load, no masking of the value.  A lane outside the valid column range reads a valid address (its
// pointer is clamped to column 0 by the caller); what it contributes lands only in its own row / column of the MFMA
// result, which the flush never writes.  Neither a branch around the load nor a select on the loaded value is allowed
// here: both make hipcc wait for the just-issued prefetch (vmcnt(0) at the join / before the v_cndmask) and serialise
// the pipeline (cdna_hip_programming.md, "three .s-level traps" (c)).
template <int W>
__device__ __forceinline__ Vec<W> load_vec(const float* p) {
    Vec<W> r;
    if constexpr (W == 4) {
#ifdef NNR_NT_WGRAD
        const f32x4 t = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
#else
        const f32x4 t = *reinterpret_cast<const f32x4*>(p);
#endif
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

template <int MI, int NI>
__device__ __forceinline__ void wgrad_job(const WgradJob& jb, const WgradArgs& a, int lane) {
    const int half = lane >> 5, m = lane & 31;
    const int dp = a.plane_pitch[jb.d_plane], xp = a.plane_pitch[jb.x_plane];
    const bool dok = MI * m < jb.d_valid, xok = NI * m < jb.x_valid;
    const float* dptr = a.ws + a.plane_off[jb.d_plane] + jb.d_col0 + (dok ? MI * m : 0) + (int64_t)half * dp;
    const float* xptr = a.ws + a.plane_off[jb.x_plane] + jb.x_col0 + (xok ? NI * m : 0) + (int64_t)half * xp;

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
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            d[u] = load_vec<MI>(dptr + (kk + 2 * u) * dp);
            x[u] = load_vec<NI>(xptr + (kk + 2 * u) * xp);
        }
    };
    auto compute = [&](const Vec<MI>(&d)[kU], const Vec<NI>(&x)[kU]) {
#pragma unroll
        for (int u = 0; u < kU; ++u) {
#pragma unroll
            for (int i = 0; i < MI; ++i) {
#pragma unroll
                for (int j = 0; j < NI; ++j) acc[i][j] = mfma32(d[u].v[i], x[u].v[j], acc[i][j]);
                bsum[i] += d[u].v[i];
            }
        }
    };
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

    // flush: D[row m'][col n] of sub-tile (i,j) -> dW[row0 + MI*m' + i][wcol0 + NI*n + j]
#ifdef NNR_ABLATE_NO_FLUSH
    return;
#endif
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
                if (cok && MI * mr + i < jb.d_valid && row < jb.rows_real) unsafeAtomicAdd(gw + (int64_t)row * jb.ldw + colw, acc[i][j][r]);
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
