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

template <int MI, int NI, int BIAS>   // BIAS: WgradJob::bias
__device__ __forceinline__ void wgrad_job(const WgradJob& jb, const WgradArgs& a, int lane, int ji) {
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
#ifdef NNR_ABLATE_NO_FLUSH
    return;
#endif
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

#ifdef NNR_TIMELINE
extern "C" int nnr_timeline_wgrad(unsigned long long* host32) {
    return (int)hipMemcpyFromSymbol(host32, HIP_SYMBOL(tl_wgrad), 32 * sizeof(unsigned long long));
}
#endif

__global__ __launch_bounds__(256, 1) void wgrad_kernel(WgradArgs a) {
    const int lane = threadIdx.x & 63;
    const int wslot = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
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

// dW[tile] = sum over the tile's splits of their partial slots (every weight belongs to exactly one tile: a plain store, the caller's
// buffer needs no zero-fill); d(bias) += its share onto the zeros the main kernel wrote (at most two shares per row: a + b == b + a,
// the result does not depend on their order).  Blocks of jobs that are not split 0 of their tile exit at once; split 0 walks the chain.
constexpr int kMaxChain = 256;   // splits of one tile (the plan builder cuts a tile's sample range into far fewer)
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(WgradArgs a) {
    const int ji = blockIdx.x >> 4;   // 16 blocks x 256 threads x one float4 = a 128 x 128 tile
    const WgradJob jb = a.jobs[ji];
    if (jb.split != 0) return;
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
        for (; s + 4 <= n; s += 4) {      // four slots in flight, added in chain (= sample) order
            f32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const f32x4*>(src + (int64_t)chain[s + u] * kSlotFloats);
#pragma unroll
            for (int u = 0; u < 4; ++u) sum += v[u];
        }
        for (; s < n; ++s) sum += *reinterpret_cast<const f32x4*>(src + (int64_t)chain[s] * kSlotFloats);
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
    hipLaunchKernelGGL(wgrad_kernel, dim3(a.n_waves / 4), dim3(256), 0, st, a);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(a.n_jobs * 16), dim3(256), 0, st, a);
    e = hipGetLastError();
    if (e == hipSuccess) e = launch_wgrad_unmerge(a, st);
    prof_after(PROF_WGRAD, st);      // the bracket covers the whole stage: main kernel + slot reduction + un-merge
    return e;
}

hipError_t launch_wgrad_unmerge(const WgradArgs& a, hipStream_t st) {
    const int threads = a.D * a.D + (a.D / 2) * a.D + a.D + a.D / 2;
    const dim3 grid((threads + 255) / 256), block(256);
    if (a.D == 256) {
        if (a.bf16 == 2) hipLaunchKernelGGL((wgrad_unmerge_kernel<256, 2>), grid, block, 0, st, a);
        else if (a.bf16 == 1) hipLaunchKernelGGL((wgrad_unmerge_kernel<256, 1>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((wgrad_unmerge_kernel<256, 0>), grid, block, 0, st, a);
    } else {
        if (a.bf16 == 2) hipLaunchKernelGGL((wgrad_unmerge_kernel<128, 2>), grid, block, 0, st, a);
        else if (a.bf16 == 1) hipLaunchKernelGGL((wgrad_unmerge_kernel<128, 1>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((wgrad_unmerge_kernel<128, 0>), grid, block, 0, st, a);
    }
    return hipGetLastError();
}

}  // namespace nnr
