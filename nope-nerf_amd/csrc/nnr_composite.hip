// nnr_composite.hip -- per-ray alpha compositing (forward + backward) and the per-ray reduction of point gradients.
// Restates model/rendering.py:119-132,145-147 (alpha from density, transmittance product with eps = 1e-6, weighted
// sums) and the backward derived in SURVEY.md Appendix A.  One wavefront per ray; the running transmittance is a
// 64-lane product scan done with wave shuffles (no LDS in the forward); HBM-bound and tiny: 20 B in per sample,
// 16 B out per ray.
#include "nnr_device.h"
#include "nnr_kernels.h"
#include "../../include/nnr.h"

namespace nnr {

constexpr int kMaxSamplesBwd = 1024;

// inclusive suffix-sum scan (lane i gets sum of lanes >= i)
__device__ __forceinline__ float wave_rscan_add(float v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        float t = __shfl_down(v, d, 64);
        if (lane + d < 64) v += t;
    }
    return v;
}

__global__ __launch_bounds__(256) void composite_fwd_kernel(CompositeArgs a) {
    const int lane = threadIdx.x & 63;
    const int ray = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ray >= a.R) return;
    const int N = a.N;
    const int64_t base = (int64_t)ray * N;
    float carry = 1.f, cr = 0.f, cg = 0.f, cb = 0.f, cz = 0.f, cw = 0.f;
    for (int j0 = 0; j0 < N; j0 += 64) {
        const int j = j0 + lane;
        const bool ok = j < N;
        float alpha = 0.f, z = 0.f;
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
        if (ok) {
            o = *reinterpret_cast<const f32x4*>(a.ws_out4 + 4 * (base + j));
            z = a.ws_z[base + j];
            const float zn = (j + 1 < N) ? a.ws_z[base + j + 1] : 0.f;
            const float delta = (j + 1 < N) ? zn - z : 1e10f;
            float dummy;
            alpha = sample_alpha(o[3], delta, j == N - 1, a.flags, dummy);
        }
        const float v = ok ? (1.f - alpha) + kEpsT : 1.f;     // rendering.py:130
        const float incl = wave_scan_mul(v, lane);
        float excl = __shfl_up(incl, 1, 64);
        if (lane == 0) excl = 1.f;
        const float T = carry * excl;
        carry *= __shfl(incl, 63, 64);
        const float w = alpha * T;
        cr += w * o[0]; cg += w * o[1]; cb += w * o[2]; cz += w * z; cw += w;
        if (ok) {
            if (a.opt_alpha) a.opt_alpha[base + j] = alpha;
            if (a.opt_z) a.opt_z[base + j] = z;
        }
    }
    cr = wave_sum(cr); cg = wave_sum(cg); cb = wave_sum(cb); cz = wave_sum(cz); cw = wave_sum(cw);
    if (lane == 0) {
        if (a.flags & NNR_F_WHITE_BG) { const float bgc = 1.f - cw; cr += bgc; cg += bgc; cb += bgc; }   // :145-147
        a.rgb[3 * ray + 0] = cr; a.rgb[3 * ray + 1] = cg; a.rgb[3 * ray + 2] = cb;
        a.dist[ray] = cz;
    }
}

// Backward: with a_j = gC.c_j + gZ z_j (- sum(gC) for white background) and B_j = sum_{m>j} w_m a_m,
//   dL/dalpha_j = T_j a_j - B_j / (1 - alpha_j + eps),  dL/dc_j = w_j gC,
// then through alpha(sigma_raw) and the colour sigmoid to the MLP's pre-activations.  B_j is accumulated in a true
// reverse scan (never as total - prefix: the division by 1-alpha+eps ~ 1e-6 would amplify the cancellation error).
// ---- backward, evaluated in DOUBLE (round 6) --------------------------------------------------------------------------------------------
// d L / d alpha_j = T_j a_j - B_j / (1 - alpha_j + eps) is a difference of two terms of like size, B a 192-long suffix sum: in fp32 its error
// against an fp64 evaluation of the step was consistently TWICE the CPU oracle's (1.6e-5 against 8e-6 relative L2 of d sigma_raw over twelve seeds,
// the one stage of the whole step where the HIP path was the worse one: tools/fp64_bisect_seeds.py), and every gradient tensor inherits d sigma_raw --
// the tail of the fp64 yardstick (pose_r, layers0.6.weight at 2.3x the CPU's distance).  The kernel handles 20 bytes per sample
// and takes microseconds: the per-sample chain (softplus', exp, transmittance product, suffix sum, the difference) now runs in fp64 on the fp32
// inputs, rounded once at the store.  The FORWARD keeps the reference's fp32 arithmetic (its outputs are compared with the reference's at 1e-6).
__device__ __forceinline__ double wave_scan_mul_d(double v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const double t = __shfl_up(v, d, 64);
        if (lane >= d) v *= t;
    }
    return v;
}
__device__ __forceinline__ double wave_rscan_add_d(double v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const double t = __shfl_down(v, d, 64);
        if (lane + d < 64) v += t;
    }
    return v;
}
// density / alpha of one sample in double (sample_alpha, nnr_device.h): returns alpha; d_alpha_d_raw receives d alpha / d sigma_raw
__device__ __forceinline__ double sample_alpha_d(double raw, double delta, bool last, uint32_t flags, double& d_alpha_d_raw) {
    double sigma, dsig;
    if (flags & kFlagReluSigma) {
        sigma = raw > 0.0 ? raw : 0.0;
        dsig = raw > 0.0 ? 1.0 : 0.0;
    } else {      // F.softplus, beta = 1, threshold 20 (model/official_nerf.py:77-80)
        sigma = raw > 20.0 ? raw : log1p(exp(raw));
        dsig = raw > 20.0 ? 1.0 : 1.0 / (1.0 + exp(-raw));
    }
    double alpha;
    if (flags & kFlagDistAlpha) {                    // rendering.py:122-128
        const double e = exp(-sigma * delta);
        alpha = last ? 1.0 : 1.0 - e;                  // alpha[:, -1] = 1 after the exp: no gradient through it
        d_alpha_d_raw = last ? 0.0 : delta * e * dsig;
    } else {                                           // official_nerf.py:82-83
        const double e = exp(-sigma);
        alpha = 1.0 - e;
        d_alpha_d_raw = e * dsig;
    }
    return alpha;
}

__global__ __launch_bounds__(256) void composite_bwd_kernel(CompositeArgs a) {
    __shared__ double sT[4][kMaxSamplesBwd];
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const int ray = blockIdx.x * 4 + wv;
    if (ray >= a.R) return;
    const int N = a.N;
    const int64_t base = (int64_t)ray * N;
    const double g0 = a.d_rgb[3 * ray], g1 = a.d_rgb[3 * ray + 1], g2 = a.d_rgb[3 * ray + 2];
    const double gz = a.d_dist[ray];
    const double gw = (a.flags & NNR_F_WHITE_BG) ? -(g0 + g1 + g2) : 0.0;
    const double eps = (double)kEpsT;
    // pass 1: transmittance of every sample
    double carry = 1.0;
    for (int j0 = 0; j0 < N; j0 += 64) {
        const int j = j0 + lane;
        const bool ok = j < N;
        double alpha = 0.0;
        if (ok) {
            const float raw = a.ws_out4[4 * (base + j) + 3];
            const float z = a.ws_z[base + j];
            const float delta = (j + 1 < N) ? a.ws_z[base + j + 1] - z : 1e10f;      // (the fp32 difference the forward used)
            double dummy;
            alpha = sample_alpha_d(raw, delta, j == N - 1, a.flags, dummy);
        }
        const double v = ok ? (1.0 - alpha) + eps : 1.0;
        const double incl = wave_scan_mul_d(v, lane);
        double excl = __shfl_up(incl, 1, 64);
        if (lane == 0) excl = 1.0;
        if (ok) sT[wv][j] = carry * excl;
        carry *= __shfl(incl, 63, 64);
    }
    // pass 2: blocks in reverse, suffix sums of w*a
    double tail = 0.0;  // sum over all samples after the current block
    const int nblk = (N + 63) / 64;
    for (int b = nblk - 1; b >= 0; --b) {
        const int j = b * 64 + lane;
        const bool ok = j < N;
        double wa = 0.0, alpha = 0.0, dadr = 0.0, T = 0.0, aj = 0.0;
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
        if (ok) {
            o = *reinterpret_cast<const f32x4*>(a.ws_out4 + 4 * (base + j));
            const float z = a.ws_z[base + j];
            const float delta = (j + 1 < N) ? a.ws_z[base + j + 1] - z : 1e10f;
            alpha = sample_alpha_d(o[3], delta, j == N - 1, a.flags, dadr);
            T = sT[wv][j];
            aj = g0 * o[0] + g1 * o[1] + g2 * o[2] + gz * z + gw;
            wa = alpha * T * aj;
        }
        const double incl = wave_rscan_add_d(wa, lane);      // sum_{m >= j in block}
        const double B = (incl - wa) + tail;                  // sum_{m > j}
        tail += __shfl(incl, 0, 64);
        if (ok) {
            const double dalpha = T * aj - B / ((1.0 - alpha) + eps);
            const double w = alpha * T;
            f32x4 d;
            d[0] = (float)(w * g0 * o[0] * (1.0 - o[0]));          // through sigmoid: c(1-c)
            d[1] = (float)(w * g1 * o[1] * (1.0 - o[1]));
            d[2] = (float)(w * g2 * o[2] * (1.0 - o[2]));
            d[3] = (float)(dalpha * dadr);
            *reinterpret_cast<f32x4*>(a.ws_dout4 + 4 * (base + j)) = d;
        }
    }
}

// d(pts_o) = sum_j dp_j, d(pts_d) = sum_j z_j dp_j, d(view) = sum_j dv_j   (p_j = o + d z_j; v shared by the ray)
__global__ __launch_bounds__(256) void ray_reduce_kernel(RayReduceArgs a) {
    const int lane = threadIdx.x & 63;
    const int ray = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ray >= a.R) return;
    const int64_t base = (int64_t)ray * a.N;
    // (summed in double and rounded once, like the rest of the way back to the pose: nnr_camera.hip, the note at M4T)
    double s[9] = {0., 0., 0., 0., 0., 0., 0., 0., 0.};      // d_pts_o, d_pts_d, d_view
    for (int j = lane; j < a.N; j += 64) {
        const f32x4 p = *reinterpret_cast<const f32x4*>(a.ws_dpts + 4 * (base + j));
        const f32x4 v = *reinterpret_cast<const f32x4*>(a.ws_dview + 4 * (base + j));
        const double z = a.ws_z[base + j];
#pragma unroll
        for (int c = 0; c < 3; ++c) { s[c] += (double)p[c]; s[3 + c] += z * (double)p[c]; s[6 + c] += (double)v[c]; }
    }
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) s[k] += __shfl_xor(s[k], d, 64);
    if (lane == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            a.d_pts_o[3 * ray + c] = (float)s[c];
            a.d_pts_d[3 * ray + c] = (float)s[3 + c];
            a.d_view[3 * ray + c] = (float)s[6 + c];
        }
    }
}

hipError_t launch_composite_fwd(const CompositeArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(composite_fwd_kernel, dim3((a.R + 3) / 4), dim3(256), 0, st, a);
    return hipGetLastError();
}
hipError_t launch_composite_bwd(const CompositeArgs& a, hipStream_t st) {
    if (a.N > kMaxSamplesBwd) return hipErrorInvalidValue;
    hipLaunchKernelGGL(composite_bwd_kernel, dim3((a.R + 3) / 4), dim3(256), 0, st, a);
    return hipGetLastError();
}
hipError_t launch_ray_reduce(const RayReduceArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(ray_reduce_kernel, dim3((a.R + 3) / 4), dim3(256), 0, st, a);
    return hipGetLastError();
}

}  // namespace nnr
