// nnr_camera.hip -- the O(1)/O(R) front and back ends of a training step, each as ONE launch instead of the dozens of
// tiny ATen kernels (and four rocSOLVER 4x4 LU inverses) the reference's formulation costs per step:
//   se3_exp     LearnPose.forward -> make_c2w -> Exp           reference model/poses.py:23-31, model/common.py:277-310
//   inv4        torch.inverse on (B,4,4)                        reference model/training.py:238, model/common.py:139-141
//   ray_setup   origin_to_world / transform_to_world / image_points_to_world + norms, masks, view dir
//                                                               reference model/rendering.py:54-87,194-195, common.py:112-237
//   depth_gather  F.interpolate(nearest) + gather at ray_idx    reference model/network.py:22-24
//   render_loss   rgb L1|L2 sum / R  +  depth L1 sum / M         reference model/losses.py:27-32,59-64,196-202
// All of them are latency-bound bookkeeping (<= a few KB); the point is launch count, not bandwidth.
#include "nnr_device.h"
#include <algorithm>
#include "nnr_kernels.h"

// No floating-point contraction in this file.  The fused front end (step_rays_*) restates the arithmetic of the separate kernels below "in
// the same order", and the two must produce the SAME rays: a multiply-add that the compiler contracts in one kernel and not in the other
// moves a ray by an ulp, an ulp flips a ReLU gate somewhere in 6 000 samples, and the smallest network gradients then differ by 0.4 %
// between the two front ends (tests/test_gpu_camera.py caught exactly that when round 5 added code to the fused kernel).  These kernels
// are latency-bound bookkeeping; the reference's own ATen ops round every product and sum separately.
#pragma clang fp contract(off)

namespace nnr {

// ------------------------------------------------------------------------------------------------ 4x4 helpers
// The forward evaluates in float, as the reference's ATen ops do (the rays must be the reference's rays).  Every BACKWARD of this file
// evaluates in double from those float values and rounds its results once: the pose gradient is a sum over all rays of terms that cancel
// to ~1e-3 of their magnitude, so the way back adds no rounding of its own.  (It is NOT what holds the pose gradients at 2.3x the CPU oracle's
// distance to fp64 in the 12-seed yardstick -- 2.30 before, 2.35 after, profiles/r06/k_ / l_yardstick_*.txt; that is the forward inverse
// below.)  These kernels are a few microseconds of latency-bound work; the double rate is not what they wait on.
template <typename T> struct M4T { T m[16]; };
using M4 = M4T<float>;
using M4d = M4T<double>;

template <typename A, typename B>
__device__ __forceinline__ auto mul4(const M4T<A>& a, const M4T<B>& b) -> M4T<decltype(A() * B())> {
    using T = decltype(A() * B());
    M4T<T> c;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            T s = T(0);
#pragma unroll
            for (int k = 0; k < 4; ++k) s += (T)a.m[4 * i + k] * (T)b.m[4 * k + j];
            c.m[4 * i + j] = s;
        }
    return c;
}
template <typename T>
__device__ __forceinline__ M4T<T> transpose4(const M4T<T>& a) {
    M4T<T> c;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) c.m[4 * i + j] = a.m[4 * j + i];
    return c;
}
// general 4x4 inverse by cofactors (adjugate / determinant), in float like the reference's torch.inverse.
// Round 6 measured the alternative -- the same cofactors evaluated in double and rounded once (profiles/r06/n_yardstick_double_inv4.txt): against an
// fp64 evaluation of the step the worst per-tensor median of HIP / CPU-fp32 drops from 2.35 to 1.37 (a cofactor is six triple products that
// cancel, and the rays carry every ulp of these matrices through the 2^9 encoding frequency into every gradient: this IS the tail of that
// yardstick, nothing downstream is).  But the reference inverts in float too (LAPACK's LU on the CPU, MAGMA's on the GPU), and the exactly
// rounded inverse is FURTHER from what it computes than these cofactors are: golden noraydir_relu_d128 pose_r 1.02e-4 against the 1e-4 bar, first-20-step
// deviation of the training run 5.2e-4 against 2.5e-4 (profiles/r06/p_*).  Parity with the reference is the gate; float stays.
__device__ __forceinline__ M4 inv4(const M4& a) {
    const float* m = a.m;
    float inv[16];
    inv[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
    inv[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
    inv[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
    inv[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
    inv[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
    inv[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
    inv[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
    inv[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
    inv[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
    inv[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
    inv[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
    inv[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
    inv[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
    inv[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
    inv[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
    inv[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
    const float det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
    const float rdet = 1.0f / det;
    M4 r;
#pragma unroll
    for (int i = 0; i < 16; ++i) r.m[i] = inv[i] * rdet;
    return r;
}
__device__ __forceinline__ M4 load4(const float* p) {
    M4 a;
#pragma unroll
    for (int i = 0; i < 16; ++i) a.m[i] = p[i];
    return a;
}
// dL/dA for Y = A^-1:  -Y^T (dL/dY) Y^T   (Y as the forward left it, the products in the gradient's type)
template <typename A, typename B>
__device__ __forceinline__ auto inv_backward(const M4T<A>& y, const M4T<B>& dy) -> M4T<decltype(A() * B())> {
    const M4T<A> yt = transpose4(y);
    auto r = mul4(mul4(yt, dy), yt);
#pragma unroll
    for (int i = 0; i < 16; ++i) r.m[i] = -r.m[i];
    return r;
}
__device__ __forceinline__ M4d load4d(const float* p) {
    M4d a;
#pragma unroll
    for (int i = 0; i < 16; ++i) a.m[i] = (double)p[i];
    return a;
}

// ------------------------------------------------------------------------------------------------ SE(3) exp
// R = I + sin(th)/th K + (1-cos(th))/th^2 K^2, th = |r| + 1e-15 (model/common.py:290-299), c2w = [[R,t],[0,0,0,1]].
__device__ __forceinline__ void se3_exp_matrix(const float* r, const float* t, float* c2w) {
    const float x = r[0], y = r[1], z = r[2];
    const float n = sqrtf(x * x + y * y + z * z);
    const float th = n + 1e-15f;
    const float a = sinf(th) / th, b = (1.f - cosf(th)) / (th * th);
    const float K[9] = {0.f, -z, y, z, 0.f, -x, -y, x, 0.f};
    float K2[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) K2[3 * i + j] = K[3 * i] * K[j] + K[3 * i + 1] * K[3 + j] + K[3 * i + 2] * K[6 + j];
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) c2w[4 * i + j] = (i == j ? 1.f : 0.f) + a * K[3 * i + j] + b * K2[3 * i + j];
        c2w[4 * i + 3] = t[i];
    }
    c2w[12] = 0.f; c2w[13] = 0.f; c2w[14] = 0.f; c2w[15] = 1.f;
}
__global__ void se3_exp_fwd_kernel(const float* r_all, const float* t_all, int idx, float* c2w) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float m[16];
    se3_exp_matrix(r_all + 3 * idx, t_all + 3 * idx, m);
    for (int k = 0; k < 16; ++k) c2w[k] = m[k];
}

// d loss / d (r, t) of one camera from d loss / d c2w (evaluated in double: see the note at M4T)
__device__ __forceinline__ void se3_exp_grad(const float* r, const double* d_c2w, float* d_r, float* d_t) {
    const double x = r[0], y = r[1], z = r[2];
    const double n = sqrt(x * x + y * y + z * z);
    const double th = n + 1e-15;
    const double s = sin(th), c = cos(th);
    const double a = s / th, b = (1. - c) / (th * th);
    const double da = c / th - s / (th * th);                                  // d(sin th / th)/d th, as autograd forms it
    const double db = s / (th * th) - 2. * (1. - c) / (th * th * th);          // d((1-cos th)/th^2)/d th
    const double K[9] = {0., -z, y, z, 0., -x, -y, x, 0.};
    double K2[9], G[9];
    for (int p = 0; p < 3; ++p)
        for (int q = 0; q < 3; ++q) {
            K2[3 * p + q] = K[3 * p] * K[q] + K[3 * p + 1] * K[3 + q] + K[3 * p + 2] * K[6 + q];
            G[3 * p + q] = d_c2w[4 * p + q];
        }
    double ga = 0., gb = 0.;
    for (int p = 0; p < 9; ++p) { ga += G[p] * K[p]; gb += G[p] * K2[p]; }
    // dL/dK = a G + b (G K^T + K^T G)
    double gK[9];
    for (int p = 0; p < 3; ++p)
        for (int q = 0; q < 3; ++q) {
            double gkT = 0., kTg = 0.;
            for (int k = 0; k < 3; ++k) { gkT += G[3 * p + k] * K[3 * q + k]; kTg += K[3 * k + p] * G[3 * k + q]; }
            gK[3 * p + q] = a * G[3 * p + q] + b * (gkT + kTg);
        }
    const double gth = ga * da + gb * db;
    const double inv_n = n > 0. ? 1. / n : 0.;                                // d|r|/dr = r/|r|, subgradient 0 at r = 0
    d_r[0] = (float)(gK[7] - gK[5] + gth * x * inv_n);
    d_r[1] = (float)(gK[2] - gK[6] + gth * y * inv_n);
    d_r[2] = (float)(gK[3] - gK[1] + gth * z * inv_n);
    d_t[0] = (float)d_c2w[3];
    d_t[1] = (float)d_c2w[7];
    d_t[2] = (float)d_c2w[11];
}
// gradients into full (n_cams,3) tables: zero everywhere except row idx
__global__ void se3_exp_bwd_kernel(const float* r_all, int idx, int n_cams, const float* d_c2w, float* d_r_all, float* d_t_all) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 3 * n_cams && i / 3 != idx) { d_r_all[i] = 0.f; d_t_all[i] = 0.f; }
    if (i != 0) return;
    double g[16];
    for (int k = 0; k < 16; ++k) g[k] = (double)d_c2w[k];
    se3_exp_grad(r_all + 3 * idx, g, d_r_all + 3 * idx, d_t_all + 3 * idx);
}

// ------------------------------------------------------------------------------------------------ batched 4x4 inverse
__global__ void inv4_fwd_kernel(const float* a, float* y, int batch) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= batch) return;
    const M4 r = inv4(load4(a + 16 * i));
    for (int k = 0; k < 16; ++k) y[16 * i + k] = r.m[k];
}
__global__ void inv4_bwd_kernel(const float* y, const float* dy, float* da, int batch) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= batch) return;
    const M4d r = inv_backward(load4(y + 16 * i), load4d(dy + 16 * i));
    for (int k = 0; k < 16; ++k) da[16 * i + k] = (float)r.m[k];
}

// ------------------------------------------------------------------------------------------------ ray setup
__device__ __forceinline__ void pixel_to_world(const RaySetupArgs& a, M4& kinv, M4& winv, M4& sinv, M4& m) {
    kinv = inv4(load4(a.K));
    winv = inv4(load4(a.W));
    sinv = inv4(load4(a.S));
    m = mul4(mul4(sinv, winv), kinv);   // model/common.py:153: scale^-1 @ world^-1 @ camera^-1
}

__global__ __launch_bounds__(256) void ray_setup_fwd_kernel(RaySetupArgs a) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.R) return;
    M4 kinv, winv, sinv, m;
    pixel_to_world(a, kinv, winv, sinv, m);
    const float px = a.pixels[2 * i], py = a.pixels[2 * i + 1];
    const float dep = a.depth ? a.depth[i] : 1.f;
    float ray[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) ray[c] = m.m[4 * c] * px + m.m[4 * c + 1] * py + m.m[4 * c + 2];   // pixels_world - camera_world
    const float n = sqrtf(ray[0] * ray[0] + ray[1] * ray[1] + ray[2] * ray[2]);
    const float q0 = ray[0] * dep, q1 = ray[1] * dep, q2 = ray[2] * dep;
    float dgt = sqrtf(q0 * q0 + q1 * q1 + q2 * q2);                                              // |points_world - camera_world|
    if (!a.normalise) dgt = dgt / n;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float d = a.normalise ? ray[c] / n : ray[c];
        a.pts_o[3 * i + c] = m.m[4 * c + 3];
        a.dir[3 * i + c] = d;
        a.view[3 * i + c] = a.use_dir ? -d : 1.f;
    }
    a.ray_norm[i] = n;
    a.d_gt[i] = dgt;
    a.mask[i] = (isfinite(dgt) && dgt != 0.f) ? 1 : 0;
}

// The per-ray part of both backward kernels (ray_setup_bwd, step_rays_bwd), in double: from the upstream gradients of one ray to
// d loss / d (pixels_world - camera_world) and d loss / d depth.
struct RayBack { double gray[3]; double gdep; };
__device__ __forceinline__ RayBack ray_backward(const M4& m, float px, float py, float depf, const float* g_dir, const float* g_view, const float* g_norm,
                                                const float* g_dgt, int i, bool use_dir, bool normalise) {
    RayBack o;
    const double dep = depf;
    double ray[3], gdir[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) ray[c] = (double)m.m[4 * c] * px + (double)m.m[4 * c + 1] * py + (double)m.m[4 * c + 2];
    const double n = sqrt(ray[0] * ray[0] + ray[1] * ray[1] + ray[2] * ray[2]);
    const double rn = 1. / n;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        double g = g_dir ? (double)g_dir[3 * i + c] : 0.;
        if (use_dir && g_view) g -= (double)g_view[3 * i + c];     // view = -dir
        gdir[c] = g;
    }
    double gn = g_norm ? (double)g_norm[i] : 0.;                   // dL/d|ray|
    o.gdep = 0.;
    const double gd = g_dgt ? (double)g_dgt[i] : 0.;
    const double sgn = dep > 0. ? 1. : (dep < 0. ? -1. : 0.);
    if (normalise) {
        const double dot = (gdir[0] * ray[0] + gdir[1] * ray[1] + gdir[2] * ray[2]) * rn * rn;
#pragma unroll
        for (int c = 0; c < 3; ++c) o.gray[c] = (gdir[c] - ray[c] * dot) * rn;   // through ray / |ray|
        if (gd != 0.) { gn += gd * fabs(dep); o.gdep = gd * n * sgn; }          // d_gt = |depth| |ray|
    } else {
#pragma unroll
        for (int c = 0; c < 3; ++c) o.gray[c] = gdir[c];
        if (gd != 0.) o.gdep = gd * sgn;                                         // d_gt = |depth| (the |ray| cancels)
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) o.gray[c] += gn * ray[c] * rn;
    return o;
}

// One workgroup of 512 threads strides over the rays and reduces in a FIXED order (lane tree, then the 8 waves in index order; in double):
// the 12 sums -- and with them every pose gradient of the step -- are bit-reproducible from run to run, which float atomics
// across workgroups are not.  R is a few thousand rays per rank; the work is O(R) and latency-bound either way, and the
// matrix chain rule (formerly a second launch behind a memset) runs in the same kernel.
__global__ __launch_bounds__(512) void ray_setup_bwd_kernel(RaySetupArgs a) {
    __shared__ double red[12][8];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    double acc[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) acc[k] = 0.;
    M4 m;
    {
        M4 kinv, winv, sinv;
        pixel_to_world(a, kinv, winv, sinv, m);
    }
    for (int i = threadIdx.x; i < a.R; i += 512) {
        const float px = a.pixels[2 * i], py = a.pixels[2 * i + 1];
        const float dep = a.depth ? a.depth[i] : 1.f;
        const RayBack rb = ray_backward(m, px, py, dep, a.g_dir, a.g_view, a.g_norm, a.g_dgt, i, a.use_dir, a.normalise);
        if (a.g_depth) a.g_depth[i] = (float)rb.gdep;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            acc[4 * c + 0] += rb.gray[c] * px;
            acc[4 * c + 1] += rb.gray[c] * py;
            acc[4 * c + 2] += rb.gray[c];
            acc[4 * c + 3] += a.g_o ? (double)a.g_o[3 * i + c] : 0.;
        }
    }
#pragma unroll
    for (int k = 0; k < 12; ++k) {
        double v = acc[k];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
        if (lane == 0) red[k][wv] = v;
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    // dL/dM (12 sums) -> dL/dK, dL/dW, dL/dS through M = S^-1 W^-1 K^-1
    M4d dm;
    for (int k = 0; k < 12; ++k) {
        double t = 0.;
        for (int w = 0; w < 8; ++w) t += red[k][w];
        dm.m[k] = t;
        a.acc[k] = (float)t;
    }
    for (int k = 12; k < 16; ++k) dm.m[k] = 0.;
    M4 kinv, winv, sinv, m2;
    pixel_to_world(a, kinv, winv, sinv, m2);
    const M4d sw = mul4(load4d(sinv.m), winv);
    const M4d d_kinv = mul4(transpose4(sw), dm);             // M = (S^-1 W^-1) K^-1
    const M4d d_sw = mul4(dm, transpose4(kinv));
    const M4d d_sinv = mul4(d_sw, transpose4(winv));
    const M4d d_winv = mul4(transpose4(sinv), d_sw);
    const M4d gk = inv_backward(kinv, d_kinv), gw = inv_backward(winv, d_winv), gs = inv_backward(sinv, d_sinv);
    for (int k = 0; k < 16; ++k) { a.gK[k] = (float)gk.m[k]; a.gW[k] = (float)gw.m[k]; a.gS[k] = (float)gs.m[k]; }
}

// ------------------------------------------------------------------------------------------------ depth gather
// F.interpolate(mode='nearest') source index: min(floor(dst * (float)in/out), in-1), float32 like ATen.
__device__ __forceinline__ int nearest_src(int dst, int dst_size, int src_size) {
    const float scale = (float)src_size / (float)dst_size;
    const int s = (int)floorf((float)dst * scale);
    return s < src_size - 1 ? s : src_size - 1;
}
__global__ void depth_gather_fwd_kernel(const float* img, const int64_t* ray_idx, float* out, int R, int h, int w, int hd, int wd) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R) return;
    const int64_t q = ray_idx[i];
    const int y = (int)(q / w), x = (int)(q - (int64_t)y * w);
    out[i] = img[(int64_t)nearest_src(y, h, hd) * wd + nearest_src(x, w, wd)];
}
// Several rays share a depth pixel whenever the mono-depth map is coarser than the image (the DPT default), and float atomics would
// make that pixel's sum depend on arrival order.  Instead ONE ray owns each hit pixel -- the lowest-numbered ray that maps to it -- and
// adds the gradients of all rays of that pixel in a fixed order (four interleaved quarter-walks, see the kernel): every ray compares its target with every other ray's (LDS tiles of 256;
// R^2 integer compares, 1 M at 1024 rays), no atomics, bit-reproducible.  g_img must hold zeros (or whatever is to be accumulated into).
__global__ __launch_bounds__(256) void depth_gather_bwd_kernel(const float* g, const int64_t* ray_idx, float* g_img, int R, int h, int w,
                                                               int hd, int wd) {
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    __shared__ __attribute__((aligned(16))) int tgt[256];
    __shared__ __attribute__((aligned(16))) float gv[256];
    auto target = [&](int k) {
        const int64_t q = ray_idx[k];
        const int y = (int)(q / w), x = (int)(q - (int64_t)y * w);
        return nearest_src(y, h, hd) * wd + nearest_src(x, w, wd);
    };
    // 64 rays per workgroup, FOUR lanes per ray (end of round 4): lane `part` of a ray's quad walks entries [64 part, 64 part + 64) of every tile
    // of 256 rays, four entries per LDS read; the four partial sums are added in the fixed order ((p0 + p1) + (p2 + p3)).  With one lane per
    // ray and one entry per read this 4-workgroup kernel was 67 us of dependent LDS latency, the longest small launch of the first phase.
    const int part = threadIdx.x & 3;
    const int i = blockIdx.x * 64 + (threadIdx.x >> 2);
    const int ti = i < R ? target(i) : -1;
    bool owner = i < R;
    float sum = 0.f;
    for (int base = 0; base < R; base += 256) {
        const int j = base + threadIdx.x;
        __syncthreads();
        tgt[threadIdx.x] = j < R ? target(j) : -2;      // (entries past R: a target no ray has)
        gv[threadIdx.x] = j < R ? g[j] : 0.f;
        __syncthreads();
#pragma unroll 4
        for (int k = 64 * part; k < 64 * part + 64; k += 4) {
            const i32x4 t4 = *reinterpret_cast<const i32x4*>(tgt + k);
            const f32x4 g4 = *reinterpret_cast<const f32x4*>(gv + k);
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (t4[e] == ti) {
                    sum += g4[e];
                    if (base + k + e < i) owner = false;
                }
        }
    }
    sum += __shfl_xor(sum, 1);
    sum += __shfl_xor(sum, 2);
    int own = owner ? 1 : 0;
    own &= __shfl_xor(own, 1);
    own &= __shfl_xor(own, 2);
    if (own && part == 0) g_img[ti] += sum;     // the only writer of this pixel
}

// The same gather with the per-image affine depth distortion applied to the R gathered values instead of the whole map
// (model/training.py:240-245 distorts the full image, model/network.py:22-24 then picks R of its pixels: the same numbers).
// out = raw * scale + shift, or (raw + shift) * scale with shift_first; scale / shift are one-element device tensors.
__global__ void depth_gather_affine_fwd_kernel(const float* img, const int64_t* ray_idx, const float* scale, const float* shift,
                                               int shift_first, float* out, int R, int h, int w, int hd, int wd) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R) return;
    const int64_t q = ray_idx[i];
    const int y = (int)(q / w), x = (int)(q - (int64_t)y * w);
    const float raw = img[(int64_t)nearest_src(y, h, hd) * wd + nearest_src(x, w, wd)];
    out[i] = shift_first ? __fmul_rn(__fadd_rn(raw, shift[0]), scale[0]) : __fadd_rn(__fmul_rn(raw, scale[0]), shift[0]);
}
// g_ss[0] = d loss / d scale, g_ss[1] = d loss / d shift; the raw map has no gradient.  One workgroup, fixed summation order
// (see ray_setup_bwd_kernel): the distortion gradients are bit-reproducible.
__global__ __launch_bounds__(1024) void depth_gather_affine_bwd_kernel(const float* g, const float* img, const int64_t* ray_idx,
                                                                       const float* scale, const float* shift, int shift_first,
                                                                       float* g_ss, int R, int h, int w, int hd, int wd) {
    __shared__ float red[2][16];
    float gs = 0.f, gh = 0.f;
    for (int i = threadIdx.x; i < R; i += 1024) {
        const int64_t q = ray_idx[i];
        const int y = (int)(q / w), x = (int)(q - (int64_t)y * w);
        const float raw = img[(int64_t)nearest_src(y, h, hd) * wd + nearest_src(x, w, wd)];
        const float gi = g[i];
        // a non-finite raw depth (masked ray) carries a zero upstream gradient; keep 0 * inf out of the sums
        if (gi != 0.f) {
            gs += shift_first ? gi * (raw + shift[0]) : gi * raw;
            gh += shift_first ? gi * scale[0] : gi;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        gs += __shfl_xor(gs, o, 64);
        gh += __shfl_xor(gh, o, 64);
    }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = gs; red[1][threadIdx.x >> 6] = gh; }
    __syncthreads();
    if (threadIdx.x < 2) {
        float t = 0.f;
        for (int k = 0; k < 16; ++k) t += red[threadIdx.x][k];
        g_ss[threadIdx.x] = t;
    }
}

// scaled pixel coordinates of flat indices: x' = 2 x/(w-1) - 1, y' = 2 y/(h-1) - 1, same float op order as arange_pixels
// (model/common.py:36-39)
__global__ void pixels_from_index_kernel(const int64_t* idx, float* out, int R, int h, int w) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R) return;
    const int64_t q = idx[i];
    const int y = (int)(q / w), x = (int)(q - (int64_t)y * w);
    out[2 * i] = __fsub_rn(__fdiv_rn(__fmul_rn(2.0f, (float)x), (float)(w - 1)), 1.0f);
    out[2 * i + 1] = __fsub_rn(__fdiv_rn(__fmul_rn(2.0f, (float)y), (float)(h - 1)), 1.0f);
}
hipError_t launch_pixels_from_index(const int64_t* idx, float* out, int R, int h, int w, hipStream_t st) {
    hipLaunchKernelGGL(pixels_from_index_kernel, dim3((R + 255) / 256), dim3(256), 0, st, idx, out, R, h, w);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ fused front end of a training step
// Everything between the learnable tables and the per-ray inputs of the render operator in ONE launch each way (model/training.py:
// 235-262 + model/network.py:22-24 + model/rendering.py:54-87 of the reference): pose row -> c2w (se3_exp_matrix) -> world_mat = c2w^-1
// (the trainer's torch.inverse), the frame's depth distortion (scale floored at the constant 0.01, the last camera's scale pinned to
// 1: model/distortions.py:19-26), pixel coordinates, the nearest-resize depth gather with the distortion applied to the gathered values,
// the colour targets, and ray_setup -- the arithmetic of se3_exp_fwd / inv4_fwd / pixels_from_index / depth_gather_affine_fwd /
// ray_setup_fwd in the same order, so the numbers are those of the separate launches.  Every thread rebuilds the five 4x4 matrices
// (a few hundred flops) instead of waiting for a one-thread kernel to publish them.
__device__ __forceinline__ void step_scale_shift(const StepRaysArgs& a, float& scale, float& shift, bool& scale_live) {
    shift = a.shifts[a.cam];
    scale = 1.f;
    scale_live = false;
    if (!(a.fix_last_scale && a.cam == a.n_cams - 1)) {       // the gauge: the last view's depth scale is 1
        const float raw = a.scales[a.cam];
        scale_live = !(raw < 0.01f);                             // below the floor the scale is the CONSTANT 0.01: no gradient
        scale = scale_live ? raw : 0.01f;
    }
}
// the reference camera of the per-image losses: its pose, the relative transform of the pair, its depth distortion
struct PairGeom { M4 c2w_ref, ref_rt, other_inv, rel; float scale_ref, shift_ref; bool ref_live; };
__device__ __forceinline__ PairGeom pair_geometry(const StepRaysArgs& a, const M4& W) {
    PairGeom g;
    se3_exp_matrix(a.r_all + 3 * a.ref, a.t_all + 3 * a.ref, g.c2w_ref.m);
    g.ref_rt = inv4(g.c2w_ref);
    if (a.cam < a.n_cams - 1) {
        g.other_inv = inv4(W);                       // inverse(world_mat): the trainer inverts numerically, so does this
        g.rel = mul4(g.ref_rt, g.other_inv);
    } else {
        g.other_inv = inv4(g.ref_rt);
        g.rel = mul4(W, g.other_inv);
    }
    g.shift_ref = a.shifts[a.ref];
    g.scale_ref = 1.f;
    g.ref_live = false;
    if (!(a.fix_last_scale && a.ref == a.n_cams - 1)) {
        const float raw = a.scales[a.ref];
        g.ref_live = !(raw < 0.01f);
        g.scale_ref = g.ref_live ? raw : 0.01f;
    }
    return g;
}
__device__ __forceinline__ void step_matrices(const StepRaysArgs& a, M4& c2w, M4& W, RaySetupArgs& rs) {
    se3_exp_matrix(a.r_all + 3 * a.cam, a.t_all + 3 * a.cam, c2w.m);
    W = inv4(c2w);
    rs.K = a.K; rs.W = W.m; rs.S = a.S;
}
__device__ __forceinline__ void step_pixel(const StepRaysArgs& a, int i, float& px, float& py, float& raw) {
    const int64_t q = a.ray_idx[i];
    const int y = (int)(q / a.w), x = (int)(q - (int64_t)y * a.w);
    px = __fsub_rn(__fdiv_rn(__fmul_rn(2.0f, (float)x), (float)(a.w - 1)), 1.0f);
    py = __fsub_rn(__fdiv_rn(__fmul_rn(2.0f, (float)y), (float)(a.h - 1)), 1.0f);
    raw = a.depth_img[(int64_t)nearest_src(y, a.h, a.hd) * a.wd + nearest_src(x, a.w, a.wd)];
}

__global__ __launch_bounds__(256) void step_rays_fwd_kernel(StepRaysArgs a) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    M4 c2w, W, kinv, winv, sinv, m;
    RaySetupArgs rs{};
    step_matrices(a, c2w, W, rs);
    pixel_to_world(rs, kinv, winv, sinv, m);
    float scale, shift;
    bool live;
    step_scale_shift(a, scale, shift, live);
    if (i == 0) {
        for (int k = 0; k < 16; ++k) { a.mats[k] = c2w.m[k]; a.mats[16 + k] = W.m[k]; }
        a.mats[32] = scale;
        a.mats[33] = shift;
        if (a.ref >= 0) {
            // The frame pair of the per-image losses (model/training.py:280-313): rel = ref_rt inverse(world_mat), or -- the last camera
            // takes the roles the other way round -- world_mat inverse(ref_rt), with ref_rt = inverse(c2w_ref); the two distortions in the
            // order (first cloud, second cloud) of nnr_aux_terms_*; scale2 = the second cloud's scale.
            PairGeom pg = pair_geometry(a, W);
            for (int k = 0; k < 16; ++k) a.mats[34 + k] = pg.rel.m[k];
            const bool swap = a.cam == a.n_cams - 1;
            a.mats[50] = swap ? pg.scale_ref : scale;  a.mats[51] = swap ? pg.shift_ref : shift;
            a.mats[52] = swap ? scale : pg.scale_ref;  a.mats[53] = swap ? shift : pg.shift_ref;
            a.mats[54] = swap ? scale : pg.scale_ref;
            a.mats[55] = 0.f;
        }
    }
    if (i >= a.R) return;
    float px, py, raw;
    step_pixel(a, i, px, py, raw);
    const float dep = a.shift_first ? __fmul_rn(__fadd_rn(raw, shift), scale) : __fadd_rn(__fmul_rn(raw, scale), shift);
    a.pixels[2 * i] = px;
    a.pixels[2 * i + 1] = py;
    if (a.img) {
        const int64_t q = a.ray_idx[i], hw = (int64_t)a.h * a.w;
#pragma unroll
        for (int c = 0; c < 3; ++c) a.rgb_gt[3 * i + c] = a.img[c * hw + q];
    }
    float ray[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) ray[c] = m.m[4 * c] * px + m.m[4 * c + 1] * py + m.m[4 * c + 2];   // pixels_world - camera_world
    const float n = sqrtf(ray[0] * ray[0] + ray[1] * ray[1] + ray[2] * ray[2]);
    const float q0 = ray[0] * dep, q1 = ray[1] * dep, q2 = ray[2] * dep;
    float dgt = sqrtf(q0 * q0 + q1 * q1 + q2 * q2);                                              // |points_world - camera_world|
    if (!a.normalise) dgt = dgt / n;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float d = a.normalise ? ray[c] / n : ray[c];
        a.pts_o[3 * i + c] = m.m[4 * c + 3];
        a.dir[3 * i + c] = d;
        a.view[3 * i + c] = a.use_dir ? -d : 1.f;
    }
    a.ray_norm[i] = n;
    a.d_gt[i] = dgt;
    a.mask[i] = (isfinite(dgt) && dgt != 0.f) ? 1 : 0;
}

// The whole way back in one launch: the per-ray part of ray_setup_bwd with the depth gradient fed straight into the distortion sums
// (depth_gather_affine_bwd), reduced in a FIXED order -- per workgroup: strided rays, lane tree, the 8 waves in index order; then the
// workgroups' partials in workgroup order by the LAST workgroup to finish (a ticket counter in the caller's scratch: round 5, before that
// ONE workgroup walked all rays, 8 per thread at 4096 rays) -- then, in that workgroup, the matrix chain rule M = S^-1 W^-1 K^-1,
// W = c2w^-1 (inv4_bwd), c2w = exp(r, t) (se3_exp_grad) -- and the full (n_cams, .) gradient tables written, zeros outside the frame's
// row.  Replaces 4 launches + the ~10 tiny ATen kernels of the autograd of `where` / indexing in Learn_Distortion.
constexpr int kStepBwdThreads = 512;   // 8 waves: 256 registers each (at 1024 threads the matrix chain spills 700 bytes per lane)
constexpr int kStepBwdMaxBlocks = 16;  // partial slots in the scratch (NNR_STEP_BWD_SCRATCH_FLOATS = 16 + 2 * 16 * 16: the partial sums are doubles)
__global__ __launch_bounds__(kStepBwdThreads) void step_rays_bwd_kernel(StepRaysArgs a) {
    __shared__ double red[14][kStepBwdThreads / 64];
    __shared__ float park[4][16];
    __shared__ double tot[16];
    __shared__ int is_last;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    double acc[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) acc[k] = 0.;
    double gs = 0., gh = 0.;
    M4 m;      // only the product survives the ray loop (six live 4x4 matrices per thread spill); thread 0 parks the others in LDS for
    {          // the chain rule at the end instead of rebuilding them there (the serial tail was two thirds of the kernel)
        M4 c2w, W, kinv, winv, sinv;
        RaySetupArgs rs{};
        step_matrices(a, c2w, W, rs);
        pixel_to_world(rs, kinv, winv, sinv, m);
        if (threadIdx.x == 0) {
#pragma unroll
            for (int k = 0; k < 16; ++k) { park[0][k] = kinv.m[k]; park[1][k] = winv.m[k]; park[2][k] = sinv.m[k]; park[3][k] = W.m[k]; }
        }
    }
    float scale, shift;
    bool live;
    step_scale_shift(a, scale, shift, live);
    for (int i = blockIdx.x * kStepBwdThreads + threadIdx.x; i < a.R; i += gridDim.x * kStepBwdThreads) {
        float px, py, raw;
        step_pixel(a, i, px, py, raw);
        const float dep = a.shift_first ? __fmul_rn(__fadd_rn(raw, shift), scale) : __fadd_rn(__fmul_rn(raw, scale), shift);
        const RayBack rb = ray_backward(m, px, py, dep, a.g_dir, a.g_view, a.g_norm, a.g_dgt, i, a.use_dir, a.normalise);
        // a non-finite raw depth (masked ray) carries a zero upstream gradient; keep 0 * inf out of the sums
        if (rb.gdep != 0.) {
            gs += a.shift_first ? rb.gdep * ((double)raw + (double)shift) : rb.gdep * (double)raw;
            gh += a.shift_first ? rb.gdep * (double)scale : rb.gdep;
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            acc[4 * c + 0] += rb.gray[c] * px;
            acc[4 * c + 1] += rb.gray[c] * py;
            acc[4 * c + 2] += rb.gray[c];
            acc[4 * c + 3] += a.g_o ? (double)a.g_o[3 * i + c] : 0.;
        }
    }
#pragma unroll
    for (int k = 0; k < 14; ++k) {
        double v = k < 12 ? acc[k] : (k == 12 ? gs : gh);
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
        if (lane == 0) red[k][wv] = v;
    }
    __syncthreads();
    // this workgroup's 14 sums (waves in index order) -> its partial slot; the last workgroup to arrive adds the slots in workgroup order
    double* const slots = reinterpret_cast<double*>(a.bwd_scratch + 16);      // (the scratch comes from a device allocator: 8-byte aligned at + 64 bytes)
    if (threadIdx.x < 14) {
        double t = 0.;
        for (int w = 0; w < kStepBwdThreads / 64; ++w) t += red[threadIdx.x][w];
        slots[16 * blockIdx.x + threadIdx.x] = t;
    }
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned int ticket = atomicAdd(reinterpret_cast<unsigned int*>(a.bwd_scratch), 1u);
        is_last = ticket == gridDim.x - 1u;
    }
    __syncthreads();
    if (!is_last) return;
    __threadfence();
    if (threadIdx.x < 14) {
        double t = 0.;
        for (unsigned int b = 0; b < gridDim.x; ++b) t += slots[16 * b + threadIdx.x];
        tot[threadIdx.x] = t;
    }
    for (int i = threadIdx.x; i < 3 * a.n_cams; i += kStepBwdThreads) { a.d_r[i] = 0.f; a.d_t[i] = 0.f; }
    for (int i = threadIdx.x; i < a.n_cams; i += kStepBwdThreads) { a.d_scales[i] = 0.f; a.d_shifts[i] = 0.f; }
    __threadfence_block();
    __syncthreads();
    if (threadIdx.x != 0) return;
    *reinterpret_cast<unsigned int*>(a.bwd_scratch) = 0u;      // the ticket counter as the next call expects it
    M4d dm;
    for (int k = 0; k < 12; ++k) dm.m[k] = tot[k];
    for (int k = 12; k < 16; ++k) dm.m[k] = 0.;
    double tot_s = tot[12], tot_h = tot[13];
    // dL/dM (12 sums) -> dL/dW through M = (S^-1 W^-1) K^-1, W^-1 = inv4(W); then W = inv4(c2w); then c2w = exp(r, t)
    const M4 kinv = load4(park[0]), winv = load4(park[1]), sinv = load4(park[2]), W = load4(park[3]);
    const M4d d_sw = mul4(dm, transpose4(kinv));
    const M4d d_winv = mul4(transpose4(sinv), d_sw);
    M4d gw = inv_backward(winv, d_winv);                      // d loss / d world_mat
    if (a.ref >= 0 && a.g_mats) {
        // the per-image losses' share (upstream gradient of mats[34, 55)): the relative transform chains into world_mat (and, unless the
        // reference side is detached -- training.detach_ref_img, the default -- into the reference pose), the pair's distortion entries
        // into the two cameras' rows
        const PairGeom pg = pair_geometry(a, W);
        M4d g_rel;
        for (int k = 0; k < 12; ++k) g_rel.m[k] = (double)a.g_mats[34 + k];
        for (int k = 12; k < 16; ++k) g_rel.m[k] = 0.;
        const bool swap = a.cam == a.n_cams - 1;
        M4d g_ref_rt;
        if (!swap) {      // rel = ref_rt inv(W)
            g_ref_rt = mul4(g_rel, transpose4(pg.other_inv));
            const M4d g_winv = mul4(transpose4(pg.ref_rt), g_rel);
            const M4d add = inv_backward(pg.other_inv, g_winv);
            for (int k = 0; k < 16; ++k) gw.m[k] += add.m[k];
        } else {          // rel = W inv(ref_rt)
            const M4d add = mul4(g_rel, transpose4(pg.other_inv));
            for (int k = 0; k < 16; ++k) gw.m[k] += add.m[k];
            const M4d g_inv = mul4(transpose4(W), g_rel);
            g_ref_rt = inv_backward(pg.other_inv, g_inv);
        }
        const double g_s_in = (double)a.g_mats[swap ? 52 : 50] + (swap ? (double)a.g_mats[54] : 0.), g_h_in = a.g_mats[swap ? 53 : 51];
        const double g_s_ref = (double)a.g_mats[swap ? 50 : 52] + (swap ? 0. : (double)a.g_mats[54]), g_h_ref = a.g_mats[swap ? 51 : 53];
        tot_s += g_s_in;
        tot_h += g_h_in;
        if (!a.detach_ref) {
            const M4d g_c2w_ref = inv_backward(pg.ref_rt, g_ref_rt);
            se3_exp_grad(a.r_all + 3 * a.ref, g_c2w_ref.m, a.d_r + 3 * a.ref, a.d_t + 3 * a.ref);
            if (pg.ref_live) a.d_scales[a.ref] = (float)g_s_ref;
            a.d_shifts[a.ref] = (float)g_h_ref;
        }
    }
    if (live) a.d_scales[a.cam] = (float)tot_s;
    a.d_shifts[a.cam] = (float)tot_h;
    const M4d gc = inv_backward(W, gw);                       // d loss / d c2w
    se3_exp_grad(a.r_all + 3 * a.cam, gc.m, a.d_r + 3 * a.cam, a.d_t + 3 * a.cam);
}

// ------------------------------------------------------------------------------------------------ loss heads
__device__ __forceinline__ float block_sum(float v, float* sm) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sm[wv] = v;
    __syncthreads();
    float t = 0.f;
    for (int k = 0; k < (int)(blockDim.x >> 6); ++k) t += sm[k];
    return t;
}

// one workgroup; rgb: sum |diff| (or diff^2) / R_total; depth: sum_valid |pred - gt| / M_total   (losses.py:27-32,59-64)
__global__ __launch_bounds__(1024) void render_loss_kernel(LossArgs a) {
    __shared__ float sm[16];
    float s_rgb = 0.f, s_l2 = 0.f, s_dep = 0.f, cnt = 0.f;
    for (int i = threadIdx.x; i < a.R; i += blockDim.x) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float d = a.rgb[3 * i + c] - a.rgb_gt[3 * i + c];
            s_rgb += a.rgb_l2 ? d * d : fabsf(d);
            s_l2 += d * d;
        }
        if (a.mask[i]) {
            float gt = a.d_gt[i];
            if (a.ndc) gt = 1.f - 1.f / gt;                                   // rendering.py:157-158
            s_dep += fabsf(a.dist[i] - gt);
            cnt += 1.f;
        }
    }
    s_rgb = block_sum(s_rgb, sm);
    s_l2 = block_sum(s_l2, sm);
    s_dep = block_sum(s_dep, sm);
    cnt = block_sum(cnt, sm);
    const float mt = a.m_total_dev ? *a.m_total_dev : a.m_total;
    const float m = mt >= 0.f ? mt : cnt;
    const float inv_r = 1.f / a.r_total, inv_m = m > 0.f ? 1.f / m : 0.f;
    if (threadIdx.x == 0) {
        // a term whose weight is 0 is not evaluated by the reference and reported as 0 (model/losses.py:164-171,190-193);
        // l2_mean only if one of the two render terms is on
        const float lrgb = a.w_rgb != 0.f ? s_rgb * inv_r : 0.f, ldep = a.w_depth != 0.f ? s_dep * inv_m : 0.f;
        a.out[0] = (a.w_rgb != 0.f ? a.w_rgb * lrgb : 0.f) + (a.w_depth != 0.f ? a.w_depth * ldep : 0.f);
        a.out[1] = lrgb;
        a.out[2] = ldep;
        a.out[3] = (a.w_rgb != 0.f || a.w_depth != 0.f) ? s_l2 / (3.f * a.r_total) : 0.f;
        a.out[4] = cnt;
    }
    const float sign_eps = 0.f;
    for (int i = threadIdx.x; i < a.R; i += blockDim.x) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float d = a.rgb[3 * i + c] - a.rgb_gt[3 * i + c];
            const float g = a.rgb_l2 ? 2.f * d : (d > sign_eps ? 1.f : (d < -sign_eps ? -1.f : 0.f));
            a.g_rgb[3 * i + c] = a.w_rgb * inv_r * g;
        }
        float gd = 0.f, gg = 0.f;
        if (a.mask[i]) {
            float gt = a.d_gt[i];
            const float raw_gt = gt;
            if (a.ndc) gt = 1.f - 1.f / gt;
            const float d = a.dist[i] - gt;
            const float sg = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
            gd = a.w_depth * inv_m * sg;
            gg = a.detach_gt ? 0.f : -gd * (a.ndc ? 1.f / (raw_gt * raw_gt) : 1.f);
        }
        a.g_dist[i] = gd;
        a.g_dgt[i] = gg;
    }
}

// ------------------------------------------------------------------------------------------------ launchers
hipError_t launch_se3_exp_fwd(const float* r_all, const float* t_all, int idx, float* c2w, hipStream_t st) {
    hipLaunchKernelGGL(se3_exp_fwd_kernel, dim3(1), dim3(64), 0, st, r_all, t_all, idx, c2w);
    return hipGetLastError();
}
hipError_t launch_se3_exp_bwd(const float* r_all, int idx, int n_cams, const float* d_c2w, float* d_r, float* d_t, hipStream_t st) {
    hipLaunchKernelGGL(se3_exp_bwd_kernel, dim3((3 * n_cams + 255) / 256), dim3(256), 0, st, r_all, idx, n_cams, d_c2w, d_r, d_t);
    return hipGetLastError();
}
hipError_t launch_inv4(const float* a, float* y, int batch, hipStream_t st) {
    hipLaunchKernelGGL(inv4_fwd_kernel, dim3((batch + 63) / 64), dim3(64), 0, st, a, y, batch);
    return hipGetLastError();
}
hipError_t launch_inv4_bwd(const float* y, const float* dy, float* da, int batch, hipStream_t st) {
    hipLaunchKernelGGL(inv4_bwd_kernel, dim3((batch + 63) / 64), dim3(64), 0, st, y, dy, da, batch);
    return hipGetLastError();
}
hipError_t launch_ray_setup_fwd(const RaySetupArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(ray_setup_fwd_kernel, dim3((a.R + 255) / 256), dim3(256), 0, st, a);
    return hipGetLastError();
}
hipError_t launch_ray_setup_bwd(const RaySetupArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(ray_setup_bwd_kernel, dim3(1), dim3(512), 0, st, a);
    return hipGetLastError();
}
hipError_t launch_step_rays_fwd(const StepRaysArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(step_rays_fwd_kernel, dim3((a.R + 255) / 256), dim3(256), 0, st, a);
    return hipGetLastError();
}
hipError_t launch_step_rays_bwd(const StepRaysArgs& a, hipStream_t st) {
    const int nb = std::min(kStepBwdMaxBlocks, std::max(1, (a.R + kStepBwdThreads - 1) / kStepBwdThreads));
    hipLaunchKernelGGL(step_rays_bwd_kernel, dim3(nb), dim3(kStepBwdThreads), 0, st, a);
    return hipGetLastError();
}
hipError_t launch_depth_gather_fwd(const float* img, const int64_t* idx, float* out, int R, int h, int w, int hd, int wd, hipStream_t st) {
    hipLaunchKernelGGL(depth_gather_fwd_kernel, dim3((R + 255) / 256), dim3(256), 0, st, img, idx, out, R, h, w, hd, wd);
    return hipGetLastError();
}
hipError_t launch_depth_gather_bwd(const float* g, const int64_t* idx, float* g_img, int R, int h, int w, int hd, int wd, hipStream_t st) {
    hipError_t e = hipMemsetAsync(g_img, 0, (size_t)hd * wd * sizeof(float), st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(depth_gather_bwd_kernel, dim3((R + 63) / 64), dim3(256), 0, st, g, idx, g_img, R, h, w, hd, wd);
    return hipGetLastError();
}
// ---- NDC rays for forward-facing scenes (model/common.py:632-675, called from Renderer.sample_ndc, rendering.py:168-180) ----
// Per ray: shift the origin to the near plane, project origin and direction with (gx, gy) = (-1/(1/K00), -1/(1/K11)) -- the
// reference writes the focal that way, the double reciprocal is kept.  ~25 torch ops forward and twice that backward become
// one launch each.  Unfused mul/add/div in the forward, in the reference's operation order.
__global__ void ndc_rays_fwd_kernel(const float* __restrict__ o, const float* __restrict__ d, const float* __restrict__ K, float near_,
                                    float* __restrict__ o_ndc, float* __restrict__ d_ndc, int R) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R) return;
    const float gx = __fdiv_rn(-1.f, __fdiv_rn(1.f, K[0])), gy = __fdiv_rn(-1.f, __fdiv_rn(1.f, K[5]));
    const float ox = o[3 * i], oy = o[3 * i + 1], oz = o[3 * i + 2], dx = d[3 * i], dy = d[3 * i + 1], dz = d[3 * i + 2];
    const float t = __fdiv_rn(-__fadd_rn(near_, oz), dz);
    const float px = __fadd_rn(ox, __fmul_rn(t, dx)), py = __fadd_rn(oy, __fmul_rn(t, dy)), pz = __fadd_rn(oz, __fmul_rn(t, dz));
    const float u = __fdiv_rn(px, pz), v = __fdiv_rn(py, pz);
    const float onz = __fadd_rn(1.f, __fdiv_rn(__fmul_rn(2.f, near_), pz));
    o_ndc[3 * i] = __fmul_rn(gx, u);
    o_ndc[3 * i + 1] = __fmul_rn(gy, v);
    o_ndc[3 * i + 2] = onz;
    d_ndc[3 * i] = __fmul_rn(gx, __fsub_rn(__fdiv_rn(dx, dz), u));
    d_ndc[3 * i + 1] = __fmul_rn(gy, __fsub_rn(__fdiv_rn(dy, dz), v));
    d_ndc[3 * i + 2] = __fsub_rn(1.f, onz);
}
// gradients with respect to the world rays (the intrinsics are treated as constants: a learnable focal takes the torch path)
__global__ void ndc_rays_bwd_kernel(const float* __restrict__ o, const float* __restrict__ d, const float* __restrict__ K, float near_,
                                    const float* __restrict__ g_o_ndc, const float* __restrict__ g_d_ndc, float* __restrict__ g_o,
                                    float* __restrict__ g_d, int R) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R) return;
    const float gx = -1.f / (1.f / K[0]), gy = -1.f / (1.f / K[5]);
    const float ox = o[3 * i], oy = o[3 * i + 1], oz = o[3 * i + 2], dx = d[3 * i], dy = d[3 * i + 1], dz = d[3 * i + 2];
    const float a = near_ + oz, t = -a / dz;
    const float px = ox + t * dx, py = oy + t * dy, pz = oz + t * dz;
    const float gox = g_o_ndc[3 * i], goy = g_o_ndc[3 * i + 1], goz = g_o_ndc[3 * i + 2];
    const float gdx = g_d_ndc[3 * i], gdy = g_d_ndc[3 * i + 1], gdz = g_d_ndc[3 * i + 2];
    const float g_u = gx * (gox - gdx), g_v = gy * (goy - gdy);            // u = px/pz, v = py/pz enter o_ndc.xy and -d_ndc.xy
    const float g_rx = gx * gdx, g_ry = gy * gdy;                          // rx = dx/dz, ry = dy/dz
    const float ipz = 1.f / pz, idz = 1.f / dz;
    const float g_px = g_u * ipz, g_py = g_v * ipz;
    const float g_pz = -(g_u * px + g_v * py) * ipz * ipz - (goz - gdz) * 2.f * near_ * ipz * ipz;   // o_ndc.z = 1 + 2 near / pz = 1 - d_ndc.z
    const float g_t = g_px * dx + g_py * dy + g_pz * dz;                   // p = o + t d
    const float g_a = -g_t * idz;                                          // t = -a / dz
    g_o[3 * i] = g_px;
    g_o[3 * i + 1] = g_py;
    g_o[3 * i + 2] = g_pz + g_a;                                           // a = near + oz
    g_d[3 * i] = t * g_px + g_rx * idz;
    g_d[3 * i + 1] = t * g_py + g_ry * idz;
    g_d[3 * i + 2] = t * g_pz + g_t * a * idz * idz - (g_rx * dx + g_ry * dy) * idz * idz;
}
hipError_t launch_ndc_rays_fwd(const float* o, const float* d, const float* K, float near_, float* o_ndc, float* d_ndc, int R, hipStream_t st) {
    hipLaunchKernelGGL(ndc_rays_fwd_kernel, dim3((R + 255) / 256), dim3(256), 0, st, o, d, K, near_, o_ndc, d_ndc, R);
    return hipGetLastError();
}
hipError_t launch_ndc_rays_bwd(const float* o, const float* d, const float* K, float near_, const float* g_o_ndc, const float* g_d_ndc,
                               float* g_o, float* g_d, int R, hipStream_t st) {
    hipLaunchKernelGGL(ndc_rays_bwd_kernel, dim3((R + 255) / 256), dim3(256), 0, st, o, d, K, near_, g_o_ndc, g_d_ndc, g_o, g_d, R);
    return hipGetLastError();
}

hipError_t launch_depth_gather_affine_fwd(const float* img, const int64_t* idx, const float* scale, const float* shift, int shift_first,
                                          float* out, int R, int h, int w, int hd, int wd, hipStream_t st) {
    hipLaunchKernelGGL(depth_gather_affine_fwd_kernel, dim3((R + 255) / 256), dim3(256), 0, st, img, idx, scale, shift, shift_first, out, R, h,
                       w, hd, wd);
    return hipGetLastError();
}
hipError_t launch_depth_gather_affine_bwd(const float* g, const float* img, const int64_t* idx, const float* scale, const float* shift,
                                          int shift_first, float* g_ss, int R, int h, int w, int hd, int wd, hipStream_t st) {
    hipLaunchKernelGGL(depth_gather_affine_bwd_kernel, dim3(1), dim3(1024), 0, st, g, img, idx, scale, shift, shift_first, g_ss,
                       R, h, w, hd, wd);
    return hipGetLastError();
}
hipError_t launch_render_loss(const LossArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(render_loss_kernel, dim3(1), dim3(1024), 0, st, a);
    return hipGetLastError();
}

}  // namespace nnr
