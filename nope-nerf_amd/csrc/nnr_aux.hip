// nnr_aux.hip -- the per-image losses between a frame and its reference frame, fused (SURVEY 8 f1 + f2).
// Replaces, per training step while pc_weight / rgb_s_weight > 0 (the whole first training phase), the ~290 small torch
// launches of reference model/training.py:315-358 (nearest-resized depths, back-projection of both depth maps, relative
// transform, re-projection, two bilinear grid_samples) and model/losses.py:114-157 (point-cloud loss, surface
// re-projection loss, with_ssim off) and their autograd, by one per-point forward kernel, the nearest-neighbour kernels
// of nnr_pointcloud.hip and one per-point backward kernel.  O(points) work (32 400 at 540x960, pc_ratio 4): latency /
// launch-count bound, not bandwidth or FLOP bound; the only heavy part is the nearest-neighbour search.
//
// Per point i of the (hr, wr) sampling grid, x' = 2x/(wr-1)-1, y' = 2y/(hr-1)-1 (arange_pixels, model/common.py:13-40):
//   d1 = max(nearest-resize(d1_img)[i], nl), d2 likewise               (training.py:318-321; no gradient where clamped)
//   pc1 = Kinv [x' d1, y' d1, d1, 1],  pc2 likewise                     (transform_to_world, common.py:112-160)
//   rot = R pc1 + t                                                     (training.py:330,353)
//   X = rot / scale2, Y = pc2 / scale2                                  (:355-357)  -> point-cloud loss (losses.py:114-148)
//   rot' = (nl,nl,nl) where -rot.z < nl;  xy = (K [rot';1]).xy / .z;  valid = max|xy| <= 1     (:331-333, common.py:436-457)
//   rgb_s = mean over valid points and 3 channels of clamp(|img1r(x',y') - img2r(xy)|, 0, 1)     (losses.py:150-157)
// with_ssim (NNR_AUX_SSIM): the per-point term becomes 0.15 clamp|.| + 0.85 SSIM, where the reference hands the (1, hr, wr, 3) colour
// tensors to its NCHW SSIM module (losses.py:222-252): the 3x3 reflect-padded average pool therefore runs over (grid x, colour
// channel) -- the window of (y, x, c) is x-1..x+1 times c-1..c+1, reflected at the row ends and at the channel ends -- and that is
// what aux_ssim_kernel restates, gradient included.
#include "../../include/nnr.h"
#include "nnr_device.h"
#include "nnr_kernels.h"
#include <cstdlib>
#include <string>
#include <type_traits>

namespace nnr {

hipError_t launch_pc_nearest_keys_at(const float* src, const float* dst, int S, int D, unsigned long long* keys, int s_off, hipStream_t st);   // nnr_pointcloud.hip

__device__ __forceinline__ int aux_nearest_src(int dst, int dst_size, int src_size) {   // F.interpolate(mode='nearest')
    const float scale = (float)src_size / (float)dst_size;
    const int s = (int)floorf((float)dst * scale);
    return s < src_size - 1 ? s : src_size - 1;
}

// grid_sample(mode='bilinear', padding_mode='zeros', align_corners=True) of a (3, h, w) image at (gx, gy) in [-1,1]; also the
// derivative of every channel with respect to (gx, gy)
struct Bilinear {
    float v[3], dx[3], dy[3];
};
__device__ __forceinline__ Bilinear sample_bilinear(const float* img, int h, int w, float gx, float gy) {
    const float ix = (gx + 1.f) * 0.5f * (float)(w - 1), iy = (gy + 1.f) * 0.5f * (float)(h - 1);
    const float fx0 = floorf(ix), fy0 = floorf(iy);
    const int x0 = (int)fx0, y0 = (int)fy0, x1 = x0 + 1, y1 = y0 + 1;
    const float ax = ix - fx0, ay = iy - fy0;
    const bool okx0 = x0 >= 0 && x0 < w, okx1 = x1 >= 0 && x1 < w, oky0 = y0 >= 0 && y0 < h, oky1 = y1 >= 0 && y1 < h;
    Bilinear r;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float* p = img + (int64_t)c * h * w;
        const float v00 = okx0 && oky0 ? p[y0 * w + x0] : 0.f, v01 = okx1 && oky0 ? p[y0 * w + x1] : 0.f;
        const float v10 = okx0 && oky1 ? p[y1 * w + x0] : 0.f, v11 = okx1 && oky1 ? p[y1 * w + x1] : 0.f;
        r.v[c] = (v00 * (1.f - ax) + v01 * ax) * (1.f - ay) + (v10 * (1.f - ax) + v11 * ax) * ay;
        r.dx[c] = ((v01 - v00) * (1.f - ay) + (v11 - v10) * ay) * 0.5f * (float)(w - 1);
        r.dy[c] = ((v10 - v00) * (1.f - ax) + (v11 - v01) * ax) * 0.5f * (float)(h - 1);
    }
    return r;
}

enum : uint32_t { kClamp1 = 1, kClamp2 = 2, kBehind = 4, kValid = 8 };

struct AuxGeom {   // shared by the forward and the backward kernel
    float xp, yp;
    int src1;       // flat index of the source pixel in the (hd, wd) depth images
    float d1, d2;
    uint32_t flags;
    float pc1[3], pc2[3], rot[3];
};

__device__ __forceinline__ AuxGeom aux_geometry(const AuxArgs& a, int i) {
    AuxGeom g;
    const int y = i / a.wr, x = i - y * a.wr;
    g.xp = 2.f * (float)x / (float)(a.wr - 1) - 1.f;
    g.yp = 2.f * (float)y / (float)(a.hr - 1) - 1.f;
    g.src1 = aux_nearest_src(y, a.hr, a.hd) * a.wd + aux_nearest_src(x, a.wr, a.wd);
    g.d1 = a.d1_img[g.src1];
    g.d2 = a.d2_img[g.src1];
    if (a.aff) {      // NNR_AUX_AFFINE: the maps are the RAW mono depths, the per-image distortion is applied here (to the S sampled values
                      // instead of hd x wd, and without the two launches + their autograd per map): model/training.py:240-245, 294-296
        const float s1 = a.aff[0], t1 = a.aff[1], s2 = a.aff[2], t2 = a.aff[3];
        g.d1 = a.shift_first ? __fmul_rn(__fadd_rn(g.d1, t1), s1) : __fadd_rn(__fmul_rn(g.d1, s1), t1);
        g.d2 = a.shift_first ? __fmul_rn(__fadd_rn(g.d2, t2), s2) : __fadd_rn(__fmul_rn(g.d2, s2), t2);
    }
    g.flags = 0;
    if (g.d1 < a.nl) { g.d1 = a.nl; g.flags |= kClamp1; }
    if (g.d2 < a.nl) { g.d2 = a.nl; g.flags |= kClamp2; }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const float* k = a.Kinv + 4 * r;
        g.pc1[r] = k[0] * (g.xp * g.d1) + k[1] * (g.yp * g.d1) + k[2] * g.d1 + k[3];
        g.pc2[r] = k[0] * (g.xp * g.d2) + k[1] * (g.yp * g.d2) + k[2] * g.d2 + k[3];
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) g.rot[r] = a.rel[4 * r] * g.pc1[0] + a.rel[4 * r + 1] * g.pc1[1] + a.rel[4 * r + 2] * g.pc1[2] + a.rel[4 * r + 3];
    if (-g.rot[2] < a.nl) g.flags |= kBehind;
    return g;
}

// projection of the (possibly replaced) rotated point: q = K[:3,:] [p;1], xy = q.xy / q.z
__device__ __forceinline__ void aux_project(const AuxArgs& a, const AuxGeom& g, float (&q)[3], float (&xy)[2]) {
    float p[3] = {g.rot[0], g.rot[1], g.rot[2]};
    if (g.flags & kBehind) p[0] = p[1] = p[2] = a.nl;
#pragma unroll
    for (int r = 0; r < 3; ++r) q[r] = a.K[4 * r] * p[0] + a.K[4 * r + 1] * p[1] + a.K[4 * r + 2] * p[2] + a.K[4 * r + 3];
    xy[0] = q[0] / q[2];
    xy[1] = q[1] / q[2];
}

__device__ __forceinline__ float block_sum(float v, float* scratch) {   // sum over a 256-thread block, result in thread 0
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) scratch[wave] = v;
    __syncthreads();
    float r = 0.f;
    if (threadIdx.x == 0) r = scratch[0] + scratch[1] + scratch[2] + scratch[3];
    __syncthreads();
    return r;
}

// forward: X, Y for the nearest-neighbour search; the re-projection loss sum / count; d(point loss)/d(xy) for the backward
__global__ __launch_bounds__(256) void aux_points_fwd_kernel(AuxArgs a) {
    __shared__ float scratch[4];
    const int i = blockIdx.x * 256 + threadIdx.x;
    float lsum = 0.f, lcnt = 0.f;
    if (i < a.S) {
        const bool mine = i >= a.s_lo && i < a.s_hi;   // data-parallel shard of the source points: this rank's share of the SUMS
        const AuxGeom g = aux_geometry(a, i);
        const float s2 = (a.flags & NNR_AUX_SCALE_PCS) ? a.scale2[0] : 1.f;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            a.X[3 * i + r] = g.rot[r] / s2;
            a.Y[3 * i + r] = g.pc2[r] / s2;
            if (a.flags & NNR_AUX_PC) { a.gXq[3 * i + r] = 0; a.gYq[3 * i + r] = 0; }      // the backward's accumulators (was a memset launch there)
        }
        if (i == 0) *reinterpret_cast<unsigned int*>(a.acc + 7) = 0u;      // the ticket counter of aux_dist_sum_kernel
        if ((a.flags & NNR_AUX_PC) && i < 2 * (((a.wr + 7) / 8) * ((a.hr + 7) / 8)))      // the "heavy tile" flags of the search (behind the 8 tiles sphere floats)
            reinterpret_cast<int*>(a.keys)[8 * (((a.wr + 7) / 8) * ((a.hr + 7) / 8)) + i] = 0;
        float gxy0 = 0.f, gxy1 = 0.f;
        uint32_t fl = g.flags;
        if (a.flags & NNR_AUX_RGBS) {
            const bool ssim = (a.flags & NNR_AUX_SSIM) != 0;
            float q[3], xy[2];
            aux_project(a, g, q, xy);
            const bool valid = fmaxf(fabsf(xy[0]), fabsf(xy[1])) <= 1.f;
            if (valid) {
                fl |= kValid;
                lcnt = 1.f;
            }
            if (valid || ssim) {
                const Bilinear s1 = sample_bilinear(a.img1r, a.hr, a.wr, g.xp, g.yp);
                const Bilinear s2b = sample_bilinear(a.img2r, a.hr, a.wr, xy[0], xy[1]);
                if (ssim) {
                    // the windows of the neighbours need the colours of EVERY point (invalid ones too: their colour is a function of
                    // their re-projection like any other); the loss and d loss / d colour come from aux_ssim_kernel
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        a.rgb1[3 * i + c] = s1.v[c];
                        a.rgb2[3 * i + c] = s2b.v[c];
                        a.drgb[6 * i + c] = s2b.dx[c];
                        a.drgb[6 * i + 3 + c] = s2b.dy[c];
                    }
                } else {
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        const float u = s1.v[c] - s2b.v[c], au = fabsf(u);
                        if (mine) lsum += fminf(au, 1.f);
                        // d clamp(|u|,0,1)/d rgb2 = -sign(u) inside (0,1), 0 at the clamps (torch: clamp passes the gradient at the
                        // bounds, abs gives 0 at 0)
                        const float gu = (au > 0.f && au <= 1.f) ? (u > 0.f ? -1.f : 1.f) : 0.f;
                        gxy0 += gu * s2b.dx[c];
                        gxy1 += gu * s2b.dy[c];
                    }
                }
            }
            a.gxy[2 * i] = gxy0;
            a.gxy[2 * i + 1] = gxy1;
        }
        a.pflags[i] = fl;
    }
    // per-block partial sums, added up in block order by aux_finish_kernel: bit-reproducible (float atomics are not)
    const float bs = block_sum(lsum, scratch), bc = block_sum(lcnt, scratch);
    if (threadIdx.x == 0) {
        a.part_fwd[4 * blockIdx.x + 0] = bs;
        a.part_fwd[4 * blockIdx.x + 1] = bc;
    }
}

// with_ssim: loss and d loss / d (re-projected colour) of every point, after aux_points_fwd_kernel has left the colours and the validity
// of all points.  Point (y, x) owns its three loss terms (if valid and in this rank's shard) and GATHERS the gradient of its own
// colours from the <= 3 window centres per channel that read them: it re-evaluates the SSIM of the centres x-1, x, x+1 of its row
// (valid, in the shard) and differentiates each with respect to the taps that land on (y, x) -- no scatter, no atomics.
__device__ __forceinline__ int reflect1(int k, int n) {   // ReflectionPad2d(1): -1 -> 1, n -> n - 2
    return k < 0 ? 1 : (k >= n ? n - 2 : k);
}

__global__ __launch_bounds__(256) void aux_ssim_kernel(AuxArgs a) {
    __shared__ float scratch[4];
    const int i = blockIdx.x * 256 + threadIdx.x;
    constexpr float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f, ninth = 1.f / 9.f;
    float lsum = 0.f;
    if (i < a.S) {
        const int y = i / a.wr, x = i - y * a.wr;
        float g[3] = {0.f, 0.f, 0.f};
        if ((a.pflags[i] & kValid) && i >= a.s_lo && i < a.s_hi) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float u = a.rgb1[3 * i + c] - a.rgb2[3 * i + c], au = fabsf(u);
                lsum += 0.15f * fminf(au, 1.f);
                g[c] += (au > 0.f && au <= 1.f) ? (u > 0.f ? -0.15f : 0.15f) : 0.f;
            }
        }
        for (int dc = -1; dc <= 1; ++dc) {
            const int xc = x + dc, ic = i + dc;
            if (xc < 0 || xc >= a.wr || !(a.pflags[ic] & kValid) || ic < a.s_lo || ic >= a.s_hi) continue;
            int px[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) px[j] = y * a.wr + reflect1(xc + j - 1, a.wr);
            float xs[3][3], ys[3][3];   // [tap column j][channel]
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    xs[j][c] = a.rgb1[3 * px[j] + c];
                    ys[j][c] = a.rgb2[3 * px[j] + c];
                }
#pragma unroll
            for (int cc = 0; cc < 3; ++cc) {
                float sx = 0.f, sy = 0.f, sxx = 0.f, syy = 0.f, sxy = 0.f;
#pragma unroll
                for (int j = 0; j < 3; ++j)
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        const int tc = cc + k - 1 < 0 ? 1 : (cc + k - 1 > 2 ? 1 : cc + k - 1);
                        const float xv = xs[j][tc], yv = ys[j][tc];
                        sx += xv; sy += yv; sxx += xv * xv; syy += yv * yv; sxy += xv * yv;
                    }
                const float mx = sx * ninth, my = sy * ninth;
                const float vx = sxx * ninth - mx * mx, vy = syy * ninth - my * my, cv = sxy * ninth - mx * my;
                const float A1 = 2.f * mx * my + C1, A2 = 2.f * cv + C2, B1 = mx * mx + my * my + C1, B2 = vx + vy + C2;
                const float n = A1 * A2, d = B1 * B2;
                const float v = (1.f - n / d) * 0.5f;
                if (dc == 0) lsum += 0.85f * fminf(fmaxf(v, 0.f), 1.f);
                if (!(v >= 0.f && v <= 1.f)) continue;   // clamp: gradient only inside [0, 1] (bounds included, like torch)
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    if (px[j] != i) continue;
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        const int tc = cc + k - 1 < 0 ? 1 : (cc + k - 1 > 2 ? 1 : cc + k - 1);
                        const float xv = xs[j][tc], yv = ys[j][tc];
                        const float dA1 = 2.f * mx * ninth, dA2 = 2.f * (xv - mx) * ninth;
                        const float dB1 = 2.f * my * ninth, dB2 = 2.f * (yv - my) * ninth;
                        const float dn = dA1 * A2 + A1 * dA2, dd = dB1 * B2 + B1 * dB2;
                        g[tc] += 0.85f * (-0.5f) * (dn * d - n * dd) / (d * d);
                    }
                }
            }
        }
        a.gxy[2 * i] = g[0] * a.drgb[6 * i] + g[1] * a.drgb[6 * i + 1] + g[2] * a.drgb[6 * i + 2];
        a.gxy[2 * i + 1] = g[0] * a.drgb[6 * i + 3] + g[1] * a.drgb[6 * i + 4] + g[2] * a.drgb[6 * i + 5];
    }
    const float bs = block_sum(lsum, scratch);
    if (threadIdx.x == 0) a.part_fwd[4 * blockIdx.x + 0] = bs;
}

// keys for both directions in one launch
__global__ void aux_fill_keys_kernel(unsigned long long* keys, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) keys[i] = ~0ull;
}

// decode (distance bits << 32 | index) of this rank's source points and leave the block's distance sum in its partial slot
__global__ __launch_bounds__(256) void aux_decode_kernel(const unsigned long long* keys, int S, int s_lo, int s_hi, int64_t* idx, float* dist,
                                                         float* part, int slot) {
    __shared__ float scratch[4];
    const int s = blockIdx.x * 256 + threadIdx.x;
    float d = 0.f;
    if (s < S && s >= s_lo && s < s_hi) {
        const unsigned long long k = keys[s];
        idx[s] = (int64_t)(unsigned int)(k & 0xffffffffu);
        d = __uint_as_float((unsigned int)(k >> 32));
        dist[s] = d;
    }
    const float bs = block_sum(d, scratch);
    if (threadIdx.x == 0) part[4 * blockIdx.x + slot] = bs;
}

// ---- nearest neighbours between the two clouds: the search that knows where the destination points can be -------------------------------
// Both clouds are depth maps lifted along the rays of a pixel grid: Y_j = c + t_j M b_j with b_j = (x'_j, y'_j, 1) the grid point of index j,
// t_j = depth / scale2 and, for the direction X -> Y,  M = Kinv[:3,:3], c = Kinv[:3,3] / scale2;  for Y -> X,  M = R Kinv[:3,:3],
// c = (R Kinv[:3,3] + t) / scale2  (aux_geometry above).  A destination point lies on the LINE c + t M b_j whatever its depth, so a source
// P = c + w with w = p_z M a, a = (u, v, 1), is at least
//        dist(P, line_j) = |w x M b_j| / |M b_j| = |p_z| |cof(M) (a x b_j)| / |M b_j|  >=  |p_z| sqrt(q(a - b_j)) / B
// away from it, where a x b = (dv, -du, k), q(du, dv) is the quadratic form of cof(M) on the first two components with the cof(M) e_3
// direction projected out (whatever k is), and B = max |M b| over the grid (a corner).  Grid points outside the ellipse
// q(du, dv) <= (rho B / p_z)^2 cannot hold a point within rho of P: the brute-force search over S destinations (nnr_pointcloud.hip: 1.05 G
// pairs per direction at 135 x 240, 176 us at the VALU floor of its instruction mix) becomes a walk over the grid rows around P's own
// projection, the ellipse shrinking with the running minimum.  The candidates that ARE visited are evaluated exactly as the brute-force
// kernel evaluates them (torch.linalg.norm's fma chain, keys (sqrt bits, index): the first index among equal rounded distances), and the
// pruning only ever skips grid points whose line is farther than the running minimum plus margins for every rounding involved (the stored
// points are within a few ulp of their lines; u, v and q carry relative errors of 1e-6: rho is inflated by 1e-3 and by 1e-5 of the
// magnitudes involved, the pixel ranges by one pixel) -- the indices are those of the exhaustive search, tests/test_aux_terms.py and
// tests/test_pointcloud.py compare them.  A source behind / beside the destination camera (p_z ~ 0), a singular M or a degenerate grid
// disable the pruning for that source: it scans everything.
// G lanes share one source: rows round-robin, the running minimum shared after every row.  One launch, blockIdx.y = direction; idx / dist are
// written directly (no key table, no fill, no decode).
constexpr int kPcLanes = 8;      // lanes per source

struct PcRays {      // per direction, wave-uniform
    float c[3], Minv[9];
    float guu, guv, gvv, det, B;      // q(du, dv) = guu du^2 + 2 guv du dv + gvv dv^2;  det = guu gvv - guv^2
    bool ok;
};

__device__ __forceinline__ PcRays pc_rays(const AuxArgs& a, int dir) {
    PcRays r;
    const float s2 = (a.flags & NNR_AUX_SCALE_PCS) ? a.scale2[0] : 1.f;
    float M[9], k3[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        k3[i] = a.Kinv[4 * i + 3];
#pragma unroll
        for (int j = 0; j < 3; ++j) M[3 * i + j] = a.Kinv[4 * i + j];
    }
    if (dir == 0) {
#pragma unroll
        for (int i = 0; i < 3; ++i) r.c[i] = k3[i] / s2;
    } else {
        float RM[9];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            r.c[i] = (a.rel[4 * i] * k3[0] + a.rel[4 * i + 1] * k3[1] + a.rel[4 * i + 2] * k3[2] + a.rel[4 * i + 3]) / s2;
#pragma unroll
            for (int j = 0; j < 3; ++j) RM[3 * i + j] = a.rel[4 * i] * M[j] + a.rel[4 * i + 1] * M[3 + j] + a.rel[4 * i + 2] * M[6 + j];
        }
#pragma unroll
        for (int i = 0; i < 9; ++i) M[i] = RM[i];
    }
    // cof(M): row i = cross product of the other two rows (cyclic); M^-1 = cof(M)^T / det
    float N[9];
    N[0] = M[4] * M[8] - M[5] * M[7]; N[1] = M[5] * M[6] - M[3] * M[8]; N[2] = M[3] * M[7] - M[4] * M[6];
    N[3] = M[7] * M[2] - M[8] * M[1]; N[4] = M[8] * M[0] - M[6] * M[2]; N[5] = M[6] * M[1] - M[7] * M[0];
    N[6] = M[1] * M[5] - M[2] * M[4]; N[7] = M[2] * M[3] - M[0] * M[5]; N[8] = M[0] * M[4] - M[1] * M[3];
    const float dm = M[0] * N[0] + M[1] * N[1] + M[2] * N[2];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) r.Minv[3 * i + j] = N[3 * j + i] / dm;
    // columns of cof(M): n1 (times dv), n2 (times -du), n3 (times k: projected out)
    float n1[3] = {N[0], N[3], N[6]}, n2[3] = {N[1], N[4], N[7]};
    const float n3[3] = {N[2], N[5], N[8]};
    const float l3 = n3[0] * n3[0] + n3[1] * n3[1] + n3[2] * n3[2];
    const float p1 = (n1[0] * n3[0] + n1[1] * n3[1] + n1[2] * n3[2]) / l3, p2 = (n2[0] * n3[0] + n2[1] * n3[1] + n2[2] * n3[2]) / l3;
#pragma unroll
    for (int i = 0; i < 3; ++i) { n1[i] -= p1 * n3[i]; n2[i] -= p2 * n3[i]; }
    r.gvv = n1[0] * n1[0] + n1[1] * n1[1] + n1[2] * n1[2];
    r.guu = n2[0] * n2[0] + n2[1] * n2[1] + n2[2] * n2[2];
    r.guv = -(n1[0] * n2[0] + n1[1] * n2[1] + n1[2] * n2[2]);
    r.det = r.guu * r.gvv - r.guv * r.guv;
    float B2 = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {      // |M b|^2 is convex in b: its maximum over the grid is at a corner
        const float bx = (k & 1) ? 1.f : -1.f, by = (k & 2) ? 1.f : -1.f;
        float l = 0.f;
#pragma unroll
        for (int i = 0; i < 3; ++i) { const float t = M[3 * i] * bx + M[3 * i + 1] * by + M[3 * i + 2]; l += t * t; }
        B2 = fmaxf(B2, l);
    }
    r.B = sqrtf(B2) * 1.0001f;
    const float big = 1e30f;
    r.ok = a.wr > 1 && a.hr > 1 && fabsf(dm) > 1e-30f && fabsf(dm) < big && l3 > 1e-30f && r.det > 1e-12f * r.guu * r.gvv && r.guu > 0.f && r.gvv > 0.f
           && B2 < big && s2 == s2 && fabsf(s2) > 1e-30f;
    return r;
}

// Wave-wide minimum / maximum of a float through DPP row operations (six VALU instructions; the __shfl_xor butterfly goes through the LDS
// crossbar -- ds_bpermute, ~100 cycles a step -- and the best-first loop below does two such reductions per turn).  Every lane returns the result.
template <bool MAX>
__device__ __forceinline__ float wave_reduce(float v) {
    auto op = [](float a, float b) { return MAX ? fmaxf(a, b) : fminf(a, b); };
    auto dpp = [](float x, auto ctrl, auto row_mask) {
        return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(x), __float_as_int(x), decltype(ctrl)::value, decltype(row_mask)::value, 0xf, false));
    };
    using std::integral_constant;
    v = op(v, dpp(v, integral_constant<int, 0xB1>{}, integral_constant<int, 0xf>{}));       // quad_perm [1, 0, 3, 2]
    v = op(v, dpp(v, integral_constant<int, 0x4E>{}, integral_constant<int, 0xf>{}));       // quad_perm [2, 3, 0, 1]
    v = op(v, dpp(v, integral_constant<int, 0x141>{}, integral_constant<int, 0xf>{}));      // row_half_mirror
    v = op(v, dpp(v, integral_constant<int, 0x140>{}, integral_constant<int, 0xf>{}));      // row_mirror: every lane of a row of 16 holds the row's result
    v = op(v, dpp(v, integral_constant<int, 0x142>{}, integral_constant<int, 0xa>{}));      // row_bcast:15 into rows 1 and 3
    v = op(v, dpp(v, integral_constant<int, 0x143>{}, integral_constant<int, 0xc>{}));      // row_bcast:31 into rows 2 and 3: lane 63 holds the wave's result
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// the value of lane (l ^ 1), (l ^ 2), (7 - (l & 7)) of a group of eight (quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror): what a minimum over
// the eight lanes needs, without the LDS crossbar
template <int CTRL>
__device__ __forceinline__ int dpp_i32(int x) { return __builtin_amdgcn_update_dpp(x, x, CTRL, 0xf, 0xf, false); }

constexpr float kPcHeavyArea = 1000.f;      // grid points in a source's window after the first guesses beyond which it goes to the tile kernel

constexpr int kPcHeavyVote = 4;             // sources of an 8 x 8 tile with such a window for the tile to go there (fewer: they are scanned here, slowly)

// One workgroup = one 8 x 8 tile of the source grid (the tile kernel's tiles), eight lanes per source.
__global__ __launch_bounds__(64 * kPcLanes) void aux_pc_search_kernel(AuxArgs a, int* __restrict__ heavy, const float* __restrict__ spheres, float heavy_area, int heavy_vote) {      // heavy: null = scan everything here
    constexpr int G = kPcLanes;
    const int dir = blockIdx.y;
    const float* __restrict__ src = dir == 0 ? a.X : a.Y;
    const float* __restrict__ dst = dir == 0 ? a.Y : a.X;
    const int wr = a.wr, hr = a.hr;
    const int sub = threadIdx.x % G, grp = threadIdx.x / G;
    const int tiles_x = (wr + 7) / 8;
    const int src_y = 8 * ((int)blockIdx.x / tiles_x) + grp / 8, src_x = 8 * ((int)blockIdx.x % tiles_x) + grp % 8;
    const int s_flat = src_y * wr + src_x;
    const bool live = src_y < hr && src_x < wr && s_flat >= a.s_lo && s_flat < a.s_hi;
    const int s = live ? s_flat : a.s_lo;      // (a source that is not this workgroup's, or nobody's: it walks along to the vote below and leaves)
    if (a.s_hi <= a.s_lo) return;
    const PcRays R = pc_rays(a, dir);
    const float sx = src[3 * s], sy = src[3 * s + 1], sz = src[3 * s + 2];
    const float wx = sx - R.c[0], wy = sy - R.c[1], wz = sz - R.c[2];
    const float px = R.Minv[0] * wx + R.Minv[1] * wy + R.Minv[2] * wz, py = R.Minv[3] * wx + R.Minv[4] * wy + R.Minv[5] * wz,
                pz = R.Minv[6] * wx + R.Minv[7] * wy + R.Minv[8] * wz;
    const float wl = sqrtf(wx * wx + wy * wy + wz * wz), cl = fabsf(R.c[0]) + fabsf(R.c[1]) + fabsf(R.c[2]);
    // prune only where the projection means something: in front of / behind the camera by more than a sliver of |p|
    const bool prune = R.ok && fabsf(pz) > 1e-4f * (fabsf(px) + fabsf(py)) && fabsf(pz) > 1e-30f && wl < 1e30f;
    const float u = prune ? px / pz : 0.f, v = prune ? py / pz : 0.f;
    const float hw = 0.5f * (float)(wr - 1), hh = 0.5f * (float)(hr - 1);
    const float fx = (u + 1.f) * hw, fy = (v + 1.f) * hh;      // P's projection in grid units
    const int x0 = (int)fminf(fmaxf(rintf(fx), 0.f), (float)(wr - 1)), y0 = (int)fminf(fmaxf(rintf(fy), 0.f), (float)(hr - 1));
    const float bz = prune ? R.B / fabsf(pz) : 0.f;

    float best_s = __builtin_inff(), thr = __builtin_inff();
    int best_i = 0x7fffffff;
    auto consider = [&](float d2, int idx) __attribute__((always_inline)) {
        if (d2 <= thr && d2 < __builtin_inff()) {
            const float sq = __fsqrt_rn(d2);
            if (sq < best_s || (sq == best_s && idx < best_i)) {
                best_s = sq;
                best_i = idx;
                const float t = sq * sq;
                thr = __builtin_fmaf(t, 4.8e-7f, t) + 1.2e-38f;      // every d2 whose rounded root is <= best_s lies below this (nnr_pointcloud.hip)
            }
        }
    };
    auto dist2 = [&](int j) __attribute__((always_inline)) {
        const float ex = sx - dst[3 * j], ey = sy - dst[3 * j + 1], ez = sz - dst[3 * j + 2];
        return __builtin_fmaf(ez, ez, __builtin_fmaf(ey, ey, ex * ex));      // torch.linalg.norm's sum of squares
    };
    static_assert(G == 8, "the exchange pattern below is for groups of eight lanes");
    auto share = [&]() __attribute__((always_inline)) {      // the group's running minimum in every lane of the group
        auto take = [&](float os, int oi) { if (os < best_s || (os == best_s && oi < best_i)) { best_s = os; best_i = oi; } };
        take(__int_as_float(dpp_i32<0xB1>(__float_as_int(best_s))), dpp_i32<0xB1>(best_i));        // lane ^ 1
        take(__int_as_float(dpp_i32<0x4E>(__float_as_int(best_s))), dpp_i32<0x4E>(best_i));        // lane ^ 2: every lane holds its quad's minimum
        take(__int_as_float(dpp_i32<0x141>(__float_as_int(best_s))), dpp_i32<0x141>(best_i));      // the mirror lane of the other quad: the group's
        const float t = best_s * best_s;
        thr = best_s < __builtin_inff() ? __builtin_fmaf(t, 4.8e-7f, t) + 1.2e-38f : __builtin_inff();
    };
    // first guesses: the 9 x 9 grid points around the projection (clamped into the grid) -- enough of them that the window they imply says
    // which kernel this source belongs to (below), also where the depths are rough
    for (int k = sub; live && k < 81; k += G) {
        const int yy = min(max(y0 + k / 9 - 4, 0), hr - 1), xx = min(max(x0 + k % 9 - 4, 0), wr - 1);
        consider(dist2(yy * wr + xx), yy * wr + xx);
    }
    share();
    // A window that is still large after the first guesses (or a source that cannot prune at all) is not for this kernel -- a candidate
    // from memory costs ~20x what it costs the tile kernel below.  The tile's sources vote: with kPcHeavyVote or more such windows the tile
    // is flagged HEAVY and everybody leaves -- aux_pc_search_tile_kernel redoes every source of a flagged tile (and exits at once for the
    // others); with fewer (rough depths: the odd bad first guess) they are scanned here.
    {
        bool mine = false;
        if (heavy && live) {
            float area = __builtin_inff();
            if (prune && thr < __builtin_inff()) {
                const float rho = sqrtf(thr) * 1.001f + 1e-5f * (wl + cl + sqrtf(thr)) + 1e-30f;
                const float t = rho * bz, r2 = t * t * 1.001f;
                area = 4.f * sqrtf(r2 * R.guu / R.det) * hh * sqrtf(r2 * R.gvv / R.det) * hw;      // the ellipse's bounding box in grid points
            }
            // ... and the tile kernel can do better only where its sphere bound bites: the destination tile under the projection must be
            // THIN against the distance in question (a smooth surface patch; with rough depths -- a sphere as deep as the depth range -- the
            // ray window is all there is, and that is this kernel's bound)
            const float r_tile = spheres[4 * ((dir == 0 ? tiles_x * ((hr + 7) / 8) : 0) + (y0 / 8) * tiles_x + x0 / 8) + 3];
            mine = !(area <= heavy_area) && !(r_tile > 3.f * best_s);      // (NaN: heavy)
        }
        const int votes = __syncthreads_count(mine && sub == 0);      // (every thread of the workgroup is here)
        if (votes >= heavy_vote) {
            if (threadIdx.x == 0) heavy[dir * (tiles_x * ((hr + 7) / 8)) + (int)blockIdx.x] = 1;
            return;
        }
        if (!live) return;      // (whole groups: the shuffles below stay inside a group)
    }
    // rows outwards from the projection: offsets 0, +1, -1, +2, -2, ..; lane `sub` takes every G-th
    const int m_end = 2 * max(y0, hr - 1 - y0);      // last useful position of the zigzag
    const float pv = 1.f / hh;                       // grid pitch in v
    for (int m0 = 0; m0 <= m_end; m0 += G) {
        // the ellipse for the group's minimum (the same in every lane of the group: a group-uniform trip count)
        float r2 = __builtin_inff();
        if (prune && thr < __builtin_inff()) {
            const float rho = sqrtf(thr) * 1.001f + 1e-5f * (wl + cl + sqrtf(thr)) + 1e-30f;
            const float t = rho * bz;
            r2 = t * t * 1.001f;
        }
        const float dv_max = r2 < __builtin_inff() ? sqrtf(r2 * R.guu / R.det) : __builtin_inff();
        // the nearest row of this trip is (m0 + 1) / 2 - 1 .. rows away from y0; y0 itself is within half a row (+ clamping) of fy
        const float near_rows = (float)((m0 + 1) / 2);
        const float gap = fabsf((float)y0 - fy);      // > 0.5 only where the projection lies outside the grid: then every row is farther
        if ((near_rows - 0.5f) * pv > dv_max + pv && (near_rows + gap - 1.f) * pv > dv_max + pv) break;
        const int m = m0 + sub;
        const int y = y0 + ((m & 1) ? (m + 1) / 2 : -(m / 2));
        if (m <= m_end && y >= 0 && y < hr) {
            int xlo = 0, xhi = wr - 1;
            bool any = true;
            if (r2 < __builtin_inff()) {
                const float dv = v - ((float)y * pv - 1.f);
                const float D = R.guu * r2 - R.det * dv * dv;
                if (D < 0.f) any = false;
                else {
                    const float du_c = -R.guv * dv / R.guu, du_r = sqrtf(D) / R.guu;
                    // du = u - u'  ->  u' in [u - du_c - du_r, u - du_c + du_r]
                    const float lo = (u - du_c - du_r + 1.f) * hw - 1.5f, hi = (u - du_c + du_r + 1.f) * hw + 1.5f;
                    if (!(lo <= (float)(wr - 1)) || !(hi >= 0.f)) any = lo != lo || hi != hi;      // outside the grid (NaN: keep the whole row)
                    else {
                        xlo = (int)fmaxf(floorf(lo), 0.f);
                        xhi = (int)fminf(ceilf(hi), (float)(wr - 1));
                    }
                }
            }
            if (any) {
                const int row = y * wr;
                for (int x = xlo; x <= xhi; x += 4) {
                    float d2[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) d2[k] = dist2(row + min(x + k, xhi));
                    if ((d2[0] <= thr) | (d2[1] <= thr) | (d2[2] <= thr) | (d2[3] <= thr)) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) consider(d2[k], row + min(x + k, xhi));      // (a repeated last point changes nothing)
                    }
                }
            }
        }
        share();
    }
    if (sub == 0) {
        int64_t* idx = dir == 0 ? a.idx_xy : a.idx_yx;
        float* dist = dir == 0 ? a.dist_xy : a.dist_yx;
        // nothing found (NaN / inf coordinates): index 0xffffffff, a NaN distance -- what the key table decoded to
        idx[s] = best_i == 0x7fffffff ? (int64_t)0xffffffffll : (int64_t)best_i;
        dist[s] = best_i == 0x7fffffff ? __uint_as_float(0xffffffffu) : best_s;
    }
}

// ---- the second kernel of the search: the HEAVY tiles -- smooth surfaces far apart ------------------------------------------------------
// aux_pc_search_kernel above reads every candidate from global memory, eight lanes per source, and prunes by the distance to the
// destination RAYS only.  That is the right bound for depth maps that are rough along the ray (the bench's white-noise depths: 7 pixels of
// radius, ~50 us), but it ignores DEPTH: where the two clouds are smooth surfaces a parallax or a not-yet-learned depth scale apart -- a real
// scene at the start of training: nearest neighbours 0.1 - 1 scene units away, 30 - 70 pixels of radius (tools/pc_window_stats.py) -- the
// ellipses hold a third of the grid, a candidate costs ~20x what it costs the exhaustive kernel, and the search took 420 us inside the
// reference's train.py, more than the exhaustive one (profiles/r05/l_*).  So the sources are split.  aux_pc_spheres_kernel gives every
// 8 x 8 tile of either cloud's grid a bounding sphere (smooth patches have small ones, whatever their pose).  In aux_pc_search_kernel the 64
// sources of a tile VOTE after their first guesses: a source whose window is still large, over a destination patch that is thin against
// the distance in question, is better off here; with kPcHeavyVote of them the tile is flagged and left to aux_pc_search_tile_kernel:
//   * sixteen waves per flagged 8 x 8 tile of sources, lane = source in each; destination tile (ty, tx) belongs to wave 4 (ty & 3) + (tx & 3)
//     (what a tile costs is set by its slowest sources -- a region the other frame does not see has windows of hundreds of tiles -- so the
//     tile's work is spread wide: 4 waves 190 us in the training loop, 16 waves 164);
//   * BEST FIRST: every destination tile of the wave has a wave-level bound |centre - centre| - both radii (no source of the tile is closer
//     to any of its points), one tile per lane and register; each turn takes the smallest bound left, stops when it exceeds the largest
//     running minimum of any source, re-tests per source (the ray window, and the tile's sphere against the source's own point and
//     minimum) and, if any source needs it, evaluates the tile for ALL 64 lanes the way the exhaustive kernel evaluates candidates: the
//     64 points go through a wave-private LDS image as pairs [x0 x1 y0 y1 z0 z1], three 16-byte broadcasts deliver four points as packed
//     operands -- the same arithmetic, keys and tie rule, ~5 instructions per pair instead of ~100;
//   * the waves share their running minima through an atomic minimum in LDS (read without synchronisation: a stale value is a larger
//     bound, never a wrong one) and merge their results at the end.
// Every bound only ever skips points that are farther than the source's running minimum (margins for every rounding involved); the indices
// are the exhaustive search's, in either kernel and in any split (tests/test_gpu_pc_search.py runs each kernel alone on everything too).
// Measured (tools/time_pc_search.py --scene, tools/gpu_loop_trace.sh; exhaustive search / first kernel alone / both): bench-like noise depths
// 339 / 55 / 61 us, smooth depths a small pose apart 307 / 47 / 55, two frames of a real scene at the identity pose 315 / 1127 / 279, the
// search inside the reference's train.py over the first epochs 351 / 464 / 200.
constexpr int kPcT = 8;      // tile side (64 points: one per lane)

// spheres[cloud][tile] = (cx, cy, cz, r); cloud 0 = X, 1 = Y.  One wave per tile.
__global__ __launch_bounds__(64) void aux_pc_spheres_kernel(AuxArgs a, float* __restrict__ spheres) {
    const int tiles_x = (a.wr + kPcT - 1) / kPcT, tiles = tiles_x * ((a.hr + kPcT - 1) / kPcT);
    const float* __restrict__ cloud = blockIdx.y == 0 ? a.X : a.Y;
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int y = kPcT * ty + (int)threadIdx.x / kPcT, x = kPcT * tx + (int)threadIdx.x % kPcT;
    const bool in = y < a.hr && x < a.wr;
    const int j = in ? y * a.wr + x : 0;
    const float inf = __builtin_inff();
    const float qx = cloud[3 * j], qy = cloud[3 * j + 1], qz = cloud[3 * j + 2];
    float lo[3] = {in ? qx : inf, in ? qy : inf, in ? qz : inf}, hi[3] = {in ? qx : -inf, in ? qy : -inf, in ? qz : -inf};
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { lo[c] = fminf(lo[c], __shfl_xor(lo[c], o, 64)); hi[c] = fmaxf(hi[c], __shfl_xor(hi[c], o, 64)); }
    const float cx = 0.5f * (lo[0] + hi[0]), cy = 0.5f * (lo[1] + hi[1]), cz = 0.5f * (lo[2] + hi[2]);
    const float ex = qx - cx, ey = qy - cy, ez = qz - cz;
    float r = in ? sqrtf(ex * ex + ey * ey + ez * ez) : 0.f;
    bool bad = in && !(r < inf);      // a NaN / inf point: the tile gets no bound (it is always visited)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) r = fmaxf(r, __shfl_xor(r, o, 64));
    bad = __any(bad) || !(cx == cx && cy == cy && cz == cz) || !(fabsf(cx) < inf && fabsf(cy) < inf && fabsf(cz) < inf);
    if (threadIdx.x == 0) {
        float* o = spheres + 4 * ((int)blockIdx.y * tiles + (int)blockIdx.x);
        o[0] = bad ? 0.f : cx; o[1] = bad ? 0.f : cy; o[2] = bad ? 0.f : cz;
        o[3] = bad ? inf : r * 1.00001f + 1e-30f;
    }
}

constexpr int kPcMaxTiles = 2048;      // destination tiles the tile kernel handles (their spheres in LDS: 32 KB; one bound per lane and register: 32): 131 072 grid points;
                                       // larger grids stay in the eight-lanes-per-source kernel

constexpr int kPcTileWaves = 16;      // waves per source tile (a 4 x 4 pattern of destination tiles: one each)

__global__ __launch_bounds__(64 * kPcTileWaves) void aux_pc_search_tile_kernel(AuxArgs a, const float* __restrict__ spheres, const int* __restrict__ heavy) {
    __shared__ f32x4 img_all[kPcTileWaves][48];      // per wave: 64 points as 16 groups of [x0 x1 y0 y1 | z0 z1 x2 x3 | y2 y3 z2 z3]
    __shared__ f32x4 sph_lds[kPcMaxTiles];
    __shared__ float res_s[kPcTileWaves][64];
    __shared__ int res_i[kPcTileWaves][64];
    __shared__ int cap_min[64];      // per source: the smallest running minimum of any wave (bit pattern of a non-negative float: ordered like it), what all waves prune with
    const int dir = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float* __restrict__ src = dir == 0 ? a.X : a.Y;
    const float* __restrict__ dst = dir == 0 ? a.Y : a.X;
    const int wr = a.wr, hr = a.hr;
    const int tiles_x = (wr + kPcT - 1) / kPcT, tiles_y = (hr + kPcT - 1) / kPcT, tiles = tiles_x * tiles_y;
    if (heavy && !heavy[dir * tiles + (int)blockIdx.x]) return;      // (the whole workgroup: the eight-lanes-per-source kernel has done this tile)
    const f32x4* __restrict__ sph_g = reinterpret_cast<const f32x4*>(spheres) + (dir == 0 ? tiles : 0);      // the DESTINATION cloud's spheres
    if (tiles > kPcMaxTiles) __builtin_trap();      // (the launcher does not send such grids here)
    constexpr bool sph_in_lds = true;
    if (sph_in_lds)
        for (int t = threadIdx.x; t < tiles; t += 64 * kPcTileWaves) sph_lds[t] = sph_g[t];
    if (wave == 0) cap_min[lane] = __float_as_int(__builtin_inff());
    __syncthreads();
    // The waves of the workgroup own the SAME tile of sources (lane = source in each of them) and share the destination tiles: tile
    // (ty, tx) belongs to wave 4 (ty & 3) + (tx & 3).  A wave walks, tests and evaluates only its own tiles, prunes with the smallest running
    // minimum of any wave (an atomic minimum in LDS, read without synchronisation: a stale value is a larger bound, never a wrong one), and
    // the results are merged at the end.  The work of a tile is what it is; what this buys is the TAIL: the sources of a region that the
    // other frame does not see have windows of hundreds of tiles, and one wave walking them alone was the kernel's duration.
    const int st = blockIdx.x;
    const int sty = st / tiles_x, stx = st - sty * tiles_x;
    const int sy = kPcT * sty + lane / kPcT, sx = kPcT * stx + lane % kPcT;
    const int s = sy * wr + sx;
    const bool live = sy < hr && sx < wr && s >= a.s_lo && s < a.s_hi;
    if (!__any(live)) return;      // (the same lanes in all four waves: the whole workgroup leaves)
    const int sc = live ? s : 0;
    const PcRays R = pc_rays(a, dir);
    const float sxx = src[3 * sc], syy = src[3 * sc + 1], szz = src[3 * sc + 2];
    const float wx = sxx - R.c[0], wy = syy - R.c[1], wz = szz - R.c[2];
    const float px = R.Minv[0] * wx + R.Minv[1] * wy + R.Minv[2] * wz, py = R.Minv[3] * wx + R.Minv[4] * wy + R.Minv[5] * wz,
                pz = R.Minv[6] * wx + R.Minv[7] * wy + R.Minv[8] * wz;
    const float wl = sqrtf(wx * wx + wy * wy + wz * wz), cl = fabsf(R.c[0]) + fabsf(R.c[1]) + fabsf(R.c[2]);
    const bool prune = R.ok && fabsf(pz) > 1e-4f * (fabsf(px) + fabsf(py)) && fabsf(pz) > 1e-30f && wl < 1e30f;
    const float u = prune ? px / pz : 0.f, v = prune ? py / pz : 0.f;
    const float hw = 0.5f * (float)(wr - 1), hh = 0.5f * (float)(hr - 1);
    const float fx = (u + 1.f) * hw, fy = (v + 1.f) * hh;
    const float bz = prune ? R.B / fabsf(pz) : 0.f;
    const f32x2 X2 = {sxx, sxx}, Y2 = {syy, syy}, Z2 = {szz, szz};

    float best_s = __builtin_inff(), thr = __builtin_inff();
    int best_i = 0x7fffffff;
    auto consider = [&](float d2, int idx) __attribute__((always_inline)) {
        if (d2 <= thr && d2 < __builtin_inff()) {
            const float sq = __fsqrt_rn(d2);
            if (sq < best_s || (sq == best_s && idx < best_i)) {
                best_s = sq;
                best_i = idx;
                const float t = sq * sq;
                thr = __builtin_fmaf(t, 4.8e-7f, t) + 1.2e-38f;      // every d2 whose rounded root is <= best_s lies below this (nnr_pointcloud.hip)
            }
        }
    };
    // the lane's window: the bounding box of its ray ellipse for the current minimum, in TILE coordinates (the whole grid where it cannot prune)
    int bx0, bx1, by0, by1;
    float cap = __builtin_inff();      // min over the four waves of the running minimum of this source
    auto window = [&]() __attribute__((always_inline)) {
        bx0 = 0; bx1 = tiles_x - 1; by0 = 0; by1 = tiles_y - 1;
        cap = __int_as_float(cap_min[lane]);
        if (prune && cap < __builtin_inff()) {
            const float t2 = cap * cap, thr_c = __builtin_fmaf(t2, 4.8e-7f, t2) + 1.2e-38f;
            const float rho = sqrtf(thr_c) * 1.001f + 1e-5f * (wl + cl + sqrtf(thr_c)) + 1e-30f;
            const float t = rho * bz, r2 = t * t * 1.001f;
            const float dvm = sqrtf(r2 * R.guu / R.det) * hh + 1.5f, dum = sqrtf(r2 * R.gvv / R.det) * hw + 1.5f;      // half extents in grid units
            const float ya = fy - dvm, yb = fy + dvm, xa = fx - dum, xb = fx + dum;
            if (ya == ya && yb == yb && xa == xa && xb == xb) {      // (NaN: keep the whole grid)
                by0 = (int)fminf(fmaxf(floorf(ya), 0.f), (float)hr) / kPcT; by1 = (int)fmaxf(fminf(ceilf(yb), (float)(hr - 1)), -1.f);
                bx0 = (int)fminf(fmaxf(floorf(xa), 0.f), (float)wr) / kPcT; bx1 = (int)fmaxf(fminf(ceilf(xb), (float)(wr - 1)), -1.f);
                by1 = by1 < 0 ? -1 : by1 / kPcT;      // (an ellipse that misses the grid: an empty window)
                bx1 = bx1 < 0 ? -1 : bx1 / kPcT;
            }
        }
        if (!live) { by0 = 1; by1 = 0; bx0 = 1; bx1 = 0; }
    };
    f32x4* const img = img_all[wave];
    float* const imgf = reinterpret_cast<float*>(img);
    const int my_slot = 12 * (lane >> 2) + ((lane & 3) >> 1) * 6 + (lane & 1);      // point l of the tile: group l / 4, pair (l & 3) / 2, half l & 1: x at +0, y at +2, z at +4
    // does any lane need tile (ty, tx)?  (its window meets the tile and the tile's sphere comes within its running minimum)
    auto needed = [&](int ty, int tx) __attribute__((always_inline)) {
        const f32x4 sp = sph_in_lds ? sph_lds[ty * tiles_x + tx] : sph_g[ty * tiles_x + tx];
        const float ex = sxx - sp[0], ey = syy - sp[1], ez = szz - sp[2];
        const float lb = sqrtf(ex * ex + ey * ey + ez * ez) * 0.999998f - sp[3];      // no point of the tile is closer than this
        return __any(ty >= by0 && ty <= by1 && tx >= bx0 && tx <= bx1 && !(lb > fminf(best_s, cap) * 1.000001f)) != 0;
    };
    // Best first.  Every destination tile of this wave gets a WAVE-level bound -- |centre of the source tile's sphere - centre of its sphere|
    // minus both radii: no source of the tile is closer to any of its points -- held one tile per lane and register (tile 64 k + lane in
    // register k).  Each turn takes the smallest bound left (a wave minimum), stops when it exceeds the largest running minimum of any
    // source, re-tests the tile per source (needed(): the ray window and the sphere against the source's own point and minimum) and
    // evaluates it.  No walk over the grid, and the near tiles -- the ones that set the minima -- come first whatever the geometry.
    constexpr int kRegs = kPcMaxTiles / 64;
    float lbk[kRegs];
    {
        const f32x4 ss = (reinterpret_cast<const f32x4*>(spheres) + (dir == 0 ? 0 : tiles))[st];      // the SOURCE tile's sphere
#pragma unroll
        for (int k = 0; k < kRegs; ++k) {
            const int t = 64 * k + lane;
            lbk[k] = __builtin_inff();
            if (t < tiles && sph_in_lds) {
                const int ty = t / tiles_x, tx = t - ty * tiles_x;
                if ((4 * (ty & 3) + (tx & 3)) == wave) {
                    const f32x4 sd = sph_lds[t];
                    const float ex = ss[0] - sd[0], ey = ss[1] - sd[1], ez = ss[2] - sd[2];
                    const float lb = sqrtf(ex * ex + ey * ey + ez * ez) * 0.999998f - ss[3] - sd[3];
                    lbk[k] = lb == lb ? fmaxf(lb, -3e38f) : -3e38f;      // (no bound: first in line)
                }
            }
        }
    }
    const float inf = __builtin_inff();
    auto fetch = [&](int ty, int tx, float (&q)[3]) __attribute__((always_inline)) {      // this lane's point of the tile (padding: +inf, never a minimum)
        const int y = kPcT * ty + lane / kPcT, x = kPcT * tx + lane % kPcT;
        const bool in = y < hr && x < wr;
        const int j = in ? y * wr + x : 0;
        q[0] = in ? dst[3 * j] : inf; q[1] = in ? dst[3 * j + 1] : inf; q[2] = in ? dst[3 * j + 2] : inf;
    };
    const int n_regs = (tiles + 63) / 64;      // (wave-uniform: registers beyond it hold +inf and are not looked at)
    float rho_max = inf;
    bool stale = true;      // the windows / the largest minimum need recomputing (something was evaluated since)
    for (int turn = 0; turn < kPcMaxTiles; ++turn) {
        // the smallest bound left: per lane, then over the wave
        float m = lbk[0];
        int km = 0;
#pragma unroll
        for (int k = 1; k < kRegs; ++k)
            if (k < n_regs && lbk[k] < m) { m = lbk[k]; km = k; }
        const float wm = wave_reduce<false>(m);
        if (!(wm < inf)) break;      // nothing left
        if (stale || (turn & 7) == 7) {      // (every few turns anyway: the other waves lower the shared minima too)
            window();                        // cap and the windows for the minima as they stand
            rho_max = wave_reduce<true>(live ? fminf(best_s, cap) : 0.f);
            stale = false;
        }
        if (wm > rho_max * 1.000001f) break;      // no tile left can hold a point within any source's minimum
        const unsigned long long who = __ballot(m == wm);
        const int owner = __builtin_ctzll(who);
        const int t = __builtin_amdgcn_readlane(64 * km + lane, owner);
        if (lane == owner) {
#pragma unroll
            for (int k = 0; k < kRegs; ++k)
                if (k == km) lbk[k] = inf;
        }
        const int ty = t / tiles_x, tx = t - ty * tiles_x;
        if (!needed(ty, tx)) continue;
        float q[3];
        fetch(ty, tx, q);
        __builtin_amdgcn_wave_barrier();      // (the previous tile's reads are done: one wave, LDS operations in order)
        imgf[my_slot] = q[0]; imgf[my_slot + 2] = q[1]; imgf[my_slot + 4] = q[2];
        __builtin_amdgcn_wave_barrier();
#pragma unroll 4
        for (int g = 0; g < 16; ++g) {
            const f32x4 A = img[3 * g], Bq = img[3 * g + 1], C = img[3 * g + 2];      // x0 x1 y0 y1 | z0 z1 x2 x3 | y2 y3 z2 z3
            const f32x2 dx0 = X2 - f32x2{A[0], A[1]}, dy0 = Y2 - f32x2{A[2], A[3]}, dz0 = Z2 - f32x2{Bq[0], Bq[1]};
            const f32x2 dx1 = X2 - f32x2{Bq[2], Bq[3]}, dy1 = Y2 - f32x2{C[0], C[1]}, dz1 = Z2 - f32x2{C[2], C[3]};
            const f32x2 e0 = __builtin_elementwise_fma(dz0, dz0, __builtin_elementwise_fma(dy0, dy0, dx0 * dx0));
            const f32x2 e1 = __builtin_elementwise_fma(dz1, dz1, __builtin_elementwise_fma(dy1, dy1, dx1 * dx1));
            if ((e0[0] <= thr) | (e0[1] <= thr) | (e1[0] <= thr) | (e1[1] <= thr)) {
                const int l0 = 4 * g, j0 = (kPcT * ty + l0 / kPcT) * wr + kPcT * tx + l0 % kPcT;      // (4 consecutive points share a tile row)
                consider(e0[0], j0); consider(e0[1], j0 + 1); consider(e1[0], j0 + 2); consider(e1[1], j0 + 3);
            }
        }
        if (best_s < cap) atomicMin(&cap_min[lane], __float_as_int(best_s));      // (for all waves, this one's next window() included)
        stale = true;
    }
    res_s[wave][lane] = best_s;
    res_i[wave][lane] = best_i;
    __syncthreads();
    if (wave == 0 && live) {      // the four waves' minima of this source: (distance, index) lexicographic, as everywhere
#pragma unroll
        for (int w = 1; w < kPcTileWaves; ++w) {
            const float os = res_s[w][lane];
            const int oi = res_i[w][lane];
            if (os < best_s || (os == best_s && oi < best_i)) { best_s = os; best_i = oi; }
        }
        int64_t* idx = dir == 0 ? a.idx_xy : a.idx_yx;
        float* dist = dir == 0 ? a.dist_xy : a.dist_yx;
        idx[s] = best_i == 0x7fffffff ? (int64_t)0xffffffffll : (int64_t)best_i;
        dist[s] = best_i == 0x7fffffff ? __uint_as_float(0xffffffffu) : best_s;
    }
}

// column `k` of an [nb][stride] table of per-block partials, summed in a fixed order by one wave: lane l adds blocks l, l + 64, ...,
// then the lane tree.  Every lane returns the total.
__device__ __forceinline__ float ordered_column_sum(const float* part, int nb, int stride, int k) {
    float v = 0.f;
    for (int b = threadIdx.x; b < nb; b += 64) v += part[stride * b + k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// out = [loss_pc, loss_rgb_s, n_valid, weighted]; sums over this rank's shard of the source points, normalisers global.  One wave.
// weighted (NNR_AUX_WEIGHTED) = w_pc loss_pc + w_rgbs loss_rgb_s over the active terms, each product and the sum rounded to fp32 as the
// torch expression of model/losses.py rounds them (w * loss, then +): two multiplies and an add that were three launches.
__device__ __forceinline__ void aux_finish_body(const AuxArgs& a) {
    const int nb = (a.S + 255) / 256;
    const bool rgbs = (a.flags & NNR_AUX_RGBS) != 0, pc = (a.flags & NNR_AUX_PC) != 0;
    const float lsum = rgbs ? ordered_column_sum(a.part_fwd, nb, 4, 0) : 0.f;
    const float lcnt = rgbs ? ordered_column_sum(a.part_fwd, nb, 4, 1) : 0.f;
    const float dxy = pc ? ordered_column_sum(a.part_fwd, nb, 4, 2) : 0.f;
    const float dyx = pc ? ordered_column_sum(a.part_fwd, nb, 4, 3) : 0.f;
    if (threadIdx.x != 0) return;
    a.acc[0] = lsum; a.acc[1] = lcnt; a.acc[2] = dxy; a.acc[3] = dyx;
    const float l_pc = pc ? (dxy + dyx) / (float)a.S : 0.f;
    const float l_rgbs = rgbs && lcnt > 0.f ? lsum / (3.f * lcnt) : 0.f;
    a.out[0] = l_pc;
    a.out[1] = l_rgbs;
    a.out[2] = lcnt;
    float wsum = 0.f;
    if (a.flags & NNR_AUX_WEIGHTED) {
        const float t_pc = __fmul_rn(a.w_pc, l_pc), t_rgbs = __fmul_rn(a.w_rgbs, l_rgbs);
        wsum = pc && rgbs ? __fadd_rn(t_pc, t_rgbs) : (pc ? t_pc : t_rgbs);
    }
    a.out[3] = wsum;
}

__global__ __launch_bounds__(64) void aux_finish_kernel(AuxArgs a) { aux_finish_body(a); }

// the block's distance sums of both directions into their partial slots (blockIdx.y = direction); the block that finishes LAST (a ticket
// counter that aux_points_fwd_kernel zeroed: acc[7]) also closes the forward -- the one-wave finishing kernel above without its launch
__global__ __launch_bounds__(256) void aux_dist_sum_kernel(AuxArgs a) {
    __shared__ float scratch[4];
    __shared__ int last;
    const int s = blockIdx.x * 256 + threadIdx.x;
    const float* dist = blockIdx.y == 0 ? a.dist_xy : a.dist_yx;
    const float d = (s < a.S && s >= a.s_lo && s < a.s_hi) ? dist[s] : 0.f;
    const float bs = block_sum(d, scratch);
    if (threadIdx.x == 0) {
        a.part_fwd[4 * blockIdx.x + 2 + blockIdx.y] = bs;
        __threadfence();      // the partial before the ticket
        const unsigned int t = atomicAdd(reinterpret_cast<unsigned int*>(a.acc + 7), 1u);
        last = t == gridDim.x * gridDim.y - 1u;
    }
    __syncthreads();
    if (!last || threadIdx.x >= 64) return;
    __threadfence();          // every block's partials (and the re-projection partials of the kernels before) are visible now
    aux_finish_body(a);
}

// Gradients of the two clouds are accumulated as 64-bit FIXED-POINT numbers (units of 2^-44): several sources may share a
// destination, and integer atomics commute exactly where float atomics make the sum depend on arrival order.  Every term is
// bounded by |g_out[0]| / S and a destination collects at most S of them, so the range (+-2^19) is never approached; the
// resolution is 6e-14, nine orders of magnitude below a term.
constexpr double kFixScale = 17592186044416.0;   // 2^44
// A term that does not fit -- NaN, inf, or so large that S of them could leave the valid range |sum| < 2^18 (a loss-scaled upstream
// gradient could do that) -- must not wrap into a finite wrong number: it POISONS the accumulator (atomicMin to the most negative
// value).  Valid sums stay inside |q| < 2^62; a poisoned accumulator stays outside that range whatever in-range terms are added before
// or after (at most 2^62 in total, wrapping included), fix_get returns NaN for it, and the trainer's NaN check sees the step.
constexpr long long kFixPoison = (long long)0x8000000000000000ull;
constexpr long long kFixValid = 1ll << 62;
__device__ __forceinline__ void fix_add(long long* p, float v, float term_limit /* 2^18 / S */) {
    if (!(fabsf(v) <= term_limit)) {        // also true for NaN
        atomicMin(p, kFixPoison);
        return;
    }
    atomicAdd(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double2ll_rn((double)v * kFixScale));
}
__device__ __forceinline__ float fix_get(const long long* p) {
    const long long q = *p;
    if (q <= -kFixValid || q >= kFixValid) return __builtin_nanf("");
    return (float)((double)q * (1.0 / kFixScale));
}

// the upstream gradients of the two losses: g_out[0], g_out[1] -- or, NNR_AUX_WEIGHTED, the ONE gradient of out[3] = w_pc loss_pc + w_rgbs
// loss_rgb_s times the weights (what autograd's mul backward would have launched two kernels for: g * w, the same product)
__device__ __forceinline__ float aux_g_pc(const AuxArgs& a) { return (a.flags & NNR_AUX_WEIGHTED) ? a.g_out[0] * a.w_pc : a.g_out[0]; }
__device__ __forceinline__ float aux_g_rgbs(const AuxArgs& a) { return (a.flags & NNR_AUX_WEIGHTED) ? a.g_out[0] * a.w_rgbs : a.g_out[1]; }

// d mean_s dist[s] * coef for the sources [s_lo, s_hi) of this rank, ACCUMULATED: g_src[s] += w (src_s - dst_j), g_dst[j] -= the same;
// blockIdx.y = direction (X -> Y, Y -> X)
__global__ void aux_pc_bwd_kernel(AuxArgs a) {
    const bool xy = blockIdx.y == 0;
    const float* src = xy ? a.X : a.Y;
    const float* dst = xy ? a.Y : a.X;
    const int64_t* idx = xy ? a.idx_xy : a.idx_yx;
    const float* dist = xy ? a.dist_xy : a.dist_yx;
    long long* g_src = xy ? a.gXq : a.gYq;
    long long* g_dst = xy ? a.gYq : a.gXq;
    const int S = a.S;
    const int s = a.s_lo + blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= a.s_hi) return;
    const int64_t j = idx[s];
    if (j < 0 || j >= S) return;   // no finite distance was found (NaN / inf coordinates): no match, no gradient -- and no wild address
    const float dd = dist[s];
    const float w = dd > 0.f ? aux_g_pc(a) / ((float)S * dd) : 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float v = w * (src[3 * s + c] - dst[3 * j + c]);
        fix_add(g_src + 3 * s + c, v, 262144.f / (float)S);
        fix_add(g_dst + 3 * j + c, -v, 262144.f / (float)S);
    }
}

// d loss / d (d1, d2) of point i, the chain of gX, gY (point-cloud loss) and of the saved d(point loss)/d(xy) (re-projection loss)
// back to the two depth values; with ACC also this point's contribution to dL/d rel[r][0..3] (r = 0..2), dL/d scale2 in acc[12] and --
// NNR_AUX_GRAD_K, a learnable focal length -- dL/d K[r][0..3] in acc[13..25) and dL/d Kinv[r][0..3] in acc[25..37)
constexpr int kAuxCols = 41, kAuxStride = 44;      // [37, 41): dL/d (scale1, shift1, scale2, shift2) of NNR_AUX_AFFINE
template <bool ACC>
__device__ __forceinline__ void aux_point_grads(const AuxArgs& a, int i, float& gd1, float& gd2, uint32_t& fl, int& src1,
                                                float (&acc)[kAuxCols]) {
    const AuxGeom g = aux_geometry(a, i);
    fl = a.pflags[i];
    src1 = g.src1;
    const bool scale = (a.flags & NNR_AUX_SCALE_PCS) != 0;
    const bool grad_k = ACC && (a.flags & NNR_AUX_GRAD_K) != 0;
    const float s2 = scale ? a.scale2[0] : 1.f;
    float g_rot[3] = {0.f, 0.f, 0.f}, g_rot_s[3] = {0.f, 0.f, 0.f}, g_pc2[3] = {0.f, 0.f, 0.f};
    if (a.flags & NNR_AUX_PC) {
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const float gx = fix_get(a.gXq + 3 * i + r), gy = fix_get(a.gYq + 3 * i + r);
            g_rot[r] = gx / s2;
            g_pc2[r] = gy / s2;
            if (ACC && scale) acc[12] -= (gx * a.X[3 * i + r] + gy * a.Y[3 * i + r]) / s2;
        }
    }
    // (with_ssim: a point's colour also enters its neighbours' windows, so every point -- valid or not, of this shard or not -- may
    // carry a gradient; aux_ssim_kernel has already restricted the CENTRES to the valid points of the shard)
    const bool ssim = (a.flags & NNR_AUX_SSIM) != 0;
    // a point behind the second camera was replaced by the constant (nl, nl, nl): nothing reaches the geometry, but its projection is
    // still a function of K
    if ((a.flags & NNR_AUX_RGBS) && (grad_k || !(fl & kBehind)) && a.acc[1] > 0.f && (ssim || ((fl & kValid) && i >= a.s_lo && i < a.s_hi))) {
        float q[3], xy[2];
        aux_project(a, g, q, xy);
        const float coef = aux_g_rgbs(a) / (3.f * a.acc[1]);
        const float gx = a.gxy[2 * i] * coef, gy = a.gxy[2 * i + 1] * coef;
        const float gq[3] = {gx / q[2], gy / q[2], -(gx * q[0] + gy * q[1]) / (q[2] * q[2])};
        if (!(fl & kBehind)) {
#pragma unroll
            for (int c = 0; c < 3; ++c) g_rot_s[c] = a.K[c] * gq[0] + a.K[4 + c] * gq[1] + a.K[8 + c] * gq[2];
        }
        if (grad_k) {
            const bool behind = (fl & kBehind) != 0;
            const float p[4] = {behind ? a.nl : g.rot[0], behind ? a.nl : g.rot[1], behind ? a.nl : g.rot[2], 1.f};
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[13 + 4 * r + c] = gq[r] * p[c];
        }
    }
    float g_pc1[3] = {0.f, 0.f, 0.f};
    const bool detach = (a.flags & NNR_AUX_DETACH_RGBS) != 0;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const float gt = g_rot[r] + g_rot_s[r];
        if (ACC) {
            acc[4 * r + 0] = gt * g.pc1[0];
            acc[4 * r + 1] = gt * g.pc1[1];
            acc[4 * r + 2] = gt * g.pc1[2];
            acc[4 * r + 3] = gt;
        }
        const float gp = g_rot[r] + (detach ? 0.f : g_rot_s[r]);   // detach_rgbs_scale: the re-projection sees a detached cloud
#pragma unroll
        for (int c = 0; c < 3; ++c) g_pc1[c] += a.rel[4 * r + c] * gp;
    }
    // pc = Kinv[:, :3] (x' d, y' d, d) + Kinv[:, 3]  ->  d pc / d d = Kinv[:, :3] (x', y', 1)
    gd1 = 0.f;
    gd2 = 0.f;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const float* k = a.Kinv + 4 * r;
        const float dir = k[0] * g.xp + k[1] * g.yp + k[2];
        gd1 += g_pc1[r] * dir;
        gd2 += g_pc2[r] * dir;
        if (grad_k) {   // d pc[r] / d Kinv[r][:] = (x' d, y' d, d, 1), clamped depths included (the clamp only cuts d L / d depth)
            acc[25 + 4 * r + 0] = g_pc1[r] * g.xp * g.d1 + g_pc2[r] * g.xp * g.d2;
            acc[25 + 4 * r + 1] = g_pc1[r] * g.yp * g.d1 + g_pc2[r] * g.yp * g.d2;
            acc[25 + 4 * r + 2] = g_pc1[r] * g.d1 + g_pc2[r] * g.d2;
            acc[25 + 4 * r + 3] = g_pc1[r] + g_pc2[r];
        }
    }
}

// backward per point.  The depth-image gradients: the (hr, wr) sampling grid reads the (hd, wd) depth maps through a nearest resize, so
// the points that share a depth pixel form a RECTANGLE of the grid (the source index is monotone in y and in x).  Its first point (row-
// major) owns the pixel: it re-evaluates the other points of the rectangle and adds their gradients in row-major order -- no float
// atomics, bit-reproducible.  With a depth map at least as fine as the grid (the usual case: the grid is the image / pc_ratio) every
// rectangle is one point and nothing is evaluated twice.
__global__ __launch_bounds__(256) void aux_points_bwd_kernel(AuxArgs a) {
    __shared__ float scratch[4];
    const int i = blockIdx.x * 256 + threadIdx.x;
    float acc[kAuxCols];   // dL/d rel[r][0..3] for r = 0..2, dL/d scale2, dL/d K rows 0..2, dL/d Kinv rows 0..2
#pragma unroll
    for (int k = 0; k < kAuxCols; ++k) acc[k] = 0.f;
    if (i < a.S) {
        float gd1, gd2;
        uint32_t fl;
        int src1;
        aux_point_grads<true>(a, i, gd1, gd2, fl, src1, acc);
        if (a.aff) {      // d depth / d (scale, shift) = (raw, 1), or (raw + shift, scale) with shift_first; a clamped depth passes nothing
            const float r1 = a.d1_img[src1], r2 = a.d2_img[src1];
            if (!(fl & kClamp1)) { acc[37] = gd1 * (a.shift_first ? r1 + a.aff[1] : r1); acc[38] = gd1 * (a.shift_first ? a.aff[0] : 1.f); }
            if (!(fl & kClamp2)) { acc[39] = gd2 * (a.shift_first ? r2 + a.aff[3] : r2); acc[40] = gd2 * (a.shift_first ? a.aff[2] : 1.f); }
        }
        if (a.g_d1_img || a.g_d2_img) {
            const int y = i / a.wr, x = i - y * a.wr;
            const int sy = aux_nearest_src(y, a.hr, a.hd), sx = aux_nearest_src(x, a.wr, a.wd);
            const bool owner = (y == 0 || aux_nearest_src(y - 1, a.hr, a.hd) != sy) && (x == 0 || aux_nearest_src(x - 1, a.wr, a.wd) != sx);
            if (owner) {
                float s1 = 0.f, s2 = 0.f;
                for (int yy = y; yy < a.hr && aux_nearest_src(yy, a.hr, a.hd) == sy; ++yy) {
                    for (int xx = x; xx < a.wr && aux_nearest_src(xx, a.wr, a.wd) == sx; ++xx) {
                        float e1 = gd1, e2 = gd2, none[kAuxCols];
                        uint32_t fj = fl;
                        int sj = src1;
                        if (yy != y || xx != x) aux_point_grads<false>(a, yy * a.wr + xx, e1, e2, fj, sj, none);
                        if (!(fj & kClamp1)) s1 += e1;
                        if (!(fj & kClamp2)) s2 += e2;
                    }
                }
                if (a.g_d1_img) a.g_d1_img[src1] += s1;      // the only writer of this pixel
                if (a.g_d2_img) a.g_d2_img[src1] += s2;
            }
        }
    }
    const bool grad_k = (a.flags & NNR_AUX_GRAD_K) != 0;
#pragma unroll
    for (int k = 0; k < kAuxCols; ++k) {
        if ((k >= 13 && k < 37 && !grad_k) || (k >= 37 && !a.aff)) continue;      // (wave-uniform: the barrier inside block_sum is safe)
        const float bs = block_sum(acc[k], scratch);
        if (threadIdx.x == 0) a.part_bwd[kAuxStride * blockIdx.x + k] = bs;
    }
}

// g_rel_scale[16 (40 with NNR_AUX_GRAD_K, 44 with NNR_AUX_AFFINE)] = block partials summed in block order (bit-reproducible): columns 0..12
// at [0, 13), the K and Kinv columns 13..36 at [16, 40), the distortion columns 37..40 at [40, 44)
__global__ __launch_bounds__(64) void aux_bwd_finish_kernel(AuxArgs a, float* g_rel_scale) {
    const int nb = (a.S + 255) / 256;
    const int n_out = (a.flags & NNR_AUX_GRAD_K) ? 40 : 16;
    for (int k = 0; k < n_out; ++k) {
        const int col = k < 13 ? k : (k >= 16 ? k - 3 : -1);
        const float t = col >= 0 ? ordered_column_sum(a.part_bwd, nb, kAuxStride, col) : 0.f;
        if (threadIdx.x == 0) g_rel_scale[k] = t;
    }
    if (a.aff) {      // NNR_AUX_AFFINE: dL/d (scale1, shift1, scale2, shift2) at [40, 44)
        for (int k = 0; k < 4; ++k) {
            const float t = ordered_column_sum(a.part_bwd, nb, kAuxStride, 37 + k);
            if (threadIdx.x == 0) g_rel_scale[40 + k] = t;
        }
    }
}

// NNR_AUX_MATS_GRAD: the same sums laid out as the gradient of nnr_step_rays' 56-float `mats` block, which is where rel ([34, 50)), the
// distortion pairs ([50, 54)) and scale2 ([54]) came from -- one tensor for autograd instead of three slices (each slice's backward is a
// zero-fill, a copy and an add)
__global__ __launch_bounds__(64) void aux_bwd_finish_mats_kernel(AuxArgs a, float* g_mats) {
    const int nb = (a.S + 255) / 256;
    float mine = 0.f;      // lane l writes g_mats[l]
    for (int k = 0; k < 17; ++k) {
        const int col = k < 13 ? k : 37 + (k - 13);
        const float t = ordered_column_sum(a.part_bwd, nb, kAuxStride, col);      // (every lane gets the total)
        if ((int)threadIdx.x == (k < 12 ? 34 + k : (k == 12 ? 54 : 50 + (k - 13)))) mine = t;
    }
    if (threadIdx.x < 56) g_mats[threadIdx.x] = mine;
}

hipError_t launch_aux_fwd(const AuxArgs& a, hipStream_t st) {
    const int S = a.S, nb = (S + 255) / 256;
    const int n = a.s_hi - a.s_lo;   // this rank's source points (all of them without data parallelism)
    hipLaunchKernelGGL(aux_points_fwd_kernel, dim3(nb), dim3(256), 0, st, a);
    if ((a.flags & NNR_AUX_RGBS) && (a.flags & NNR_AUX_SSIM)) hipLaunchKernelGGL(aux_ssim_kernel, dim3(nb), dim3(256), 0, st, a);
    if (a.flags & NNR_AUX_PC) {
        static const bool brute = [] { const char* e = std::getenv("NNR_PC_SEARCH"); return e && std::string(e) == "brute"; }();
        static const bool rows = [] { const char* e = std::getenv("NNR_PC_SEARCH"); return e && std::string(e) == "rows"; }();
        if (!brute) {      // the ray-aware search: both directions, indices and distances, one launch
            // NNR_PC_SEARCH=rows / tiles: one of the two kernels alone (A/B); default: the light sources in the first, the heavy tiles in the second
            static const bool only_tiles = [] { const char* e = std::getenv("NNR_PC_SEARCH"); return e && std::string(e) == "tiles"; }();
            if (n > 0) {
                const int tiles = ((a.wr + kPcT - 1) / kPcT) * ((a.hr + kPcT - 1) / kPcT);
                float* spheres = reinterpret_cast<float*>(a.keys);      // (the key table of the exhaustive path: 4 S floats, unused here; 10 tiles words needed)
                int* heavy = reinterpret_cast<int*>(a.keys) + 8 * tiles;
                static const float area = [] { const char* e = std::getenv("NNR_PC_HEAVY_AREA"); return e ? (float)std::atof(e) : kPcHeavyArea; }();      // (tuning knobs)
                static const int vote = [] { const char* e = std::getenv("NNR_PC_HEAVY_VOTE"); return e ? std::atoi(e) : kPcHeavyVote; }();
                const bool rows_only = rows || tiles > kPcMaxTiles;
                if (!rows_only) hipLaunchKernelGGL(aux_pc_spheres_kernel, dim3(tiles, 2), dim3(64), 0, st, a, spheres);
                if (!only_tiles || rows_only)
                    hipLaunchKernelGGL(aux_pc_search_kernel, dim3(tiles, 2), dim3(64 * kPcLanes), 0, st, a, rows_only ? nullptr : heavy, spheres, area, vote);
                if (!rows_only) hipLaunchKernelGGL(aux_pc_search_tile_kernel, dim3(tiles, 2), dim3(64 * kPcTileWaves), 0, st, a, spheres, only_tiles ? nullptr : heavy);
            }
            hipLaunchKernelGGL(aux_dist_sum_kernel, dim3(nb, 2), dim3(256), 0, st, a);      // + the finishing step, in its last block
            return hipGetLastError();
        }
        // NNR_PC_SEARCH=brute: the exhaustive search of nnr_pointcloud.hip (rounds 1-4), the A/B baseline of the kernel above
        hipLaunchKernelGGL(aux_fill_keys_kernel, dim3((2 * S + 255) / 256), dim3(256), 0, st, a.keys, 2 * S);
        if (n > 0) {   // the O(S^2 / W) part: only this rank's sources search the whole destination cloud
            hipError_t e = launch_pc_nearest_keys_at(a.X + 3 * a.s_lo, a.Y, n, S, a.keys + a.s_lo, a.s_lo, st);
            if (e != hipSuccess) return e;
            e = launch_pc_nearest_keys_at(a.Y + 3 * a.s_lo, a.X, n, S, a.keys + S + a.s_lo, a.s_lo, st);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(aux_decode_kernel, dim3(nb), dim3(256), 0, st, a.keys, S, a.s_lo, a.s_hi, a.idx_xy, a.dist_xy, a.part_fwd, 2);
        hipLaunchKernelGGL(aux_decode_kernel, dim3(nb), dim3(256), 0, st, a.keys + S, S, a.s_lo, a.s_hi, a.idx_yx, a.dist_yx, a.part_fwd, 3);
    }
    hipLaunchKernelGGL(aux_finish_kernel, dim3(1), dim3(64), 0, st, a);
    return hipGetLastError();
}

hipError_t launch_aux_bwd(const AuxArgs& a, float* g_rel_scale, hipStream_t st) {
    const int S = a.S, nb = (S + 255) / 256;
    const int n = a.s_hi - a.s_lo;
    if ((a.flags & NNR_AUX_PC) && n > 0)      // (the accumulators gXq / gYq were zeroed by the forward: ONE backward per forward)
        hipLaunchKernelGGL(aux_pc_bwd_kernel, dim3((n + 255) / 256, 2), dim3(256), 0, st, a);
    hipLaunchKernelGGL(aux_points_bwd_kernel, dim3(nb), dim3(256), 0, st, a);
    if (a.flags & NNR_AUX_MATS_GRAD) hipLaunchKernelGGL(aux_bwd_finish_mats_kernel, dim3(1), dim3(64), 0, st, a, g_rel_scale);
    else hipLaunchKernelGGL(aux_bwd_finish_kernel, dim3(1), dim3(64), 0, st, a, g_rel_scale);
    return hipGetLastError();
}

}  // namespace nnr
