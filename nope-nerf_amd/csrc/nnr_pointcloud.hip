// nnr_pointcloud.hip -- nearest neighbour between two point clouds and the point-to-point error built on it.
// Replaces Loss.comp_closest_pts_idx_with_split / comp_point_point_error (model/losses.py:125-148), which materialise the
// (3, S, D) difference tensor (3.1 GB per direction at S = D = 16 128) every training step while pc_weight > 0.
//
// Roofline: VALU-bound (no reuse to speak of for a matrix unit: the per-pair work is 3 subtractions, a 3-term sum of
// squares and a compare; the dot-product form |x|^2 + |y|^2 - 2xy would put it on MFMA but changes the rounding and with
// it the argmin).  8 VALU instructions per (source, destination) pair per lane, S*D pairs: 2.1 G lane-instructions per
// direction = 53 us at 64 lanes/clk/CU x 256 CUs x 2.4 GHz (half that with packed fp32).  HBM traffic is the two clouds (390 KB) and 20 B per source point.
//
// Parity: the reference takes argmin over sqrt(dx^2 + dy^2 + dz^2) as torch evaluates it -- the sum of squares as the fma
// chain fma(dz,dz, fma(dy,dy, dx*dx)) (checked bit-for-bit against torch.linalg.norm on the host) -- and returns the FIRST
// index of the minimum.  The kernel forms the same fp32 value and orders candidates by (sqrt bits, index).
#include "nnr_device.h"
#include "nnr_kernels.h"
#include <cstdlib>

namespace nnr {

constexpr int kPcBlock = 256;   // lanes per workgroup
constexpr int kPcTile = 1024;   // destination points staged in LDS at a time (16 KB as float4)
// PER = source points per lane, held as float2 pairs so that the subtract / square / fma chain issues as packed fp32 (v_pk_*: two
// pairs per instruction); one LDS read feeds PER pairs.  Round 4 measured what bounds the kernel (profiles/r04/k_pc_nearest_sweep.txt,
// 32 400 x 32 400 points, the first training phase's clouds at 540 x 960): NOT the instruction count -- the time falls linearly with the
// number of workgroups up to ~2048 (8 waves per SIMD: 6.0 / 3.2 / 1.7 / 0.85 / 0.46 / 0.34 ms at 32 / 64 / 128 / 256 / 512 / 1024
// workgroups), i.e. it is the latency of the dependent chain LDS read -> subtract -> multiply -> fma -> fma -> compare -> branch that
// more waves hide, and longer destination ranges per workgroup (fewer takes of the bookkeeping branch) do not pay for the waves they
// cost.  PER = 2 with 2048 workgroups: 316 us against 343 for PER = 4 / 1024 (rounds 1-3).
// keys[s] = min over this block's destination range of (sqrt(d2) bits << 32 | index): distances are >= 0, so their bit
// patterns order like the values, and equal distances order by index (= first occurrence).
// (Measured at the end of round 4 and removed again: the destination points through the SCALAR cache instead of the LDS tile -- wave-uniform
// global addresses, s_load_dwordx8 / x4 for four points per trip, the coordinates as scalar operands of the packed instructions, no LDS and no
// barrier: the same indices, 12 % SLOWER (profiles/r04/w_pc_nearest_scalar_vs_lds.txt; commit 122ab07).  The LDS pipe is not what bounds the
// search; the bookkeeping branch was, see pc_nearest_one_kernel.)
// This kernel -- two sources per lane -- is the search of rounds 1-4, kept behind NNR_PC_PER=2 as the A/B baseline of the product kernel below.
template <int kPcPer>
__global__ __launch_bounds__(kPcBlock) void pc_nearest_kernel(const float* __restrict__ src, const float* __restrict__ dst, int S,
                                                              int D, int d_per_block, unsigned long long* __restrict__ keys) {
    __shared__ f32x4 tile[kPcTile];
    const int s0 = (blockIdx.x * kPcBlock + threadIdx.x) * kPcPer;
    constexpr int kPairs = kPcPer / 2;
    f32x2 x[kPairs], y[kPairs], z[kPairs];
#pragma unroll
    for (int u = 0; u < kPcPer; ++u) {
        const int sc = s0 + u < S ? s0 + u : S - 1;
        x[u >> 1][u & 1] = src[3 * sc];
        y[u >> 1][u & 1] = src[3 * sc + 1];
        z[u >> 1][u & 1] = src[3 * sc + 2];
    }
    const int d0 = blockIdx.y * d_per_block, d1 = min(D, d0 + d_per_block);
    f32x2 best_d2[kPairs];
    float best_s[kPcPer];
    int best_i[kPcPer];
#pragma unroll
    for (int u = 0; u < kPcPer; ++u) {
        best_d2[u >> 1][u & 1] = __builtin_inff();
        best_s[u] = __builtin_inff();
        best_i[u] = 0x7fffffff;
    }
    // four destination points (q: 12 floats) against this lane's sources; the four in index order, first index wins among equals
    auto trip = [&](const float (&q)[12], int i0) __attribute__((always_inline)) {
        f32x2 d2[4][kPairs];
        bool any = false;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#pragma unroll
            for (int v = 0; v < kPairs; ++v) {
                const f32x2 dx = x[v] - q[3 * k], dy = y[v] - q[3 * k + 1], dz = z[v] - q[3 * k + 2];
                // torch.linalg.norm's sum of squares: fma(dz,dz, fma(dy,dy, dx*dx)), every step rounded to fp32
                d2[k][v] = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, dx * dx));
                any = any | (d2[k][v][0] < best_d2[v][0]) | (d2[k][v][1] < best_d2[v][1]);
            }
        }
        // rare after the first few points: the sqrt and the bookkeeping stay out of the steady-state loop
        if (any) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int u = 0; u < kPcPer; ++u) {
                    const float v = d2[k][u >> 1][u & 1];
                    if (v < best_d2[u >> 1][u & 1]) {
                        const float sq = __fsqrt_rn(v);
                        if (sq < best_s[u]) { best_s[u] = sq; best_i[u] = i0 + k; }   // equal sqrt: the earlier index stays
                        best_d2[u >> 1][u & 1] = v;
                    }
                }
        }
    };
    {
        for (int t0 = d0; t0 < d1; t0 += kPcTile) {
            const int n = min(kPcTile, d1 - t0);
            __syncthreads();
            const int n4 = (n + 3) & ~3;      // four destination points per trip: pad with points at infinity (their d2 is +inf: never a minimum)
            for (int i = threadIdx.x; i < n4; i += kPcBlock) {
                const float* p = dst + 3 * (int64_t)(t0 + (i < n ? i : 0));
                tile[i] = i < n ? f32x4{p[0], p[1], p[2], 0.f} : f32x4{__builtin_inff(), __builtin_inff(), __builtin_inff(), 0.f};
            }
            __syncthreads();
            // Four points per trip, ONE branch for the four (round 4): four independent chains per source pair and a quarter of the branches.
#pragma unroll 1
            for (int i = 0; i < n4; i += 4) {
                float q[12];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const f32x4 t = tile[i + k];   // same address in every lane: an LDS broadcast
                    q[3 * k] = t[0]; q[3 * k + 1] = t[1]; q[3 * k + 2] = t[2];
                }
                trip(q, t0 + i);
            }
        }
    }
#pragma unroll
    for (int u = 0; u < kPcPer; ++u)
        if (s0 + u < S && best_i[u] != 0x7fffffff)
            atomicMin(keys + s0 + u, ((unsigned long long)__float_as_uint(best_s[u]) << 32) | (unsigned int)best_i[u]);
}

// ONE source per lane, the packed instructions over PAIRS OF DESTINATION points (end of round 4).  With two sources per lane and the destination
// range cut ~50 ways for occupancy, a wave's 128 running minima improve somewhere in nearly every trip of a ~500-point range (a minimum over n
// points improves ~ln n times, most of them early): the bookkeeping branch -- eight conditional blocks with a sqrt each -- was the steady state,
// not the exception.  Here a wave carries 64 minima (half the chance per trip, half the blocks when it happens), the same number of waves needs
// a quarter of the range cuts (four times longer ranges: the early phase is a smaller share), and the arithmetic per pair is unchanged: the
// tile holds the points as pairs [x0 x1 y0 y1 z0 z1], three 16-byte LDS broadcasts deliver four points as packed operands.
template <int GROUPS, int SRC>      // groups of four destination points per trip (one branch per trip); sources per lane (s, s + 256, ..: each with its own minimum)
__global__ __launch_bounds__(kPcBlock) void pc_nearest_one_kernel(const float* __restrict__ src, const float* __restrict__ dst, int S, int D,
                                                                  int d_per_block, unsigned long long* keys, int seed, int s_off) {
    __shared__ f32x4 tile[3 * kPcTile / 4];      // 12 floats per four points
    int s[SRC], sc[SRC];
    float sx[SRC], sy[SRC], sz[SRC];
    f32x2 x[SRC], y[SRC], z[SRC];
#pragma unroll
    for (int u = 0; u < SRC; ++u) {
        s[u] = (blockIdx.x * SRC + u) * kPcBlock + threadIdx.x;
        sc[u] = s[u] < S ? s[u] : S - 1;
        sx[u] = src[3 * sc[u]]; sy[u] = src[3 * sc[u] + 1]; sz[u] = src[3 * sc[u] + 2];
        x[u] = f32x2{sx[u], sx[u]}; y[u] = f32x2{sy[u], sy[u]}; z[u] = f32x2{sz[u], sz[u]};
    }
    const int d0 = blockIdx.y * d_per_block, d1 = min(D, d0 + d_per_block);
    // The running minimum as the key it will be merged with: (sqrt bits, index), lexicographic -- candidates may arrive in ANY order (seeds
    // from other ranges, below), the smaller index wins among equal rounded distances as in the reference.  The steady-state test is one
    // compare of d2 against `thr`, an upper bound of every d2 whose rounded square root is <= best_s (s^2 (1 + 4 ulp): sqrt_rn(v) <= s implies
    // v <= s^2 (1 + 2^-23) up to the rounding of s^2 itself); what passes is decided exactly.
    float best_s[SRC], thr[SRC];
    int best_i[SRC];
#pragma unroll
    for (int u = 0; u < SRC; ++u) { best_s[u] = __builtin_inff(); thr[u] = __builtin_inff(); best_i[u] = 0x7fffffff; }
    auto consider = [&](int u, float v, int idx) __attribute__((always_inline)) {
        if (v <= thr[u] && v < __builtin_inff()) {      // (padding points and infinite distances never match)
            const float sq = __fsqrt_rn(v);
            if (sq < best_s[u] || (sq == best_s[u] && idx < best_i[u])) {
                best_s[u] = sq;
                best_i[u] = idx;
                const float t = sq * sq;
                thr[u] = __builtin_fmaf(t, 4.8e-7f, t) + 1.2e-38f;      // (+ the smallest normal: the relative margin vanishes when t is subnormal)
            }
        }
    };
    // Seeds (end of round 4).  A range that starts from +infinity spends its first hundreds of points improving a minimum that another range
    // has long beaten.  (i) keys[s] only ever decreases and every value it held is a real candidate's key: whatever is there when this block
    // starts -- the later half of the grid starts after the earlier half has finished -- is a valid starting point.  (ii) Before any key
    // exists: the eight destination points around the source's own index, wherever they lie (the two clouds of the trainer are depth maps of
    // neighbouring frames on the same pixel grid: the point at the same pixel is a good first guess).  The result is the same minimum over a
    // superset of the range's candidates: indices bit-identical (tests/test_pointcloud.py).
    unsigned long long k0[SRC];
    float seed_s[SRC];
    int seed_i[SRC];
#pragma unroll
    for (int u = 0; u < SRC; ++u) {
        k0[u] = ~0ull;
        if (seed) {
            k0[u] = s[u] < S ? __atomic_load_n(keys + s[u], __ATOMIC_RELAXED) : ~0ull;
            if (k0[u] != ~0ull) {
                best_s[u] = __uint_as_float((unsigned int)(k0[u] >> 32));
                best_i[u] = (int)(unsigned int)(k0[u] & 0xffffffffu);
                const float t = best_s[u] * best_s[u];
                thr[u] = __builtin_fmaf(t, 4.8e-7f, t) + 1.2e-38f;      // (+ the smallest normal: the relative margin vanishes when t is subnormal)
            } else if (seed > 1) {
#pragma unroll
                for (int j = -4; j < 4; ++j) {
                    const int q = min(max(sc[u] + s_off + j, 0), D - 1);      // s_off: the source range's place in its cloud (data parallel shards)
                    const float ex = sx[u] - dst[3 * q], ey = sy[u] - dst[3 * q + 1], ez = sz[u] - dst[3 * q + 2];
                    consider(u, __builtin_fmaf(ez, ez, __builtin_fmaf(ey, ey, ex * ex)), q);
                }
            }
        }
        seed_s[u] = best_s[u];
        seed_i[u] = best_i[u];
    }
    for (int t0 = d0; t0 < d1; t0 += kPcTile) {
        const int n = min(kPcTile, d1 - t0);
        __syncthreads();
        constexpr int kTrip = 4 * GROUPS;
        const int n4 = (n + kTrip - 1) / kTrip * kTrip;      // pad with points at infinity (their d2 is +inf: never a minimum)
        float* const tf = reinterpret_cast<float*>(tile);
        for (int i = threadIdx.x; i < n4; i += kPcBlock) {
            const float* p = dst + 3 * (int64_t)(t0 + (i < n ? i : 0));
            const float inf = __builtin_inff();
            const int o = 6 * (i >> 1) + (i & 1);          // pair i / 2: [x0 x1 y0 y1 z0 z1]
            tf[o] = i < n ? p[0] : inf;
            tf[o + 2] = i < n ? p[1] : inf;
            tf[o + 4] = i < n ? p[2] : inf;
        }
        __syncthreads();
#pragma unroll 1
        for (int i = 0; i < n4; i += kTrip) {
            float d2[SRC][kTrip];
            bool any = false;
#pragma unroll
            for (int g = 0; g < GROUPS; ++g) {
                const int t = 3 * ((i >> 2) + g);
                const f32x4 a = tile[t], b = tile[t + 1], c = tile[t + 2];   // x0 x1 y0 y1 | z0 z1 x2 x3 | y2 y3 z2 z3
#pragma unroll
                for (int u = 0; u < SRC; ++u) {
                    const f32x2 dx0 = x[u] - f32x2{a[0], a[1]}, dy0 = y[u] - f32x2{a[2], a[3]}, dz0 = z[u] - f32x2{b[0], b[1]};
                    const f32x2 dx1 = x[u] - f32x2{b[2], b[3]}, dy1 = y[u] - f32x2{c[0], c[1]}, dz1 = z[u] - f32x2{c[2], c[3]};
                    // torch.linalg.norm's sum of squares: fma(dz,dz, fma(dy,dy, dx*dx)), every step rounded to fp32
                    const f32x2 e0 = __builtin_elementwise_fma(dz0, dz0, __builtin_elementwise_fma(dy0, dy0, dx0 * dx0));
                    const f32x2 e1 = __builtin_elementwise_fma(dz1, dz1, __builtin_elementwise_fma(dy1, dy1, dx1 * dx1));
                    d2[u][4 * g] = e0[0]; d2[u][4 * g + 1] = e0[1]; d2[u][4 * g + 2] = e1[0]; d2[u][4 * g + 3] = e1[1];
                    any = any | (e0[0] <= thr[u]) | (e0[1] <= thr[u]) | (e1[0] <= thr[u]) | (e1[1] <= thr[u]);
                }
            }
            if (any) {
#pragma unroll
                for (int u = 0; u < SRC; ++u)
#pragma unroll
                    for (int k = 0; k < kTrip; ++k) consider(u, d2[u][k], t0 + i + k);
            }
        }
    }
#pragma unroll
    for (int u = 0; u < SRC; ++u)
        if (s[u] < S && best_i[u] != 0x7fffffff && (best_s[u] != seed_s[u] || best_i[u] != seed_i[u] || k0[u] == ~0ull))
            atomicMin(keys + s[u], ((unsigned long long)__float_as_uint(best_s[u]) << 32) | (unsigned int)best_i[u]);
}

__global__ void pc_fill_keys_kernel(unsigned long long* keys, int S) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < S) keys[s] = ~0ull;
}

__global__ void pc_decode_kernel(const unsigned long long* keys, int S, int64_t* idx, float* dist) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= S) return;
    const unsigned long long k = keys[s];
    idx[s] = (int64_t)(unsigned int)(k & 0xffffffffu);
    dist[s] = __uint_as_float((unsigned int)(k >> 32));
}

// d mean_s ||src_s - dst_idx(s)|| : g_src[s] = g/S * (src_s - dst_idx)/dist  (0 where dist == 0, like torch's norm backward),
// g_dst[j] -= the same for every source matched to j.  Several sources may share a destination; instead of float atomics (whose sum
// depends on arrival order) thread j GATHERS: it walks all sources in index order (LDS tiles of the match table) and adds the terms of
// those matched to it -- S x D integer compares, no atomics, bit-reproducible.
__global__ __launch_bounds__(256) void pc_error_bwd_kernel(const float* __restrict__ src, const float* __restrict__ dst,
                                                           const int64_t* __restrict__ idx, const float* __restrict__ dist,
                                                           const float* __restrict__ g_loss, int S, int D, float* __restrict__ g_src,
                                                           float* __restrict__ g_dst) {
    __shared__ int match[256];
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (g_src && t < S) {
        const int64_t j = idx[t];
        // A source with NaN / inf coordinates never beats the initial key: its decoded index is 0xffffffff and its "distance" a NaN
        // bit pattern.  No match -> no gradient (and no access 51 GB past the destination cloud); the trainer's NaN check stops the
        // run on the loss value.
        const float dd = dist[t];
        const float w = (j != 0xffffffffll && dd > 0.f) ? g_loss[0] / ((float)S * dd) : 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) g_src[3 * t + c] = j != 0xffffffffll ? w * (src[3 * t + c] - dst[3 * j + c]) : 0.f;
    }
    if (!g_dst) return;
    float acc[3] = {0.f, 0.f, 0.f};
    float dj[3] = {0.f, 0.f, 0.f};
    if (t < D) { dj[0] = dst[3 * t]; dj[1] = dst[3 * t + 1]; dj[2] = dst[3 * t + 2]; }
    for (int base = 0; base < S; base += 256) {
        const int s = base + threadIdx.x;
        __syncthreads();
        match[threadIdx.x] = (s < S && idx[s] != 0xffffffffll) ? (int)idx[s] : -1;
        __syncthreads();
        const int n = S - base < 256 ? S - base : 256;
        if (t < D) {
#pragma unroll 8
            for (int k = 0; k < n; ++k) {
                if (match[k] == t) {
                    const int q = base + k;
                    const float dd = dist[q];
                    const float w = dd > 0.f ? g_loss[0] / ((float)S * dd) : 0.f;
#pragma unroll
                    for (int c = 0; c < 3; ++c) acc[c] -= w * (src[3 * q + c] - dj[c]);
                }
            }
        }
    }
    if (t < D) {
#pragma unroll
        for (int c = 0; c < 3; ++c) g_dst[3 * t + c] += acc[c];
    }
}

// keys must hold ~0 (or earlier candidates): the search only lowers them
hipError_t launch_pc_nearest_keys_at(const float* src, const float* dst, int S, int D, unsigned long long* keys, int s_off, hipStream_t st);
hipError_t launch_pc_nearest_keys(const float* src, const float* dst, int S, int D, unsigned long long* keys, hipStream_t st) {
    return launch_pc_nearest_keys_at(src, dst, S, D, keys, 0, st);
}
// s_off: index of src[0] in the cloud the S sources are a range of (the "same pixel" seed looks around s_off + s in dst)
hipError_t launch_pc_nearest_keys_at(const float* src, const float* dst, int S, int D, unsigned long long* keys, int s_off, hipStream_t st) {
    // The product search: pc_nearest_one_kernel, one source per lane, eight points per trip, seeded ranges, the destination range cut until
    // there are ~8192 workgroups (>= 256 points per range).  End of round 4, at 20 736 / 32 400 points: 88 / 189 us against 145 / 297 for
    // the search of rounds 1-4 (two sources per lane, 2048 workgroups: NNR_PC_PER=2).  Sweeps: profiles/r04/x2_*, y2_*, y3_* (workgroups 256 ..
    // 16 384; 4 / 8 / 16 points per trip; two sources per lane of the seeded kernel: 104 / 198 us).
    // NNR_PC_WGS: workgroups to aim for; NNR_PC_SEED: 0 = every range starts from +infinity, 1 = from the key already there, 2 (default) = and,
    // before a key exists, from the points around the source's own index.
    static const int per = [] { const char* e = std::getenv("NNR_PC_PER"); return e && std::atoi(e) == 2 ? 2 : 1; }();
    static const int wgs = [] { const char* e = std::getenv("NNR_PC_WGS"); const int v = e ? std::atoi(e) : (per == 1 ? 8192 : 2048); return v < 1 ? 1 : v; }();
    static const int seed = [] { const char* e = std::getenv("NNR_PC_SEED"); return e ? std::atoi(e) : 2; }();
    const int bx = (S + kPcBlock * per - 1) / (kPcBlock * per);
    int split = (wgs + bx - 1) / bx;
    const int max_split = (D + 255) / 256;
    split = split < 1 ? 1 : (split > max_split ? max_split : split);
    const int d_per_block = ((D + split - 1) / split + 255) / 256 * 256;
    const int by = (D + d_per_block - 1) / d_per_block;
    if (per == 1) hipLaunchKernelGGL((pc_nearest_one_kernel<2, 1>), dim3(bx, by), dim3(kPcBlock), 0, st, src, dst, S, D, d_per_block, keys, seed, s_off);
    else hipLaunchKernelGGL(pc_nearest_kernel<2>, dim3(bx, by), dim3(kPcBlock), 0, st, src, dst, S, D, d_per_block, keys);
    return hipGetLastError();
}

hipError_t launch_pc_nearest(const float* src, const float* dst, int S, int D, int64_t* idx, float* dist, unsigned long long* keys,
                             hipStream_t st) {
    hipLaunchKernelGGL(pc_fill_keys_kernel, dim3((S + 255) / 256), dim3(256), 0, st, keys, S);
    hipError_t e = launch_pc_nearest_keys(src, dst, S, D, keys, st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(pc_decode_kernel, dim3((S + 255) / 256), dim3(256), 0, st, keys, S, idx, dist);
    return hipGetLastError();
}

hipError_t launch_pc_error_bwd(const float* src, const float* dst, const int64_t* idx, const float* dist, const float* g_loss, int S,
                               int D, float* g_src, float* g_dst, hipStream_t st) {
    const int n = g_dst && D > S ? D : S;
    hipLaunchKernelGGL(pc_error_bwd_kernel, dim3((n + 255) / 256), dim3(256), 0, st, src, dst, idx, dist, g_loss, S, D, g_src, g_dst);
    return hipGetLastError();
}

}  // namespace nnr
