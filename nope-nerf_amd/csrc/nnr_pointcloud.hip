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
    for (int t0 = d0; t0 < d1; t0 += kPcTile) {
        const int n = min(kPcTile, d1 - t0);
        __syncthreads();
        const int n4 = (n + 3) & ~3;      // four destination points per trip: pad with points at infinity (their d2 is +inf: never a minimum)
        for (int i = threadIdx.x; i < n4; i += kPcBlock) {
            const float* p = dst + 3 * (int64_t)(t0 + (i < n ? i : 0));
            tile[i] = i < n ? f32x4{p[0], p[1], p[2], 0.f} : f32x4{__builtin_inff(), __builtin_inff(), __builtin_inff(), 0.f};
        }
        __syncthreads();
        // Four points per trip, ONE branch for the four (round 4): the chain LDS read -> subtract -> multiply -> fma -> fma -> compare ->
        // branch is what this kernel waits for (see above); four independent chains per source pair and a quarter of the branches.
#pragma unroll 1
        for (int i = 0; i < n4; i += 4) {
            f32x2 d2[4][kPairs];
            bool any = false;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const f32x4 q = tile[i + k];   // same address in every lane: an LDS broadcast
#pragma unroll
                for (int v = 0; v < kPairs; ++v) {
                    const f32x2 dx = x[v] - q[0], dy = y[v] - q[1], dz = z[v] - q[2];
                    // torch.linalg.norm's sum of squares: fma(dz,dz, fma(dy,dy, dx*dx)), every step rounded to fp32
                    d2[k][v] = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, dx * dx));
                    any = any | (d2[k][v][0] < best_d2[v][0]) | (d2[k][v][1] < best_d2[v][1]);
                }
            }
            // rare after the first few points: the sqrt and the bookkeeping stay out of the steady-state loop; the four points in index order
            if (any) {
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int u = 0; u < kPcPer; ++u) {
                        const float v = d2[k][u >> 1][u & 1];
                        if (v < best_d2[u >> 1][u & 1]) {
                            const float sq = __fsqrt_rn(v);
                            if (sq < best_s[u]) { best_s[u] = sq; best_i[u] = t0 + i + k; }   // equal sqrt: the earlier index stays
                            best_d2[u >> 1][u & 1] = v;
                        }
                    }
            }
        }
    }
#pragma unroll
    for (int u = 0; u < kPcPer; ++u)
        if (s0 + u < S && best_i[u] != 0x7fffffff)
            atomicMin(keys + s0 + u, ((unsigned long long)__float_as_uint(best_s[u]) << 32) | (unsigned int)best_i[u]);
}

// ---- the same search with exact pruning (round 4) ------------------------------------------------------------------------------------
// The brute-force kernel above evaluates S x D pairs; in the trainer both clouds are back-projected depth maps in raster order (32 400
// points each at 540 x 960), where 64 consecutive destination points are a short curve in space and almost all of them are far from any
// given source.  A workgroup takes 512 consecutive sources (two per lane) and a PART of the destination's 64-point chunks; it first forms
// the axis-aligned box of every chunk of its part (LDS), then visits a chunk only if some source of the wave could still find its
// minimum there:  lb2 (1 - 1e-5) <= best_d2,  lb2 = the squared distance from the source to the box, best_d2 the smallest d2 the
// source has seen.  Exactness: every point of a skipped chunk has d2 >= lb2 > best_d2 (1 + 1e-5) -- five orders above the rounding of
// either side and of the sqrt bucket in which two d2 give the same distance -- so neither the minimum nor a tie for it can sit there;
// visited points are evaluated with the brute-force kernel's arithmetic (the fma chain of torch.linalg.norm, then sqrt) and merged by
// (sqrt bits, index) lexicographically -- the order of visits does not matter, the first index of the minimum wins as in torch.argmin.
// The chunk nearest to the wave's middle source is visited first (a good bound at once, whatever the order of the points); parts
// are merged through the same 64-bit atomicMin as the brute-force kernel's destination ranges.  A chunk's points are not staged:
// lane k holds point k and v_readlane hands its coordinates to the whole wave as scalars.
constexpr int kPcChunk = 64;
constexpr int kPcMaxPartChunks = 128;      // boxes of a part in LDS: 4 KB

__global__ __launch_bounds__(256) void pc_nearest_pruned_kernel(const float* __restrict__ src, const float* __restrict__ dst, int S, int D,
                                                                int chunks_per_part, unsigned long long* __restrict__ keys) {
    __shared__ float box[kPcMaxPartChunks][8];      // lo.xyz, hi.xyz of every chunk of this part (empty chunk: lo = +inf, hi = -inf)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n_chunks = (D + kPcChunk - 1) / kPcChunk;
    const int c0 = blockIdx.y * chunks_per_part, c1 = min(n_chunks, c0 + chunks_per_part);
    const float inf = __builtin_inff();
    // ---- boxes: wave w takes the chunks c0 + w, c0 + w + 4, ... ----
    for (int c = c0 + wave; c < c1; c += 4) {
        const int j = c * kPcChunk + lane;
        float lo[3], hi[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float v = j < D ? dst[3 * (int64_t)j + a] : __builtin_nanf("");
            lo[a] = v == v ? v : inf;            // a NaN coordinate never wins a comparison in the search either: keep it out of the box
            hi[a] = v == v ? v : -inf;
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1)
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                lo[a] = fminf(lo[a], __shfl_xor(lo[a], d, 64));
                hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], d, 64));
            }
        if (lane == 0) {
#pragma unroll
            for (int a = 0; a < 3; ++a) { box[c - c0][a] = lo[a]; box[c - c0][4 + a] = hi[a]; }
        }
    }
    __syncthreads();
    // ---- this lane's two sources ----
    const int s0 = (blockIdx.x * 256 + threadIdx.x) * 2;
    float x[2], y[2], z[2], best_d2[2], best_thr[2], best_s[2];      // best_thr = best_d2 (1 + 4e-7): see visit()
    int best_i[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int sc = s0 + u < S ? s0 + u : S - 1;
        x[u] = src[3 * (int64_t)sc]; y[u] = src[3 * (int64_t)sc + 1]; z[u] = src[3 * (int64_t)sc + 2];
        best_d2[u] = inf; best_thr[u] = inf; best_s[u] = inf; best_i[u] = 0x7fffffff;
    }
    // squared distance of source u to the box of part-chunk c (0 inside; NaN-free for finite sources: an empty box gives +inf)
    auto lb2 = [&](int c, int u) __attribute__((always_inline)) {
        const float ex = fmaxf(fmaxf(box[c][0] - x[u], x[u] - box[c][4]), 0.f);
        const float ey = fmaxf(fmaxf(box[c][1] - y[u], y[u] - box[c][5]), 0.f);
        const float ez = fmaxf(fmaxf(box[c][2] - z[u], z[u] - box[c][6]), 0.f);
        return fmaf(ez, ez, fmaf(ey, ey, ex * ex));
    };
    auto visit = [&](int c) __attribute__((always_inline)) {      // all 64 points of chunk c0 + c against this lane's two sources
        const int j = (c0 + c) * kPcChunk + lane;
        const float px = j < D ? dst[3 * (int64_t)j] : inf, py = j < D ? dst[3 * (int64_t)j + 1] : inf, pz = j < D ? dst[3 * (int64_t)j + 2] : inf;
#pragma unroll 8
        for (int k = 0; k < kPcChunk; ++k) {
            const float qx = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, px), k));
            const float qy = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, py), k));
            const float qz = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pz), k));
            float d2[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const float dx = x[u] - qx, dy = y[u] - qy, dz = z[u] - qz;
                d2[u] = fmaf(dz, dz, fmaf(dy, dy, dx * dx));      // torch.linalg.norm's sum of squares, as in the brute-force kernel
            }
            // A candidate is anything whose DISTANCE could equal the best one: two d2 up to 1.2e-7 apart (relative) round to the same sqrt,
            // and of those the smaller INDEX wins (the brute-force kernel meets it first) even if its d2 is the larger one -- hence the
            // threshold a hair above the smallest d2 seen, and the lexicographic (sqrt, index) comparison.  d2 = +inf is never a match.
            if (((d2[0] <= best_thr[0]) & (d2[0] < inf)) | ((d2[1] <= best_thr[1]) & (d2[1] < inf))) {
                const int idx = (c0 + c) * kPcChunk + k;
#pragma unroll
                for (int u = 0; u < 2; ++u)
                    if ((d2[u] <= best_thr[u]) & (d2[u] < inf)) {
                        const float sq = __fsqrt_rn(d2[u]);
                        if (sq < best_s[u] || (sq == best_s[u] && idx < best_i[u])) { best_s[u] = sq; best_i[u] = idx; }
                        if (d2[u] < best_d2[u]) { best_d2[u] = d2[u]; best_thr[u] = d2[u] * 1.0000004f; }
                    }
            }
        }
    };
    // ---- the chunk nearest to the wave's middle source first ----
    const int n_part = c1 - c0;
    int seed = 0;
    if (n_part > 0) {
        const float mx = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x[0]), 32));
        const float my = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, y[0]), 32));
        const float mz = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, z[0]), 32));
        float best = inf;
        for (int c = 0; c < n_part; ++c) {
            const float ex = fmaxf(fmaxf(box[c][0] - mx, mx - box[c][4]), 0.f), ey = fmaxf(fmaxf(box[c][1] - my, my - box[c][5]), 0.f);
            const float ez = fmaxf(fmaxf(box[c][2] - mz, mz - box[c][6]), 0.f);
            const float l = fmaf(ez, ez, fmaf(ey, ey, ex * ex));
            if (l < best) { best = l; seed = c; }
        }
        seed = __builtin_amdgcn_readfirstlane(seed);
        visit(seed);
    }
    for (int c = 0; c < n_part; ++c) {
        if (c == seed) continue;
        const bool need = (lb2(c, 0) * 0.99999f <= best_d2[0]) | (lb2(c, 1) * 0.99999f <= best_d2[1]);
        if (__builtin_amdgcn_ballot_w64(need) != 0) visit(c);      // wave-uniform: readlane needs every lane's point
    }
#pragma unroll
    for (int u = 0; u < 2; ++u)
        if (s0 + u < S && best_i[u] != 0x7fffffff)
            atomicMin(keys + s0 + u, ((unsigned long long)__float_as_uint(best_s[u]) << 32) | (unsigned int)best_i[u]);
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
hipError_t launch_pc_nearest_keys(const float* src, const float* dst, int S, int D, unsigned long long* keys, hipStream_t st) {
    // PER sources per lane and the number of workgroups to aim for (the destination range is split until there are about that many;
    // at least 256 destination points per workgroup): knobs for experiments, defaults from the round-4 sweep (profiles/r04/)
    // NNR_PC_PRUNED=1: the pruned search instead of the brute-force kernel -- the same indices bit for bit (tests/test_gpu_pc_pruned.py).
    // NOT the default: measured (profiles/r04/m_pc_nearest_pruned.txt, n_*) it is 1.3 - 2x SLOWER than brute force on clouds whose depth is
    // white noise per pixel (bench.py's synthetic batch: the boxes of 64 consecutive pixels span the whole depth range and prune little)
    // and it is bound by the latency of its v_readlane -> subtract -> fma -> compare chain like the brute-force kernel is by its own;
    // on smooth depth maps see the same file.
    static const bool pruned = std::getenv("NNR_PC_PRUNED") != nullptr && std::getenv("NNR_PC_BRUTE") == nullptr;
    if (pruned) {
        static const int parts_target = [] { const char* e = std::getenv("NNR_PC_PARTS"); const int v = e ? std::atoi(e) : 8; return v < 1 ? 1 : v; }();
        const int n_chunks = (D + kPcChunk - 1) / kPcChunk;
        int cpp = (n_chunks + parts_target - 1) / parts_target;
        cpp = cpp > kPcMaxPartChunks ? kPcMaxPartChunks : (cpp < 1 ? 1 : cpp);
        const int by = (n_chunks + cpp - 1) / cpp;
        hipLaunchKernelGGL(pc_nearest_pruned_kernel, dim3((S + 511) / 512, by), dim3(256), 0, st, src, dst, S, D, cpp, keys);
        return hipGetLastError();
    }
    static const int per = [] { const char* e = std::getenv("NNR_PC_PER"); const int v = e ? std::atoi(e) : 2; return v == 4 ? 4 : 2; }();
    static const int wgs = [] { const char* e = std::getenv("NNR_PC_WGS"); const int v = e ? std::atoi(e) : 2048; return v < 1 ? 1 : v; }();
    const int bx = (S + kPcBlock * per - 1) / (kPcBlock * per);
    int split = (wgs + bx - 1) / bx;
    const int max_split = (D + 255) / 256;
    split = split < 1 ? 1 : (split > max_split ? max_split : split);
    const int d_per_block = ((D + split - 1) / split + 255) / 256 * 256;
    const int by = (D + d_per_block - 1) / d_per_block;
    if (per == 4) hipLaunchKernelGGL(pc_nearest_kernel<4>, dim3(bx, by), dim3(kPcBlock), 0, st, src, dst, S, D, d_per_block, keys);
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
