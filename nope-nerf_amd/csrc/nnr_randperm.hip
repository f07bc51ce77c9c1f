// nnr_randperm.hip -- the first r entries of torch.randperm(n, device='cuda'), bit for bit, without sorting all n keys.
// Replaces the pixel pick of reference model/training.py:257 (`torch.randperm(h*w)[:n_points]`): 22 launches / 0.2 ms of a
// 5 ms step for 518 400 pixels of which 1 024 are kept.
//
// What torch does (ATen/native/cuda/Randperm.cu, Randperm.cuh; 64-bit key branch, n > ~46 000): draw n int64 keys with
// random_(), stable-radix-sort (key & mask, index) over the low `bits` bits, then re-shuffle every island of equal keys with
// Fisher-Yates driven by Philox (seed, subsequence = island start position, offset = the generator's offset before the
// call).  The first r positions of that permutation are the indices of the r smallest masked keys in (key, index) order,
// with the islands that START among them shuffled the same way.  So: keep the candidates below a threshold chosen for
// E = r + 20 sqrt(r) + 64 expected hits (>= 16 sigma above r, >= 40 sigma below the buffer), order those, replay the island
// shuffles.  The candidates are ordered by COUNTING, not by a sorting network: the rank of a candidate is the number of smaller
// ones, m^2 independent 64-bit compares spread over the whole chip (m = 1.7 k for the 1024-ray pick, 10 k for the 8192-ray pick
// of an 8-rank data-parallel step, where every rank draws the whole step's permutation), instead of log^2(m) barrier-separated
// passes of one workgroup (48 us at m = 1.7 k, 184 us at m = 10 k for the previous in-LDS bitonic sort).
// The host side draws the keys with the same torch call and advances the generator exactly as torch would
// (nope-nerf_amd/nnr/sampling.py, which also self-checks against torch.randperm on first use).
#include <hiprand/hiprand_kernel.h>

#include <cmath>

#include "nnr_device.h"
#include "nnr_kernels.h"

namespace nnr {

constexpr int kRpCaps[3] = {4096, 16384, 65536};   // candidate capacities
constexpr int kSelKeys = 8;                         // keys per thread of the select kernel
constexpr int kRankTile = 1024;                     // candidates staged in LDS per pass of the rank kernel

// scratch layout (32-bit words): [0] candidate count, [1] status (1 = fewer than r candidates or more than `cap`),
// [2, 2 + cap) rank of every candidate, then `cap` u64 candidates (masked key << idx_bits | index), then `cap` u64 in order.
__device__ __forceinline__ unsigned int* rp_rank(unsigned int* scratch) { return scratch + 2; }
__device__ __forceinline__ unsigned long long* rp_cand(unsigned int* scratch, unsigned int cap) {
    return reinterpret_cast<unsigned long long*>(scratch + 2 + cap);
}

// keys below the threshold -> candidate list (any order).  One global atomic per workgroup: the hits of a workgroup reserve
// their slots through an LDS counter first (with one global atomic per hit the kernel was bound by same-address atomics:
// 24 us for 1.7 k hits, 68 us for 10 k).
__global__ __launch_bounds__(256) void randperm_select_kernel(const int64_t* __restrict__ keys, int64_t n, unsigned long long mask,
                                                              unsigned long long limit, int idx_bits, unsigned int cap,
                                                              unsigned int* __restrict__ scratch) {
    __shared__ unsigned int s_count, s_base;
    if (threadIdx.x == 0) s_count = 0;
    __syncthreads();
    const int64_t i0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * kSelKeys;
    unsigned long long k[kSelKeys];
    unsigned int hits = 0;                                    // bit u: key i0 + u is a candidate
#pragma unroll
    for (int u = 0; u < kSelKeys; ++u) {
        k[u] = i0 + u < n ? ((unsigned long long)keys[i0 + u] & mask) : ~0ull;
        hits |= (i0 + u < n && k[u] < limit) ? (1u << u) : 0u;
    }
    const unsigned int nh = __popc(hits);
    const unsigned int local = nh ? atomicAdd(&s_count, nh) : 0u;
    __syncthreads();
    if (threadIdx.x == 0) s_base = s_count ? atomicAdd(scratch, s_count) : 0u;
    __syncthreads();
    unsigned long long* cand = rp_cand(scratch, cap);
#pragma unroll
    for (int u = 0; u < kSelKeys; ++u)
        if ((hits >> u) & 1u) {
            const unsigned int p = s_base + local + __popc(hits & ((1u << u) - 1u));
            if (p < cap) cand[p] = (k[u] << idx_bits) | (unsigned long long)(i0 + u);
        }
}

// rank[i] += #{ j in this workgroup's slice of the candidates : cand[j] < cand[i] }.  grid.x = blocks of 256 candidates,
// grid.y = slices of the compared-against range; packed (key, index) values are distinct, so ranks are a permutation of 0..m-1.
__global__ __launch_bounds__(256) void randperm_rank_kernel(unsigned int* __restrict__ scratch, unsigned int cap) {
    __shared__ unsigned long long tile[kRankTile];
    const unsigned int m = min(scratch[0], cap);
    if (blockIdx.x * 256u >= m) return;
    const unsigned long long* cand = rp_cand(scratch, cap);
    const unsigned int i = blockIdx.x * 256u + threadIdx.x;
    const unsigned long long mine = i < m ? cand[i] : 0ull;
    const unsigned int per = (m + gridDim.y - 1) / gridDim.y;
    const unsigned int j0 = blockIdx.y * per, j1 = min(m, j0 + per);
    unsigned int below = 0;
    for (unsigned int t0 = j0; t0 < j1; t0 += kRankTile) {
        __syncthreads();
        for (unsigned int t = threadIdx.x; t < kRankTile; t += 256) tile[t] = t0 + t < j1 ? cand[t0 + t] : ~0ull;
        __syncthreads();
        const unsigned int cnt = min((unsigned)kRankTile, j1 - t0);
        for (unsigned int t = 0; t < cnt; t += 4) {   // the padding compares as "not below": no tail handling needed
            below += (tile[t] < mine) + (tile[t + 1] < mine) + (tile[t + 2] < mine) + (tile[t + 3] < mine);
        }
    }
    if (i < m && below) atomicAdd(rp_rank(scratch) + i, below);
}

__global__ __launch_bounds__(1024) void randperm_finish_kernel(unsigned int* __restrict__ scratch, unsigned int cap, int r, int idx_bits,
                                                               unsigned long long seed, unsigned long long offset,
                                                               int64_t* __restrict__ out) {
    const unsigned long long* cand = rp_cand(scratch, cap);
    unsigned long long* s = rp_cand(scratch, cap) + cap;       // the candidates in (masked key, index) order: what a stable
    const unsigned int* rank = rp_rank(scratch);               // sort by key leaves
    const unsigned int count = scratch[0];
    const int m = (int)min(count, cap);
    if (count < (unsigned)r || count > cap) {
        // The candidate buffer under- or overflowed: the output would be zero-filled or truncated.  By construction this has
        // probability < 1e-50 (16 sigma below, 40 sigma above the expected count), so nobody polls a status word for it -- but a
        // wrong pixel pick must never pass silently: record the status and abort the launch (the process dies with a HIP
        // exception at its next synchronisation).
        if (threadIdx.x == 0) scratch[1] = 1u;
        __threadfence_system();
        __builtin_trap();
    }
    for (int i = threadIdx.x; i < m; i += 1024) s[rank[i]] = cand[i];
    __syncthreads();
    // islands of equal keys that begin inside the first r positions: torch's randperm_handle_duplicate_keys_kernel, one thread
    // per island start (positions are those of the full sorted array, since every smaller key is a candidate)
    const unsigned long long idx_mask = (1ull << idx_bits) - 1;
    for (int tid = threadIdx.x; tid < r && tid < m - 1; tid += 1024) {
        const unsigned long long key = s[tid] >> idx_bits;
        if (key != (s[tid + 1] >> idx_bits)) continue;
        if (tid != 0 && key == (s[tid - 1] >> idx_bits)) continue;
        int island = 0;
        do { island++; } while (tid + island < m && (s[tid + island] >> idx_bits) == key);
        hiprandStatePhilox4_32_10_t state;
        hiprand_init(seed, tid, offset, &state);
        for (int i = island - 1; i > 0; i--) {
            const unsigned int q = hiprand(&state) % (i + 1);
            if ((unsigned)i != q) {   // swap the data (index) parts; the keys are equal
                const unsigned long long t = s[tid + i];
                s[tid + i] = s[tid + q];
                s[tid + q] = t;
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < r; i += 1024) out[i] = i < m ? (int64_t)(s[i] & idx_mask) : 0;
    // leave the counters the way this call found them: ZERO.  The scratch buffer is persistent (nnr/sampling.py keeps one per device and
    // capacity, zero-filled once), so no memset launch precedes the select kernel of the next pick.
    unsigned int* rank_w = rp_rank(scratch);
    for (int i = threadIdx.x; i < m; i += 1024) rank_w[i] = 0u;
    if (threadIdx.x == 0) { scratch[0] = 0u; scratch[1] = 0u; }
}

// expected candidate count and the buffer that holds it with >= 40 sigma to spare; 0 = r beyond the largest buffer
long double randperm_expected(int r) { return (long double)r + 20.0L * sqrtl((long double)r) + 64.0L; }
unsigned int randperm_capacity(int r) {
    const long double e = randperm_expected(r), hi = e + 40.0L * sqrtl(e);
    for (int c : kRpCaps)
        if (hi <= c) return (unsigned)c;
    return 0;
}

hipError_t launch_randperm_prefix(const int64_t* keys, int64_t n, int bits, int r, unsigned long long seed, unsigned long long offset,
                                  int64_t* out, unsigned int* scratch, hipStream_t st) {
    int idx_bits = 1;
    while ((1ll << idx_bits) < n) ++idx_bits;
    const unsigned long long mask = bits >= 64 ? ~0ull : ((1ull << bits) - 1);
    const unsigned int cap = randperm_capacity(r);
    const long double frac = randperm_expected(r) / (long double)n;
    const unsigned long long limit = (unsigned long long)((long double)(bits >= 64 ? 18446744073709551615.0L : (long double)(1ull << bits)) * frac);
    // (count, status and ranks are zero on entry: the caller zero-fills the buffer ONCE, every call leaves them zeroed)
    const int64_t per_block = 256 * kSelKeys;
    hipLaunchKernelGGL(randperm_select_kernel, dim3((unsigned)((n + per_block - 1) / per_block)), dim3(256), 0, st, keys, n, mask, limit,
                       idx_bits, cap, scratch);
    const unsigned int gx = cap / 256, gy = cap <= 4096 ? 16 : cap <= 16384 ? 8 : 4;     // >= 256 workgroups when all are live
    hipLaunchKernelGGL(randperm_rank_kernel, dim3(gx, gy), dim3(256), 0, st, scratch, cap);
    hipLaunchKernelGGL(randperm_finish_kernel, dim3(1), dim3(1024), 0, st, scratch, cap, r, idx_bits, seed, offset, out);
    return hipGetLastError();
}

}  // namespace nnr
