// nnr_randperm.hip -- the first r entries of torch.randperm(n, device='cuda'), bit for bit, without sorting all n keys.
// Replaces the pixel pick of reference model/training.py:257 (`torch.randperm(h*w)[:n_points]`): 22 launches / 0.2 ms of a
// 5 ms step for 518 400 pixels of which 1 024 are kept.
//
// What torch does (ATen/native/cuda/Randperm.cu, Randperm.cuh; 64-bit key branch, n > ~46 000): draw n int64 keys with
// random_(), stable-radix-sort (key & mask, index) over the low `bits` bits, then re-shuffle every island of equal keys with
// Fisher-Yates driven by Philox (seed, subsequence = island start position, offset = the generator's offset before the
// call).  The first r positions of that permutation are the indices of the r smallest masked keys in (key, index) order,
// with the islands that START among them shuffled the same way.  So: keep the candidates below a threshold chosen for
// E = r + 20 sqrt(r) + 64 expected hits (>= 16 sigma above r, >= 40 sigma below the buffer), sort those in LDS, replay the
// island shuffles.  Two buffer sizes: 4096 candidates (32 KB of LDS) for the single-GPU pick of 1024 rays, 16384 (128 KB) for
// the up-to-12288-ray picks of a data-parallel step, where every rank draws the whole step's permutation.
// The host side draws the keys with the same torch call and advances the generator exactly as torch would
// (nope-nerf_amd/nnr/sampling.py, which also self-checks against torch.randperm on first use).
#include <hiprand/hiprand_kernel.h>

#include <cmath>

#include "nnr_device.h"
#include "nnr_kernels.h"

namespace nnr {

constexpr int kRpCapSmall = 4096, kRpCapLarge = 16384;   // candidate capacities (powers of two: bitonic sort)

// scratch layout: [0] candidate count (u32), [1] status (u32: 1 = fewer than r candidates or more than the capacity), then
// `cap` u64 candidates (masked key << idx_bits | index)
__global__ void randperm_select_kernel(const int64_t* __restrict__ keys, int64_t n, unsigned long long mask, unsigned long long limit,
                                       int idx_bits, unsigned int cap, unsigned int* __restrict__ scratch) {
    unsigned long long* cand = reinterpret_cast<unsigned long long*>(scratch + 2);
    // two consecutive keys per thread, the grid covers n once
    const int64_t i0 = 2 * ((int64_t)blockIdx.x * blockDim.x + threadIdx.x);
    if (i0 >= n) return;
    const long long kk[2] = {keys[i0], i0 + 1 < n ? keys[i0 + 1] : 0ll};
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        if (i0 + u >= n) break;
        const unsigned long long k = (unsigned long long)kk[u] & mask;
        if (k < limit) {
            const unsigned int p = atomicAdd(scratch, 1u);
            if (p < cap) cand[p] = (k << idx_bits) | (unsigned long long)(i0 + u);
        }
    }
}

template <int kRpCap>
__global__ __launch_bounds__(1024) void randperm_finish_kernel(unsigned int* __restrict__ scratch, int r, int idx_bits,
                                                               unsigned long long seed, unsigned long long offset,
                                                               int64_t* __restrict__ out) {
    __shared__ unsigned long long s[kRpCap];
    const unsigned long long* cand = reinterpret_cast<const unsigned long long*>(scratch + 2);
    const unsigned int count = scratch[0];
    const int m = count < (unsigned)kRpCap ? (int)count : kRpCap;
    if (threadIdx.x == 0) scratch[1] = (count < (unsigned)r || count > (unsigned)kRpCap) ? 1u : 0u;
    int sort_n = 2048;                                          // the smallest power of two holding the candidates
    while (sort_n < m) sort_n <<= 1;
    for (int i = threadIdx.x; i < sort_n; i += 1024) s[i] = i < m ? cand[i] : ~0ull;
    __syncthreads();
    // bitonic sort, ascending by (masked key, index): the order a stable sort by key leaves
    for (int k = 2; k <= sort_n; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < sort_n / 2; t += 1024) {
                const int lo = ((t / j) * 2 * j) + (t % j), hi = lo + j;
                const bool up = (lo & k) == 0;
                const unsigned long long a = s[lo], b = s[hi];
                if ((a > b) == up) { s[lo] = b; s[hi] = a; }
            }
            __syncthreads();
        }
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
}

// expected candidate count and the buffer that holds it with >= 40 sigma to spare; 0 = r too large for the LDS sort
long double randperm_expected(int r) { return (long double)r + 20.0L * sqrtl((long double)r) + 64.0L; }
unsigned int randperm_capacity(int r) {
    const long double e = randperm_expected(r), hi = e + 40.0L * sqrtl(e);
    return hi <= kRpCapSmall ? kRpCapSmall : hi <= kRpCapLarge ? kRpCapLarge : 0;
}

hipError_t launch_randperm_prefix(const int64_t* keys, int64_t n, int bits, int r, unsigned long long seed, unsigned long long offset,
                                  int64_t* out, unsigned int* scratch, hipStream_t st) {
    int idx_bits = 1;
    while ((1ll << idx_bits) < n) ++idx_bits;
    const unsigned long long mask = bits >= 64 ? ~0ull : ((1ull << bits) - 1);
    const long double expect = randperm_expected(r);
    const unsigned int cap = randperm_capacity(r);
    const long double frac = expect / (long double)n;
    const unsigned long long limit = (unsigned long long)((long double)(bits >= 64 ? 18446744073709551615.0L : (long double)(1ull << bits)) * frac);
    hipError_t e = hipMemsetAsync(scratch, 0, 2 * sizeof(unsigned int), st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(randperm_select_kernel, dim3((unsigned)((n / 2 + 256) / 256)), dim3(256), 0, st, keys, n, mask, limit, idx_bits,
                       cap, scratch);
    if (cap == (unsigned)kRpCapSmall)
        hipLaunchKernelGGL(randperm_finish_kernel<kRpCapSmall>, dim3(1), dim3(1024), 0, st, scratch, r, idx_bits, seed, offset, out);
    else
        hipLaunchKernelGGL(randperm_finish_kernel<kRpCapLarge>, dim3(1), dim3(1024), 0, st, scratch, r, idx_bits, seed, offset, out);
    return hipGetLastError();
}

}  // namespace nnr
