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
// shuffles.  The candidates are ordered by COUNTING inside buckets of their leading key bits (the keys are uniform): a candidate's place is
// its bucket's start (a prefix sum over 4096 bucket sizes) + the number of smaller members of its bucket -- one workgroup, O(m) work
// (m = 1.7 k for the 1024-ray pick, 10 k for the 8192-ray pick of an 8-rank data-parallel step: every rank needs the whole step's pick,
// because the depth loss is normalised by the step's GLOBAL count of valid depths).  History: an in-LDS bitonic sort (48 us at m = 1.7 k,
// 184 us at 10 k), then counting over all m^2 pairs spread over the chip (5 us / 40 us, a kernel of its own).
// The host side draws the keys with the same torch call and advances the generator exactly as torch would
// (nope-nerf_amd/nnr/sampling.py, which also self-checks against torch.randperm on first use).
#include <hiprand/hiprand_kernel.h>

#include <cmath>

#include "nnr_device.h"
#include "nnr_kernels.h"

namespace nnr {

constexpr int kRpCaps[3] = {4096, 16384, 65536};   // candidate capacities
constexpr int kSelKeys = 8;                         // keys per thread of the select kernel

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

// (rounds 2-4 ranked the candidates by counting over ALL pairs -- m^2 64-bit compares spread over the chip in a kernel of its own: 5 us at
// m = 1.7 k, but 40 us at the m = 10 k of an 8-rank data-parallel step, where every rank needs the whole step's pick.  The keys are uniform:
// the finishing workgroup now buckets them -- see below -- and a candidate is compared with its bucket only.)
constexpr int kRpBuckets = 4096;

__global__ __launch_bounds__(1024) void randperm_finish_kernel(unsigned int* __restrict__ scratch, unsigned int cap, int r, int idx_bits,
                                                               unsigned long long limit, unsigned long long seed, unsigned long long offset,
                                                               int64_t* __restrict__ out) {
    unsigned long long* const cand = rp_cand(scratch, cap);       // in: the candidates in any order; then: `s`
    unsigned long long* const grouped = rp_cand(scratch, cap) + cap;   // the candidates grouped by bucket
    unsigned long long* const s = cand;                            // the candidates in (masked key, index) order: what a stable sort by key leaves
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
    // Order the candidates: bucket by the leading bits (uniform keys: m / 4096 .. m / 2048 per bucket), prefix-sum the bucket sizes, and a
    // candidate's position is its bucket's start + the number of smaller members of its bucket.  All in this one workgroup: histogram and
    // scan in LDS, the candidates regrouped by bucket in the second candidate array, the ordered ones back into the first.
    {
        __shared__ unsigned int base[kRpBuckets + 1], cur[kRpBuckets], wsum[16];
        const unsigned long long top = limit << idx_bits;                          // every candidate is below this
        const int nbits = top > 1 ? 64 - __builtin_clzll(top - 1) : 1;
        const int sh = nbits > 12 ? nbits - 12 : 0;
        for (int b = threadIdx.x; b < kRpBuckets; b += 1024) cur[b] = 0u;
        __syncthreads();
        for (int i = threadIdx.x; i < m; i += 1024) atomicAdd(&cur[(unsigned int)(cand[i] >> sh)], 1u);
        __syncthreads();
        // exclusive scan of the 4096 counts: four per thread, wave scan, the 16 wave totals
        unsigned int c[4], tsum = 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) { c[u] = cur[4 * threadIdx.x + u]; tsum += c[u]; }
        unsigned int inc = tsum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned int o = __shfl_up(inc, d, 64);
            if ((int)(threadIdx.x & 63) >= d) inc += o;
        }
        if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = inc;
        __syncthreads();
        unsigned int before = inc - tsum;
        for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) before += wsum[w];
#pragma unroll
        for (int u = 0; u < 4; ++u) { base[4 * threadIdx.x + u] = before; before += c[u]; }
        if (threadIdx.x == 1023) base[kRpBuckets] = before;
        __syncthreads();
        for (int b = threadIdx.x; b < kRpBuckets; b += 1024) cur[b] = base[b];
        __syncthreads();
        for (int i = threadIdx.x; i < m; i += 1024) {
            const unsigned long long v = cand[i];
            grouped[atomicAdd(&cur[(unsigned int)(v >> sh)], 1u)] = v;
        }
        __syncthreads();      // (every candidate has been read: its array now takes the ordered ones)
        for (int p = threadIdx.x; p < m; p += 1024) {
            const unsigned long long mine = grouped[p];
            const unsigned int b = (unsigned int)(mine >> sh), j0 = base[b], j1 = base[b + 1];
            unsigned int below = 0;
            for (unsigned int j = j0; j < j1; j += 4) {      // four independent loads in flight (the last one repeated past the end: never below)
                unsigned long long o[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) o[u] = grouped[min(j + u, j1 - 1)];
#pragma unroll
                for (int u = 0; u < 4; ++u) below += (j + u < j1) & (o[u] < mine);      // packed (key, index) values are distinct
            }
            s[j0 + below] = mine;
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
    // leave the counters the way this call found them: ZERO.  The scratch buffer is persistent (nnr/sampling.py keeps one per device and
    // capacity, zero-filled once), so no memset launch precedes the select kernel of the next pick.
    unsigned int* rank_w = rp_rank(scratch);
    for (int i = threadIdx.x; i < m; i += 1024) rank_w[i] = 0u;
    if (threadIdx.x == 0) { scratch[0] = 0u; scratch[1] = 0u; }
}

// ---- rows of torch.rand(total, device='cuda') without drawing the rest (the stratified-sampling jitter of a data-parallel shard) ------------
// torch's uniform kernel (ATen/native/cuda/DistributionTemplates.h: distribution_nullary_kernel, unroll 4) runs T = 256 * grid threads; thread
// t starts Philox at (seed, subsequence t, offset) and its c-th curand_uniform4 call yields the elements t + T (4 c + ii), ii = 0..3.  So
// element li is component (li / T) & 3 of call (li / T) >> 2 of thread li % T -- one Philox evaluation per element here (a rank's rows are
// less than T elements apart from each other only within a call: nothing to share), against T * calls for the whole tensor.  The value:
// curand_uniform's (0, 1] with 1 mapped to 0 (uniform_kernel's "reverse the bounds").  The caller advances the generator by what the full
// draw would have consumed (nnr/sampling.py, which also checks this against torch.rand itself on first use).
__global__ __launch_bounds__(256) void uniform_rows_kernel(unsigned long long seed, unsigned long long offset, unsigned long long T,
                                                           unsigned long long first, unsigned long long n, float* __restrict__ out) {
    const unsigned long long e = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    const unsigned long long li = first + e, q = li / T, t = li - q * T;
    hiprandStatePhilox4_32_10_t state;
    hiprand_init(seed, t, offset + 4ull * (q >> 2), &state);
    const float4 r = hiprand_uniform4(&state);
    const unsigned int ii = (unsigned int)(q & 3ull);
    const float v = ii == 0 ? r.x : (ii == 1 ? r.y : (ii == 2 ? r.z : r.w));
    out[e] = v == 1.f ? 0.f : v;
}

hipError_t launch_uniform_rows(unsigned long long seed, unsigned long long offset, unsigned long long threads, unsigned long long first,
                               unsigned long long n, float* out, hipStream_t st) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(uniform_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, seed, offset, threads, first, n, out);
    return hipGetLastError();
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
    hipLaunchKernelGGL(randperm_finish_kernel, dim3(1), dim3(1024), 0, st, scratch, cap, r, idx_bits, limit, seed, offset, out);
    return hipGetLastError();
}

}  // namespace nnr
