// hbm_stash_write.hip -- what HBM WRITE rate the stash pattern of the bf16 training kernels can reach on the box, with nothing else
// going on: the forward writes 2.8 GB per launch at 4096 x 128 (the bf16 activations of 8 layers + colour hidden + encodings +
// gates), the input-gradient kernel 2.4 GB, both in 1 KiB wave-stores to tile-major planes (nnr_layout.h): a wave owns T = 2 chunks
// of 32 samples and, layer after layer, writes the 16 blocks [chunk][group] of that layer's plane -- 16 KiB contiguous per chunk and
// plane, but the planes are 256 MiB apart and the 16 stores of a layer are spread over the layer's MFMA work.
// Patterns:
//   seq     every wave writes one contiguous range (a fill kernel)
//   stash   the kernels' order: for plane, for group: the 1 KiB blocks of the wave's two chunks
//   stash1  the same with a single block per store step (one chunk per wave)
// each with ordinary and non-temporal stores, at 1 workgroup of 4 waves per CU (the kernels' occupancy: LDS-bound) and at 4.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/hbm_stash_write.hip -o tools/ubench/hbm_stash_write
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int kPlanes = 9, kGroups = 16;        // 9 planes of 16 blocks per chunk = 144 KiB per chunk, 4.5 KB per sample
constexpr int kBlockF4 = 64;                    // one 1 KiB block = 64 lanes x 16 bytes

template <bool NT>
__device__ __forceinline__ void put(f32x4* p, f32x4 v) {
    if (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}

// mode 0: seq, 1: stash (T chunks per wave), grid-stride over chunk groups like the kernels' grid (one workgroup = 4 waves = 4 T chunks)
template <bool NT, int T, int MODE>
__global__ __launch_bounds__(256) void writer(f32x4* dst, long chunks, long plane_f4) {
    extern __shared__ char lds[];               // occupancy control only
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const f32x4 v = {1.f, 2.f, 3.f, (float)lane};
    const long first = ((long)blockIdx.x * 4 + wave) * T;
    if (first >= chunks) return;
    if (MODE == 0) {
        // the same bytes per wave, contiguous: T * kPlanes * kGroups blocks
        f32x4* p = dst + first * (kPlanes * kGroups * kBlockF4) + lane;
#pragma unroll 4
        for (int i = 0; i < T * kPlanes * kGroups; ++i) put<NT>(p + (long)i * kBlockF4, v);
    } else {
        for (int pl = 0; pl < kPlanes; ++pl) {
            f32x4* base = dst + pl * plane_f4 + lane;
#pragma unroll 4
            for (int g = 0; g < kGroups; ++g)
#pragma unroll
                for (int n = 0; n < T; ++n) put<NT>(base + ((first + n) * kGroups + g) * kBlockF4, v);
        }
    }
}

template <bool NT, int T, int MODE>
static void run(const char* name, f32x4* dst, long chunks, long plane_f4, int lds_bytes) {
    const int blocks = (int)((chunks + 4 * T - 1) / (4 * T));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(writer<NT, T, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((writer<NT, T, MODE>), dim3(blocks), dim3(256), lds_bytes, 0, dst, chunks, plane_f4);
    CK(hipEventRecord(e0));
    const int reps = 10;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((writer<NT, T, MODE>), dim3(blocks), dim3(256), lds_bytes, 0, dst, chunks, plane_f4);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    const double bytes = (double)chunks * kPlanes * kGroups * 1024;
    printf("%-58s %s  %d WG/CU: %.3f ms  %.2f TB/s\n", name, NT ? "non-temporal" : "ordinary    ", lds_bytes > 100000 ? 1 : 4, ms, bytes / ms * 1e-9);
}

int main() {
    const long chunks = 16384;                                  // 524 288 samples
    const long plane_f4 = chunks * kGroups * kBlockF4;          // 256 MiB planes
    f32x4* dst;
    CK(hipMalloc(&dst, (size_t)kPlanes * plane_f4 * sizeof(f32x4)));
    for (int lds : {144 * 1024, 36 * 1024}) {
        run<false, 2, 0>("seq: one contiguous range per wave", dst, chunks, plane_f4, lds);
        run<true, 2, 0>("seq: one contiguous range per wave", dst, chunks, plane_f4, lds);
        run<false, 2, 1>("stash: plane by plane, group by group, 2 chunks per wave", dst, chunks, plane_f4, lds);
        run<true, 2, 1>("stash: plane by plane, group by group, 2 chunks per wave", dst, chunks, plane_f4, lds);
        run<false, 1, 1>("stash: plane by plane, group by group, 1 chunk per wave", dst, chunks, plane_f4, lds);
        run<true, 1, 1>("stash: plane by plane, group by group, 1 chunk per wave", dst, chunks, plane_f4, lds);
    }
    CK(hipFree(dst));
    return 0;
}
