// hbm_stream.hip -- what HBM read rate the two load paths of this repository's kernels can reach on the box, for sizing the bf16
// weight-gradient kernel (which is bound by streaming 9 KB/sample exactly once):
//   A. plain global_load_dwordx4 to registers, many waves per CU
//   B. global_load_lds_dwordx4 (LDS-DMA, 1 KiB per wave-instruction) into a ring of 32 KiB stages, 4 waves per workgroup,
//      counted vmcnt + one barrier pair per stage -- the staging pattern of an LDS-fed MFMA kernel with one workgroup per CU
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/hbm_stream.hip -o tools/ubench/hbm_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int UNROLL>
__global__ __launch_bounds__(256) void plain_stream(const f32x4* __restrict__ src, size_t f4_per_block, float* out) {
    const f32x4* p = src + (size_t)blockIdx.x * f4_per_block + threadIdx.x;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (size_t i = 0; i < f4_per_block; i += 256 * UNROLL) {
        f32x4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = p[i + 256 * u];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc += v[u];
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[0] = 1.f;
}

constexpr int kStageF4 = 2048;   // 32 KiB

template <int NST, int PIECES /* per wave per stage */, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void dma_stream(const f32x4* __restrict__ src, int n_stage, float* out) {
    extern __shared__ __attribute__((aligned(16))) f32x4 lds[];
    constexpr int kF4 = PIECES * WAVES * 64;   // f32x4 per stage
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const f32x4* base = src + (size_t)blockIdx.x * n_stage * kF4;
    auto issue = [&](int s) {
        const f32x4* g = base + (size_t)s * kF4 + (PIECES * wave) * 64 + lane;
        f32x4* l = lds + (s % NST) * kF4 + (PIECES * wave) * 64;
#pragma unroll
        for (int i = 0; i < PIECES; ++i)
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(g + i * 64), (lds_ptr_t)(l + i * 64), 16, 0, 0);
    };
    for (int s = 0; s < NST - 1 && s < n_stage; ++s) issue(s);
    float acc = 0.f;
    for (int s = 0; s < n_stage; ++s) {
        if (s + NST - 1 < n_stage) {
            issue(s + NST - 1);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES * (NST - 1)) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        acc += lds[(s % NST) * kF4 + threadIdx.x][0];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    if (acc == 12345.678f) out[0] = 1.f;
}

template <class F>
static double time_ms(F launch, int reps = 5) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

// The access pattern of the bf16 weight-gradient kernel (nnr_wgrad_bf16.hip): every stage is half from one plane, half from another
// (`two`), the planes advance by `stride_kib` per stage instead of the bytes read (17 KiB for 16 read), waves take 1 KiB blocks
// round-robin (`rr`) and lanes read permuted 64-byte groups of their block (`swz`).
template <int NST>
__global__ __launch_bounds__(256) void dma_pattern(const char* __restrict__ src, int n_stage, float* out, int two, int stride_kib,
                                                   int rr, int swz, size_t plane_gap) {
    extern __shared__ __attribute__((aligned(16))) f32x4 lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t stage_stride = two ? (size_t)stride_kib * 1024 : (size_t)stride_kib * 2048;
    const char* base = src + (size_t)blockIdx.x * n_stage * stage_stride;
    const int ph = lane >> 5, pc = lane & 31;
    const int lane_off = swz ? (32 * ph + (pc ^ (4 * ph))) * 16 : lane * 16;
    auto issue = [&](int s) {
        f32x4* l = lds + (s % NST) * 2048;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int blk = rr ? wave + 4 * q : wave * 8 + q;     // 32 blocks of 1 KiB per stage
            const char* g = two ? base + (blk >> 4) * plane_gap + (size_t)s * stage_stride + (blk & 15) * 1024
                                : base + (size_t)s * stage_stride + blk * 1024;
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(g + lane_off), (lds_ptr_t)(l + blk * 64), 16, 0, 0);
        }
    };
    for (int s = 0; s < NST - 1 && s < n_stage; ++s) issue(s);
    float acc = 0.f;
    for (int s = 0; s < n_stage; ++s) {
        if (s + NST - 1 < n_stage) {
            issue(s + NST - 1);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8 * (NST - 1)) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        acc += lds[(s % NST) * 2048 + threadIdx.x][0];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    if (acc == 12345.678f) out[0] = 1.f;
}

int main() {
    const size_t bytes = (size_t)4 << 30;
    f32x4* src;
    float* out;
    CK(hipMalloc(&src, bytes));
    CK(hipMalloc(&out, 64));
    CK(hipMemset(src, 1, bytes));
    const size_t n_f4 = bytes / 16;
    for (int bpc : {1, 2, 4, 8, 16}) {
        const int grid = 256 * bpc;
        const size_t per = n_f4 / grid;
        double ms = time_ms([&] { hipLaunchKernelGGL(plain_stream<8>, dim3(grid), dim3(256), 0, 0, src, per, out); });
        printf("plain x4 loads, 8 in flight/thread, %2d WG/CU: %.3f ms  %.2f TB/s\n", bpc, ms, bytes / ms / 1e9);
    }
#define RUN_DMA(NST, PIECES, WAVES, BPC)                                                                                     \
    do {                                                                                                                     \
        const int grid = 256 * (BPC);                                                                                        \
        const int kF4 = (PIECES) * (WAVES) * 64;                                                                             \
        const int n_stage = (int)(n_f4 / grid / kF4);                                                                        \
        const size_t lds_bytes = (size_t)(NST) * kF4 * 16;                                                                   \
        auto k = dma_stream<NST, PIECES, WAVES>;                                                                             \
        CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));               \
        double ms = time_ms([&] { hipLaunchKernelGGL(k, dim3(grid), dim3(64 * (WAVES)), lds_bytes, 0, src, n_stage, out); }); \
        printf("LDS-DMA ring: %d stages x %3d KiB, %d waves, %d WG/CU (%3zu KiB LDS/WG): %.3f ms  %.2f TB/s\n", NST, kF4 * 16 / 1024, \
               WAVES, BPC, lds_bytes / 1024, ms, (double)grid * n_stage * kF4 * 16 / ms / 1e9);                              \
    } while (0)
    RUN_DMA(2, 8, 4, 1);
    RUN_DMA(3, 8, 4, 1);
    RUN_DMA(4, 8, 4, 1);
    RUN_DMA(4, 4, 4, 1);
    RUN_DMA(6, 4, 4, 1);
    RUN_DMA(8, 4, 4, 1);
    RUN_DMA(2, 8, 4, 2);
    RUN_DMA(3, 4, 4, 2);
    RUN_DMA(4, 4, 4, 2);
    RUN_DMA(4, 4, 8, 1);
    RUN_DMA(4, 2, 8, 1);
    RUN_DMA(8, 2, 8, 1);
    RUN_DMA(4, 2, 4, 4);
    {
        auto k = dma_pattern<4>;
        CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 32768));
        struct { int two, stride, rr, swz; const char* what; } cases[] = {
            {0, 16, 0, 0, "one stream (as above)"}, {0, 16, 1, 0, "blocks round-robin over the waves"},
            {0, 16, 0, 1, "lanes permuted in 64 B groups"}, {1, 16, 0, 0, "two planes, 16 KiB each per stage"},
            {1, 17, 0, 0, "two planes, 17 KiB stride"}, {1, 17, 1, 1, "two planes, 17 KiB stride, round-robin, permuted (the wgrad pattern)"},
            {1, 24, 0, 0, "two planes, 24 KiB stride"}};
        for (auto& c : cases)
            for (size_t gap : {(size_t)2 << 30, ((size_t)2 << 30) - (5 << 20) - 13 * 4096}) {
                if (!c.two && gap != (size_t)2 << 30) continue;
                const size_t per_wg = (c.two ? bytes / 2 : bytes) / 256;
                const int n_stage = (int)(per_wg / ((size_t)c.stride * (c.two ? 1024 : 2048))) - 1;
                double ms = time_ms([&] { hipLaunchKernelGGL(k, dim3(256), dim3(256), 4 * 32768, 0, (const char*)src, n_stage, out, c.two, c.stride,
                                                             c.rr, c.swz, gap); });
                printf("pattern: %-70s gap %zu: %.3f ms  %.2f TB/s\n", c.what, gap, ms, 256.0 * n_stage * 32768 / ms / 1e9);
            }
    }
    return 0;
}
