// Does the ORDER of accumulating MFMAs matter?  96 x v_mfma_f32_32x32x16_bf16 per iteration on 16 accumulators (4 x 4, as a weight-gradient step),
// (a) six consecutive MFMAs into the same accumulator (the weight gradient's order: sub-tile outer, term inner), (b) the same 96 with consecutive
// MFMAs into four different accumulators (term outer, sub-tile inner).  One wave per SIMD, 256 workgroups x 256 threads; counter ticks per MFMA.
//   hipcc --offload-arch=gfx950 -O3 mfma_chain.hip -o mfma_chain
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int ORDER>
__global__ __launch_bounds__(256, 1) void chain(float* out, uint64_t* clk, int iters) {
    f32x16 acc[4][4];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    bf16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 8; ++r) { a[i][r] = (__bf16)(0.001f * (threadIdx.x + i + r)); b[i][r] = (__bf16)(0.002f * (threadIdx.x - i + r)); }
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int g = 0; g < 24; ++g) {
                const int j = ORDER == 0 ? g / 6 : g % 4;
                const int t = ORDER == 0 ? g % 6 : g / 4;
                __builtin_amdgcn_sched_barrier(0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(i + t) & 3], b[(j + t) & 3], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            asm volatile("" : "+a"(acc[i][0]), "+a"(acc[i][1]), "+a"(acc[i][2]), "+a"(acc[i][3]));
        }
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][15];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

int main() {
    float* d_out; uint64_t* d_clk; uint64_t clk;
    hipMalloc(&d_out, 256 * 256 * 4); hipMalloc(&d_clk, 8);
    const int iters = 2000;
    for (int order = 0; order < 2; ++order)
        for (int rep = 0; rep < 3; ++rep) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            if (order == 0) chain<0><<<256, 256>>>(d_out, d_clk, iters); else chain<1><<<256, 256>>>(d_out, d_clk, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(&clk, d_clk, 8, hipMemcpyDeviceToHost);
            printf("%s: %.3f ms for %d x 96 MFMAs per wave = %.1f ns per MFMA (%.1f TFLOP/s on 1024 waves), %.2f counter ticks per MFMA\n",
                   order == 0 ? "six in a row into one accumulator " : "consecutive into four accumulators", ms, iters, ms * 1e6 / (96.0 * iters),
                   1024.0 * 96 * iters * 32768.0 * 2 / 2 / (ms * 1e-3) / 1e12, (double)clk / (96.0 * iters));
        }
    return 0;
}
