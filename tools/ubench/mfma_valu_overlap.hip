// mfma_valu_overlap.hip -- can ONE wave hide VALU work under its own MFMAs on gfx950?  The bf16 MLP kernels run one wave per SIMD and
// put their epilogue (v_accvgpr_read x2, v_cvt_pk_bf16_f32, v_pk_max_i16, v_pk_min_u16, v_lshl_or_b32 per value pair) into the gaps
// between v_mfma_f32_32x32x16_bf16 instructions (8 passes = 32 cycles each).  Measured here, per wave and with all four SIMDs of every CU
// busy: cycles per MFMA for a stream of 12 independent MFMAs with K extra instructions after each one, K = 0, 2, 4, 6, 8, of three kinds
// (plain fp32 FMAs on VGPRs / the epilogue mix reading OTHER accumulators / ds_read_b128).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/mfma_valu_overlap.hip -o tools/ubench/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int kAcc = 12;
static double g_alone = 0;

template <int KIND, int K>
__device__ __forceinline__ void extra(float (&v)[8], f32x16 (&other)[4], uint32_t (&pk)[4], uint32_t& mw, const char* lds, f32x4& frag, int j) {
    if (KIND == 0) {
#pragma unroll
        for (int i = 0; i < K; ++i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[i % 8]) : "v"(v[(i + 1) % 8]));
    } else if (KIND == 1) {   // the epilogue mix, K / 6 pairs' worth (K = 2: the two reads only)
        float x0, x1;
        if (K >= 2) asm volatile("v_accvgpr_read_b32 %0, %2\n\tv_accvgpr_read_b32 %1, %3" : "=v"(x0), "=v"(x1) : "a"(other[j & 3][0]), "a"(other[j & 3][1]));
        if (K >= 3) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk[j & 3]) : "v"(x0), "v"(x1));
        if (K >= 4) asm volatile("v_pk_max_i16 %0, %0, 0" : "+v"(pk[j & 3]));
        if (K >= 6) {
            uint32_t t;
            asm volatile("v_pk_min_u16 %0, %1, %2" : "=v"(t) : "v"(pk[j & 3]), "s"(0x00010001u));
            asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(mw) : "v"(t));
        }
        if (K >= 8) asm volatile("v_accvgpr_read_b32 %0, %2\n\tv_accvgpr_read_b32 %1, %3" : "=v"(x0), "=v"(x1) : "a"(other[j & 3][2]), "a"(other[j & 3][3]));
    } else {
#pragma unroll
        for (int i = 0; i < K; ++i) asm volatile("ds_read_b128 %0, %1" : "=v"(frag) : "v"((unsigned)(size_t)lds + 16 * (threadIdx.x & 63)));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}

template <int KIND, int K>
__global__ __launch_bounds__(256, 1) void bench(unsigned long long* out, int iters) {
    extern __shared__ char lds[];
    f32x16 acc[kAcc], other[4];   // 12 + 4 tiles = the 256 AGPRs
#pragma unroll
    for (int i = 0; i < kAcc; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) other[i][r] = (float)(threadIdx.x + r);
    float v[8] = {1.f, 1.0001f, 0.9999f, 1.f, 1.f, 1.f, 1.f, 1.f};
    uint32_t pk[4] = {0, 0, 0, 0}, mw = 0x00010001u;
    f32x4 frag = {0, 0, 0, 0};
    const bf16x8 a = {1, 1, 1, 1, 1, 1, 1, 1}, b = {1, 1, 1, 1, 1, 1, 1, 1};
#pragma unroll
    for (int i = 0; i < 4; ++i) asm volatile("" : "+a"(other[i]));
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < kAcc; ++j) {
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[j]) : "v"(a), "v"(b));
            extra<KIND, K>(v, other, pk, mw, lds, frag, j);
        }
    }
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = v[0] + frag[0] + (float)pk[0] + (float)pk[1] + (float)mw;
#pragma unroll
    for (int i = 0; i < kAcc; ++i) s += acc[i][0];
    if (s == 12345.678f) out[1] = 1;
    if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = t1 - t0;
}

// The kernels' row: 12 MFMAs whose A operands are six fragments refilled IN PLACE by ds_read_b128 right after their second (last) MFMA
// and waited for with a counted s_waitcnt before their first MFMA of the next row; UNITS epilogue units of K instructions in evenly
// spread gaps.  READS = 0: fragments stay in registers (no LDS traffic).
template <int READS, int UNITS, int K>
__global__ __launch_bounds__(256, 1) void row_bench(unsigned long long* out, int iters) {
    extern __shared__ char lds[];
    f32x16 acc[kAcc], other[4];
#pragma unroll
    for (int i = 0; i < kAcc; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) other[i][r] = (float)(threadIdx.x + r);
    float v[8] = {1.f, 1.0001f, 0.9999f, 1.f, 1.f, 1.f, 1.f, 1.f};
    uint32_t pk[4] = {0, 0, 0, 0}, mw = 0x00010001u;
    f32x4 dummy = {0, 0, 0, 0};
    f32x4 frag[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) frag[i] = f32x4{1.f, 1.f, 1.f, 1.f};
    const bf16x8 b = {1, 1, 1, 1, 1, 1, 1, 1};
    const unsigned addr = (unsigned)(size_t)lds + 16 * (threadIdx.x & 63) + 1024 * (threadIdx.x >> 6);
#pragma unroll
    for (int i = 0; i < 4; ++i) asm volatile("" : "+a"(other[i]));
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < kAcc; ++j) {
            const int f = j / 2;
            if (READS && (j & 1) == 0) asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(frag[f]));   // the five younger refills may be outstanding
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[j]) : "v"(frag[f]), "v"(b));
            if (READS && (j & 1) == 1) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(frag[f]) : "v"(addr), "n"(4096 * (j / 2)));
#pragma unroll
            for (int u = 0; u < UNITS; ++u)
                if (j == (u * kAcc) / (UNITS ? UNITS : 1)) extra<1, K>(v, other, pk, mw, lds, dummy, j + u);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = v[0] + (float)pk[0] + (float)pk[1] + (float)mw + frag[0][0] + frag[5][1];
#pragma unroll
    for (int i = 0; i < kAcc; ++i) s += acc[i][0];
    if (s == 12345.678f) out[1] = 1;
    if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = t1 - t0;
}

template <int READS, int UNITS, int K>
static void run_row(unsigned long long* out) {
    const int iters = 2000;
    hipLaunchKernelGGL((row_bench<READS, UNITS, K>), dim3(256), dim3(256), 32768, 0, out, iters);
    hipLaunchKernelGGL((row_bench<READS, UNITS, K>), dim3(256), dim3(256), 32768, 0, out, iters);
    CK(hipDeviceSynchronize());
    unsigned long long t;
    CK(hipMemcpy(&t, out, 8, hipMemcpyDeviceToHost));
    const double per = (double)t / (iters * (double)kAcc);
    printf("row of 12 MFMAs, %s, %d units of %d instructions: %7.3f ticks per MFMA = %.2f x\n",
           READS ? "6 in-place fragment refills (ds_read_b128, counted waits)" : "fragments in registers", UNITS, K, per, per / g_alone);
}

// Issue cost of single instruction kinds: 8 independent copies of one op after every MFMA (in-place ops on 8 registers; a dead or
// shared destination would make hipcc put an s_nop between two asm statements and double the count).
template <int OP>
__global__ __launch_bounds__(256, 1) void op_bench(unsigned long long* out, int iters) {
    f32x16 acc[kAcc], other[4];
#pragma unroll
    for (int i = 0; i < kAcc; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) other[i][r] = (float)(threadIdx.x + r);
    uint32_t x[8], y[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x * 77 + i; y[i] = 0; }
    const bf16x8 a = {1, 1, 1, 1, 1, 1, 1, 1}, b = {1, 1, 1, 1, 1, 1, 1, 1};
#pragma unroll
    for (int i = 0; i < 4; ++i) asm volatile("" : "+a"(other[i]));
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < kAcc; ++j) {
            if (OP < 100) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[j]) : "v"(a), "v"(b));
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (OP == 9) asm volatile("s_nop 0");
                if (OP == 12) asm volatile("v_and_b32 %0, %0, %1" : "+v"(x[i]) : "v"(x[(i + 1) & 7]));
            }
        }
    }
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += (float)y[i];
#pragma unroll
    for (int i = 0; i < kAcc; ++i) s += acc[i][0];
    if (s == 12345.678f) out[1] = 1;
    if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = t1 - t0;
}

template <int OP>
static void run_op(const char* what, unsigned long long* out) {
    const int iters = 2000;
    hipLaunchKernelGGL((op_bench<OP>), dim3(256), dim3(256), 4096, 0, out, iters);
    hipLaunchKernelGGL((op_bench<OP>), dim3(256), dim3(256), 4096, 0, out, iters);
    CK(hipDeviceSynchronize());
    unsigned long long t;
    CK(hipMemcpy(&t, out, 8, hipMemcpyDeviceToHost));
    const double per = (double)t / (iters * (double)kAcc);
    printf("8 independent %-34s after each MFMA: %7.3f ticks per MFMA -> ~%.1f cycles per instruction beyond the MFMA's own 8\n", what, per,
           (per - 8.0) / 8.0);
}

template <int KIND, int K>
static void run(const char* what, unsigned long long* out) {
    const int iters = 2000;
    hipLaunchKernelGGL((bench<KIND, K>), dim3(256), dim3(256), 4096, 0, out, iters);
    hipLaunchKernelGGL((bench<KIND, K>), dim3(256), dim3(256), 4096, 0, out, iters);
    CK(hipDeviceSynchronize());
    unsigned long long t;
    CK(hipMemcpy(&t, out, 8, hipMemcpyDeviceToHost));
    // s_memtime ticks at a fixed 100 MHz: ratios to the bare MFMA stream are what matters
    const double per = (double)t / (iters * (double)kAcc);
    if (KIND == 0 && K == 0) g_alone = per;
    printf("%-44s K = %d: %7.3f memtime ticks per MFMA = %.2f x the MFMA stream alone\n", what, K, per, per / g_alone);
}

int main() {
    unsigned long long* out;
    CK(hipMalloc(&out, 64));
    run<0, 0>("MFMA stream alone", out);
    run<0, 2>("+ K fp32 FMAs on VGPRs after each MFMA", out);
    run<0, 4>("+ K fp32 FMAs on VGPRs after each MFMA", out);
    run<0, 6>("+ K fp32 FMAs on VGPRs after each MFMA", out);
    run<0, 8>("+ K fp32 FMAs on VGPRs after each MFMA", out);
    run<1, 2>("+ K instructions of the epilogue mix", out);
    run<1, 3>("+ K instructions of the epilogue mix", out);
    run<1, 4>("+ K instructions of the epilogue mix", out);
    run<1, 6>("+ K instructions of the epilogue mix", out);
    run<1, 8>("+ K instructions of the epilogue mix", out);
    run<2, 1>("+ K ds_read_b128 (waited) after each MFMA", out);
    run<2, 2>("+ K ds_read_b128 (waited) after each MFMA", out);
    run_op<12>("v_and_b32 (in place)", out);
    run_op<9>("s_nop 0", out);
    run_row<0, 0, 4>(out);
    run_row<1, 0, 4>(out);
    run_row<1, 4, 4>(out);
    run_row<1, 4, 6>(out);
    run_row<1, 8, 3>(out);
    run_row<0, 4, 6>(out);
    run_row<0, 8, 3>(out);
    CK(hipFree(out));
    return 0;
}
