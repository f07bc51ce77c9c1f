// LDS read throughput of one CU: 4 waves (one per SIMD), each streaming 32 KiB of LDS through ds_read_b128 / ds_read_b64 /
// ds_read2_b64 with 8 independent reads in flight.  Prints bytes per clock per CU.   hipcc -O3 --offload-arch=gfx950 lds_read.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k(float* out, unsigned long long* cyc, int iters) {
    __shared__ __attribute__((aligned(16))) f32x4 lds[8192];   // 128 KiB
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = f32x4{1.f, 2.f, 3.f, 4.f};
    __syncthreads();
    const int lane = threadIdx.x & 63;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
    const unsigned a128 = base + lane * 16, a64 = base + lane * 8, a32 = base + lane * 4;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        // 32 KiB per wave and iteration, every wave the SAME addresses (like the weight panels); asm volatile: really issued
#define RD128(j) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r[(j) & 7]) : "v"(a128), "n"((j) * 1024));
#define RD64(j) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(q[(j) & 7]) : "v"(a64), "n"((j) * 512));
#define RD32(j) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(w[(j) & 7]) : "v"(a32), "n"((j) * 256));
#define R8(M, j) M(j) M(j + 1) M(j + 2) M(j + 3) M(j + 4) M(j + 5) M(j + 6) M(j + 7)
        f32x4 r[8]; f32x2 q[8]; float w[8];
        if (MODE == 0) { R8(RD128, 0) R8(RD128, 8) R8(RD128, 16) R8(RD128, 24) asm volatile("s_waitcnt lgkmcnt(0)"); acc += r[0] + r[7]; }
        else if (MODE == 1) { R8(RD64, 0) R8(RD64, 8) R8(RD64, 16) R8(RD64, 24) R8(RD64, 32) R8(RD64, 40) R8(RD64, 48) R8(RD64, 56) asm volatile("s_waitcnt lgkmcnt(0)"); acc[0] += q[0][0] + q[7][1]; }
        else { R8(RD32, 0) R8(RD32, 8) R8(RD32, 16) R8(RD32, 24) R8(RD32, 32) R8(RD32, 40) R8(RD32, 48) R8(RD32, 56) asm volatile("s_waitcnt lgkmcnt(0)"); acc[0] += w[0] + w[7]; }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.f) out[0] = 1.f;
}

template <int MODE, int WAVES>
int run(const char* what, float* out, unsigned long long* cyc) {
    const int iters = 2000;
    hipLaunchKernelGGL((k<MODE, WAVES>), dim3(256), dim3(64 * WAVES), 0, 0, out, cyc, iters);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<MODE, WAVES>), dim3(256), dim3(64 * WAVES), 0, 0, out, cyc, iters);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long c;
    CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
    const double per_iter = MODE == 0 ? 32768 : MODE == 1 ? 32768 : 16384;
    printf("%-28s %d waves: %.1f bytes / tick / CU  (%llu s_memtime ticks, kernel %.3f ms => %.0f MHz tick rate; %.1f GB/s per CU)\n", what, WAVES, (double)WAVES * iters * per_iter / c, c, ms, c / ms / 1e3, (double)WAVES * iters * per_iter / ms / 1e6);
    return 0;
}

int main() {
    float* out; unsigned long long* cyc;
    CK(hipMalloc(&out, 64)); CK(hipMalloc(&cyc, 64));
    run<0, 4>("ds_read_b128", out, cyc); run<1, 4>("ds_read_b64 x2", out, cyc); run<2, 4>("ds_read_b32 x4", out, cyc);
    run<0, 8>("ds_read_b128", out, cyc); run<1, 8>("ds_read_b64 x2", out, cyc);
    run<0, 1>("ds_read_b128", out, cyc);
    return 0;
}
