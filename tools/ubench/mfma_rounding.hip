// tools/ubench/mfma_rounding.hip -- how does the matrix pipe round?  One wave, one v_mfma_f32_32x32x16_bf16 (and one
// v_mfma_f32_32x32x2_f32): row 0 of A against 32 different columns of B and values of C, each column a question about alignment /
// truncation of the products against the accumulator.  Prints every D[0][n] next to the exactly rounded (RNE) answer.
// Experiment only (the design of csrc/nnr_split.h rests on its answers: profiles/r03/n_mfma_rounding.txt) -- not part of libnnr.so.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/mfma_rounding.hip -o tools/ubench/build/mfma_rounding
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void mfma_bf16_kernel(const float* a16, const float* b16x32, const float* c32, float* d32) {
    const int lane = threadIdx.x, h = lane >> 5, n = lane & 31;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) {
        const int k = 8 * h + i;
        a[i] = (__bf16)(n == 0 ? a16[k] : 0.f);      // A[m = n][k]: only row 0 is non-zero
        b[i] = (__bf16)b16x32[k * 32 + n];           // B[k][n]
    }
    f32x16 c;
    for (int r = 0; r < 16; ++r) c[r] = 0.f;
    if (h == 0) c[0] = c32[n];                       // C[0][n] sits in register 0 of lanes 0..31
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    if (h == 0) d32[n] = c[0];
}

__global__ void mfma_f32_kernel(const float* a2, const float* b2x32, const float* c32, float* d32) {
    const int lane = threadIdx.x, h = lane >> 5, n = lane & 31;
    const float a = n == 0 ? a2[h] : 0.f, b = b2x32[h * 32 + n];
    f32x16 c;
    for (int r = 0; r < 16; ++r) c[r] = 0.f;
    if (h == 0) c[0] = c32[n];
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
    if (h == 0) d32[n] = c[0];
}

struct Case { const char* what; double c; std::vector<double> prod; };   // products a_k * b_k with a_k = 1 (so b_k = the product)

int main() {
    const double u = std::ldexp(1.0, -23);   // ulp of 1.0
    std::vector<Case> cases = {
        {"c=1, one product 0.75 ulp", 1.0, {0.75 * u}},
        {"c=1, one product 0.5 ulp (tie)", 1.0, {0.5 * u}},
        {"c=1, one product 0.25 ulp", 1.0, {0.25 * u}},
        {"c=1, 16 products of 1/4 ulp (sum 4 ulp)", 1.0, std::vector<double>(16, 0.25 * u)},
        {"c=1, 16 products of 1/16 ulp (sum 1 ulp)", 1.0, std::vector<double>(16, u / 16)},
        {"c=1, 12 products of 1/16 ulp (sum 0.75 ulp)", 1.0, std::vector<double>(12, u / 16)},
        {"c=1, 16 products of 1/64 ulp (sum 0.25 ulp)", 1.0, std::vector<double>(16, u / 64)},
        {"c=1, one product -0.25 ulp(1) = -0.5 ulp below", 1.0, {-0.25 * u}},
        {"c=1, one product -0.375 ulp", 1.0, {-0.375 * u}},
        {"c=0, products 1 and 15 x 1/16 ulp", 0.0, [&] { std::vector<double> v(16, u / 16); v[0] = 1.0; return v; }()},
        {"c=0, products 1 and 0.75 ulp", 0.0, {1.0, 0.75 * u}},
        {"c=0, products 1, -1 and 2^-30", 0.0, {1.0, -1.0, std::ldexp(1.0, -30)}},
        {"c=2^24, 16 products of 1 (each below the ulp 2)", std::ldexp(1.0, 24), std::vector<double>(16, 1.0)},
        {"c=2^20, 16 products of 1 + 2^-7", std::ldexp(1.0, 20), std::vector<double>(16, 1.0 + std::ldexp(1.0, -7))},
        {"c=1, 8 products +3/16 ulp and 8 of -1/16 ulp (sum 1 ulp)", 1.0, [&] { std::vector<double> v(16, -u / 16); for (int i = 0; i < 8; ++i) v[i] = 3 * u / 16; return v; }()},
        {"c=2^-16, products 1 and -1 (cancel), then c survives?", std::ldexp(1.0, -16), {1.0, -1.0}},
        {"c=1+ulp, 16 products of 2^-8 (large-ish)", 1.0 + u, std::vector<double>(16, std::ldexp(1.0, -8) * (1 + std::ldexp(1.0, -7)))},
    };
    const int NC = (int)cases.size();
    std::vector<float> a(16, 1.f), b(16 * 32, 0.f), c(32, 0.f), d(32), d2(32);
    for (int n = 0; n < NC; ++n) {
        c[n] = (float)cases[n].c;
        for (size_t k = 0; k < cases[n].prod.size(); ++k) b[k * 32 + n] = (float)cases[n].prod[k];
    }
    float *da, *db, *dc, *dd;
    hipMalloc(&da, 64); hipMalloc(&db, 16 * 32 * 4); hipMalloc(&dc, 128); hipMalloc(&dd, 128);
    hipMemcpy(da, a.data(), 64, hipMemcpyHostToDevice);
    hipMemcpy(db, b.data(), 16 * 32 * 4, hipMemcpyHostToDevice);
    hipMemcpy(dc, c.data(), 128, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(mfma_bf16_kernel, dim3(1), dim3(64), 0, 0, da, db, dc, dd);
    hipMemcpy(d.data(), dd, 128, hipMemcpyDeviceToHost);
    printf("v_mfma_f32_32x32x16_bf16  (u = ulp(1) = 2^-23; 'exact' = the exact sum rounded to nearest even)\n");
    for (int n = 0; n < NC; ++n) {
        long double s = cases[n].c;
        for (double p : cases[n].prod) s += (long double)(float)p;     // every product here is a bf16-exact value
        const float exact = (float)s;
        printf("  %-62s got %.9g (%a)  exact %.9g (%a)  %s\n", cases[n].what, d[n], d[n], exact, exact, d[n] == exact ? "==" : "DIFFERS");
    }
    // the fp32 instruction: two products per MFMA
    std::vector<Case> c2 = {
        {"c=1, products 0.375 ulp + 0.375 ulp (sum 0.75)", 1.0, {0.375 * u, 0.375 * u}},
        {"c=1, one product 0.75 ulp", 1.0, {0.75 * u}},
        {"c=2^24, products 1 + 1", std::ldexp(1.0, 24), {1.0, 1.0}},
        {"c=1, one product -0.375 ulp", 1.0, {-0.375 * u}},
        {"c=0, products 1 and 0.75 ulp", 0.0, {1.0, 0.75 * u}},
        {"c=1, product (1+2^-12)^2 - 1 scaled: a=b=1+2^-12 times 2^-12", 1.0, {}},
    };
    std::vector<float> a2(2, 1.f), b2(2 * 32, 0.f), cc(32, 0.f);
    for (size_t n = 0; n < c2.size(); ++n) {
        cc[n] = (float)c2[n].c;
        for (size_t k = 0; k < c2[n].prod.size(); ++k) b2[k * 32 + n] = (float)c2[n].prod[k];
    }
    hipMemcpy(da, a2.data(), 8, hipMemcpyHostToDevice);
    hipMemcpy(db, b2.data(), 2 * 32 * 4, hipMemcpyHostToDevice);
    hipMemcpy(dc, cc.data(), 128, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(mfma_f32_kernel, dim3(1), dim3(64), 0, 0, da, db, dc, dd);
    hipMemcpy(d2.data(), dd, 128, hipMemcpyDeviceToHost);
    printf("v_mfma_f32_32x32x2_f32\n");
    for (size_t n = 0; n + 1 < c2.size(); ++n) {
        long double s = c2[n].c;
        for (double p : c2[n].prod) s += (long double)(float)p;
        const float exact = (float)s;
        printf("  %-62s got %.9g (%a)  exact %.9g (%a)  %s\n", c2[n].what, d2[n], d2[n], exact, exact, d2[n] == exact ? "==" : "DIFFERS");
    }
    return 0;
}
