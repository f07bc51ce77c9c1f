// tools/micro/split2_f16.hip -- gate (i) of VERDICT r05 item 1: is an fp32 product taken as THREE v_mfma_f32_32x32x16_f16 terms of two-term
// fp16 operands (x = x_h + x_m; hh + hm + mh, fp32 accumulate) within the bit budget of the fp32 instruction?  Stand-alone, one wave.
//   A. what the matrix pipe does with fp16 SUBNORMAL inputs (the m term of a small value is one) and how it rounds small products
//      against a large accumulator (the questions of tools/ubench/mfma_rounding.hip, for the fp16 instruction)
//   B. error budget: 32 x 32 dot products of length 256 (weights ~ N(0, 1/16), activations = relu(N(0, 1)) -- a hidden layer; and
//      gradients spanning 12 orders of magnitude per column) evaluated by
//        f32     128 x v_mfma_f32_32x32x2_f32                       (the reference instruction)
//        bf16x3  six bf16 terms per 16 k-values                     (the product of rounds 3-5, nnr_split.h)
//        f16x2   three fp16 terms, operands scaled by powers of two (weights: matrix max -> [2^13, 2^14); activations: per column max -> [2^3, 2^4))
//        f16x2r  the same with the residual carried at 2^11: x_m' = fp16((x s - x_h) 2^11) against w_hs = fp16(w s_w 2^-11)
//        f16x2u  activations unscaled, residual at 2^11 (the forward without a per-sample scale)
//      each against the exact (long double) dot product; printed: rms and max of |error| / sum_k |w_k x_k|, in units of 2^-24
//   hipcc --offload-arch=gfx950 -O2 tools/micro/split2_f16.hip -o /tmp/split2_f16 && /tmp/split2_f16
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// ---- A: one MFMA, row 0 of A = a16[k], column n of B = b[k][n], C[0][n] = c[n] ----
__global__ void one_f16(const float* a16, const float* b16x32, const float* c32, float* d32) {
    const int lane = threadIdx.x, h = lane >> 5, n = lane & 31;
    h16x8 a, b;
    for (int i = 0; i < 8; ++i) {
        const int k = 8 * h + i;
        a[i] = (_Float16)(n == 0 ? a16[k] : 0.f);
        b[i] = (_Float16)b16x32[k * 32 + n];
    }
    f32x16 c;
    for (int r = 0; r < 16; ++r) c[r] = 0.f;
    if (h == 0) c[0] = c32[n];
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    if (h == 0) d32[n] = c[0];
}

// ---- B: D[m][n] = sum_k W[m][k] X[k][n], K = 256, M = N = 32; result row m = (r & 3) + 8 (r >> 2) + 4 h of lane (h, n) in register r ----
constexpr int K = 256;
__device__ void store_tile(const f32x16& c, float* d) {
    const int lane = threadIdx.x, h = lane >> 5, n = lane & 31;
    for (int r = 0; r < 16; ++r) d[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + n] = c[r];
}
__global__ void dot_f32(const float* W, const float* X, float* D) {
    const int lane = threadIdx.x, h = lane >> 5, n = lane & 31;
    f32x16 c;
    for (int r = 0; r < 16; ++r) c[r] = 0.f;
    for (int k = 0; k < K; k += 2) c = __builtin_amdgcn_mfma_f32_32x32x2f32(W[n * K + k + h], X[(k + h) * 32 + n], c, 0, 0, 0);
    store_tile(c, D);
}
__global__ void dot_bf16x3(const float* W, const float* X, float* D) {
    const int lane = threadIdx.x, h = lane >> 5, n = lane & 31;
    f32x16 c;
    for (int r = 0; r < 16; ++r) c[r] = 0.f;
    for (int k0 = 0; k0 < K; k0 += 16) {
        bf16x8 w[3], x[3];
        for (int i = 0; i < 8; ++i) {
            float a = W[n * K + k0 + 8 * h + i], b = X[(k0 + 8 * h + i) * 32 + n];
            for (int t = 2; t >= 0; --t) {      // [2] = h, [1] = m, [0] = l
                w[t][i] = (__bf16)a; a -= (float)w[t][i];
                x[t][i] = (__bf16)b; b -= (float)x[t][i];
            }
        }
        const int wt[6] = {0, 1, 1, 2, 2, 2}, xt[6] = {2, 1, 2, 0, 1, 2};      // the order of nnr_split.h: small products first
        for (int t = 0; t < 6; ++t) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[wt[t]], x[xt[t]], c, 0, 0, 0);
    }
    store_tile(c, D);
}
// MODE 0: plain residual, per-column scale; 1: residual at 2^11, per-column scale; 2: residual at 2^11, no activation scale
template <int MODE>
__global__ void dot_f16x2(const float* W, const float* X, float* D, float sw, const float* sx) {
    const int lane = threadIdx.x, h = lane >> 5, n = lane & 31;
    const float s = MODE == 2 ? 1.f : sx[n];
    const float up = MODE == 0 ? 1.f : 2048.f;
    f32x16 c;
    for (int r = 0; r < 16; ++r) c[r] = 0.f;
    for (int k0 = 0; k0 < K; k0 += 16) {
        h16x8 wh, wm, whs, xh, xm;
        for (int i = 0; i < 8; ++i) {
            const float a = W[n * K + k0 + 8 * h + i] * sw, b = X[(k0 + 8 * h + i) * 32 + n] * s;
            wh[i] = (_Float16)a; wm[i] = (_Float16)(a - (float)wh[i]); whs[i] = (_Float16)(a / up);
            xh[i] = (_Float16)b; xm[i] = (_Float16)((b - (float)xh[i]) * up);
        }
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(wm, xh, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(whs, xm, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh, c, 0, 0, 0);
    }
    const float inv = 1.f / (sw * s);
    for (int r = 0; r < 16; ++r) c[r] *= inv;
    store_tile(c, D);
}

static float pow2_to(float m, int top) {      // power of two s with m s in [2^(top-1), 2^top)
    int e;
    frexpf(m, &e);      // m = f 2^e, f in [0.5, 1)
    return ldexpf(1.f, top - e);
}

int main() {
    // ---------------- A ----------------
    struct Q { const char* what; double c; std::vector<double> a, b; };
    const double u = ldexp(1.0, -23);
    std::vector<Q> qs = {
        {"a = 2^-20 (fp16 subnormal) x b = 2^10: 2^-10 if subnormal inputs are honoured, 0 if flushed", 0.0, {ldexp(1.0, -20)}, {1024.0}},
        {"a = 2^-24 (smallest subnormal) x b = 2^14", 0.0, {ldexp(1.0, -24)}, {16384.0}},
        {"a = 3 x 2^-24 x b = 2^-24 (product 3 x 2^-48, far below fp16's range but not fp32's)", 0.0, {3 * ldexp(1.0, -24)}, {ldexp(1.0, -24)}},
        {"c = 1, one product 0.75 ulp", 1.0, {1.0}, {0.75 * u}},
        {"c = 1, one product 0.5 ulp (tie)", 1.0, {1.0}, {0.5 * u}},
        {"c = 1, 16 products of 1/16 ulp (sum 1 ulp)", 1.0, std::vector<double>(16, 1.0), std::vector<double>(16, u / 16)},
        {"c = 1, 8 products +3/16 ulp and 8 of -1/16 ulp (sum 1 ulp)", 1.0, std::vector<double>(16, 1.0),
         [&] { std::vector<double> v(16, -u / 16); for (int i = 0; i < 8; ++i) v[i] = 3 * u / 16; return v; }()},
        {"c = 0, products 1, -1 and 2^-24 (what is left after a cancellation)", 0.0, {1.0, 1.0, 1.0}, {1.0, -1.0, ldexp(1.0, -24)}},
        {"c = 0, 16 products (1 + 2^-10)^2: exact 22-bit products summed", 0.0, std::vector<double>(16, 1.0 + ldexp(1.0, -10)),
         std::vector<double>(16, 1.0 + ldexp(1.0, -10))},
        {"c = 0, 65504 x 65504 (the largest finite fp16 squared)", 0.0, {65504.0}, {65504.0}},
    };
    {
        const int NC = (int)qs.size();
        std::vector<float> a(16 * 32, 0.f), b(16 * 32, 0.f), c(32, 0.f), d(32);
        // A differs per question here, so the questions run one launch each
        float *da, *db, *dc, *dd;
        CK(hipMalloc(&da, 64)); CK(hipMalloc(&db, 16 * 32 * 4)); CK(hipMalloc(&dc, 128)); CK(hipMalloc(&dd, 128));
        printf("A. v_mfma_f32_32x32x16_f16, one instruction per question (u = 2^-23)\n");
        for (int q = 0; q < NC; ++q) {
            std::vector<float> a16(16, 0.f), bb(16 * 32, 0.f), cc(32, 0.f);
            long double exact = qs[q].c;
            for (size_t k = 0; k < qs[q].a.size(); ++k) {
                a16[k] = (float)qs[q].a[k];
                bb[k * 32] = (float)qs[q].b[k];
                exact += (long double)(float)(_Float16)a16[k] * (long double)(float)(_Float16)bb[k * 32];
            }
            cc[0] = (float)qs[q].c;
            CK(hipMemcpy(da, a16.data(), 64, hipMemcpyHostToDevice));
            CK(hipMemcpy(db, bb.data(), 16 * 32 * 4, hipMemcpyHostToDevice));
            CK(hipMemcpy(dc, cc.data(), 128, hipMemcpyHostToDevice));
            hipLaunchKernelGGL(one_f16, dim3(1), dim3(64), 0, 0, da, db, dc, dd);
            CK(hipMemcpy(d.data(), dd, 128, hipMemcpyDeviceToHost));
            printf("  %-95s got %.9g (= c %+.4f u)   exactly rounded %.9g\n", qs[q].what, d[0], (d[0] - qs[q].c) / u, (double)(float)exact);
        }
    }
    // ---------------- B ----------------
    std::mt19937 rng(12345);
    std::normal_distribution<float> nd(0.f, 1.f);
    for (int kind = 0; kind < 3; ++kind) {
        // 0: hidden layer (weights N(0, 1/16), activations relu(N(0,1)) with a sixth of them tiny); 1: gradients, column n scaled by 10^(-12 n / 31);
        // 2: a "dying" layer: every activation ~1e-4
        std::vector<float> W(32 * K), X(K * 32);
        for (auto& w : W) w = 0.0625f * nd(rng);
        for (int k = 0; k < K; ++k)
            for (int n = 0; n < 32; ++n) {
                float v = nd(rng);
                if (kind == 0) v = v > 0 ? v * ((k % 6) == 0 ? 1e-3f : 1.f) : 0.f;
                if (kind == 1) v *= powf(10.f, -12.f * n / 31.f);
                if (kind == 2) v = fabsf(v) * 1e-4f;
                X[k * 32 + n] = v;
            }
        float wmax = 0.f;
        for (float w : W) wmax = fmaxf(wmax, fabsf(w));
        std::vector<float> sx(32);
        for (int n = 0; n < 32; ++n) {
            float m = 1e-30f;
            for (int k = 0; k < K; ++k) m = fmaxf(m, fabsf(X[k * 32 + n]));
            sx[n] = pow2_to(m, 4);
        }
        const float sw = pow2_to(wmax, 14);
        float *dW, *dX, *dD, *dS;
        CK(hipMalloc(&dW, W.size() * 4)); CK(hipMalloc(&dX, X.size() * 4)); CK(hipMalloc(&dD, 32 * 32 * 4)); CK(hipMalloc(&dS, 128));
        CK(hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dS, sx.data(), 128, hipMemcpyHostToDevice));
        std::vector<long double> exact(32 * 32), mag(32 * 32);
        for (int m = 0; m < 32; ++m)
            for (int n = 0; n < 32; ++n) {
                long double s = 0, a = 0;
                for (int k = 0; k < K; ++k) {
                    const long double p = (long double)W[m * K + k] * (long double)X[k * 32 + n];
                    s += p; a += fabsl(p);
                }
                exact[m * 32 + n] = s; mag[m * 32 + n] = a;
            }
        const char* names[5] = {"f32", "bf16x3", "f16x2", "f16x2r", "f16x2u"};
        printf("B%d. %s: |error| / sum |w x| in units of 2^-24 (rms, max over 1024 dot products of length 256)\n", kind,
               kind == 0 ? "hidden layer (relu activations, a sixth of them ~1e-3)" : kind == 1 ? "gradients, columns spanning 12 decades (per-column scale)" : "activations all ~1e-4");
        for (int v = 0; v < 5; ++v) {
            if (v == 0) hipLaunchKernelGGL(dot_f32, dim3(1), dim3(64), 0, 0, dW, dX, dD);
            if (v == 1) hipLaunchKernelGGL(dot_bf16x3, dim3(1), dim3(64), 0, 0, dW, dX, dD);
            if (v == 2) hipLaunchKernelGGL(dot_f16x2<0>, dim3(1), dim3(64), 0, 0, dW, dX, dD, sw, dS);
            if (v == 3) hipLaunchKernelGGL(dot_f16x2<1>, dim3(1), dim3(64), 0, 0, dW, dX, dD, sw, dS);
            if (v == 4) hipLaunchKernelGGL(dot_f16x2<2>, dim3(1), dim3(64), 0, 0, dW, dX, dD, sw, dS);
            std::vector<float> D(32 * 32);
            CK(hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost));
            double ss = 0, mx = 0;
            int cnt = 0;
            for (int i = 0; i < 32 * 32; ++i) {
                if (mag[i] == 0) continue;
                const double e = (double)(fabsl((long double)D[i] - exact[i]) / mag[i]) * ldexp(1.0, 24);
                ss += e * e; mx = fmax(mx, e); ++cnt;
            }
            printf("  %-7s rms %8.3f   max %8.3f\n", names[v], sqrt(ss / cnt), mx);
        }
        hipFree(dW); hipFree(dX); hipFree(dD); hipFree(dS);
    }
    return 0;
}
