// tools/ubench/adam_variants.hip -- which arithmetic does torch.optim.Adam(fused=True) carry out on this build?  One element-wise
// Adam update with the expression of each quantity selectable (va: first moment, vb: second moment, vc: parameter update); the driver
// (tools/adam_variants.py) compares every variant BITWISE with torch's own kernel.  Experiment only -- not part of libnnr.so.
#include <hip/hip_runtime.h>
#include <stdint.h>
#pragma clang diagnostic ignored "-Wunused-value"

__global__ void adam_variant_kernel(float* p, const float* g, float* m, float* v, int64_t n, double lr, double beta1, double beta2,
                                    double eps, float step, int va, int vb, int vc) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    float param = p[e];
    const float grad = g[e];
    float exp_avg = m[e], exp_avg_sq = v[e];
    const float b1 = (float)beta1, b2 = (float)beta2;
    switch (va) {
        case 0: exp_avg = beta1 * exp_avg + (1 - beta1) * grad; break;                                   // double, contraction allowed
        case 1: exp_avg = b1 * exp_avg + (1 - b1) * grad; break;                                          // float, contraction allowed
        case 2: exp_avg = exp_avg + (1 - b1) * (grad - exp_avg); break;                                   // float lerp
        case 3: exp_avg = (float)__dadd_rn(__dmul_rn(beta1, (double)exp_avg), __dmul_rn(1 - beta1, (double)grad)); break;   // double, no fma
        case 4: exp_avg = __fadd_rn(__fmul_rn(b1, exp_avg), __fmul_rn(1.f - b1, grad)); break;           // float, no fma
        case 5: exp_avg = exp_avg + (float)(1 - beta1) * (grad - exp_avg); break;                         // float lerp, weight from double
        case 6: exp_avg = (float)(exp_avg + (1 - beta1) * ((double)grad - exp_avg)); break;               // double lerp
        case 7: exp_avg = __builtin_fmaf((float)(1 - beta1), grad - exp_avg, exp_avg); break;             // float lerp as one fma
        case 8: {   // at::lerp's two-sided formula
            const float w = (float)(1 - beta1), d = grad - exp_avg;
            exp_avg = w < 0.5f ? exp_avg + w * d : grad - d * (1.f - w);
        } break;
        case 9: exp_avg = __fadd_rn(exp_avg, __fmul_rn((float)(1 - beta1), __fsub_rn(grad, exp_avg))); break;   // float lerp, no fma
        case 10: exp_avg = exp_avg + (1 - beta1) * (grad - exp_avg); break;                               // lerp: float difference, double weight
        case 11: exp_avg = (float)__dadd_rn((double)exp_avg, __dmul_rn(1 - beta1, (double)(grad - exp_avg))); break;   // the same, no fma
        case 12: exp_avg = (float)__fma_rn(1 - beta1, (double)(grad - exp_avg), (double)exp_avg); break;   // the same, one fma
        case 14: exp_avg = __builtin_fmaf((float)(1 - beta1), grad, __builtin_fmaf(-(float)(1 - beta1), exp_avg, exp_avg)); break;   // "faster lerp", float
        case 15: exp_avg = (float)__fma_rn(1 - beta1, (double)grad, __fma_rn(-(1 - beta1), (double)exp_avg, (double)exp_avg)); break;   // "faster lerp", double
        case 16: exp_avg = (float)__fma_rn(1 - beta1, (double)grad - (double)exp_avg, (double)exp_avg); break;   // double lerp, one fma
        case 17: exp_avg = (float)__dadd_rn((double)exp_avg, __dmul_rn(1 - beta1, __dsub_rn((double)grad, (double)exp_avg))); break;   // double lerp, no fma
        case 18: exp_avg = (float)__fma_rn(beta1, (double)exp_avg, __dmul_rn(1 - beta1, (double)grad)); break;    // beta form, fma on the first product
        case 19: exp_avg = (float)__fma_rn(1 - beta1, (double)grad, __dmul_rn(beta1, (double)exp_avg)); break;    // beta form, fma on the second product
        case 13: {   // at::lerp with a double weight: weight < 0.5 branch
            const double w = 1 - beta1;
            const float d = grad - exp_avg;
            exp_avg = w < 0.5 ? (float)(exp_avg + w * d) : (float)(grad - d * (1 - w));
        } break;
    }
    switch (vb) {
        case 0: exp_avg_sq = beta2 * exp_avg_sq + (1 - beta2) * grad * grad; break;                      // double
        case 1: exp_avg_sq = b2 * exp_avg_sq + (1 - b2) * grad * grad; break;                             // float
        case 2: exp_avg_sq = (float)__dadd_rn(__dmul_rn(beta2, (double)exp_avg_sq), __dmul_rn(__dmul_rn(1 - beta2, (double)grad), (double)grad)); break;
        case 3: exp_avg_sq = __fadd_rn(__fmul_rn(b2, exp_avg_sq), __fmul_rn(__fmul_rn(1.f - b2, grad), grad)); break;
        case 4: exp_avg_sq = b2 * exp_avg_sq + (float)(1 - beta2) * grad * grad; break;
        case 5: exp_avg_sq = __builtin_fmaf(b2, exp_avg_sq, (float)(1 - beta2) * grad * grad); break;
        case 6: exp_avg_sq = __builtin_fmaf((float)(1 - beta2) * grad, grad, b2 * exp_avg_sq); break;
        case 7: exp_avg_sq = __fadd_rn(__fmul_rn(b2, exp_avg_sq), __fmul_rn((float)(1 - beta2), __fmul_rn(grad, grad))); break;
        case 8: exp_avg_sq = (float)__fma_rn(beta2, (double)exp_avg_sq, (1 - beta2) * (double)grad * (double)grad); break;   // double, fma on the decayed moment
    }
    float bc1, bc2s;
    if (vc & 8) {      // bias corrections in float
        bc1 = 1.f - powf(b1, step);
        bc2s = sqrtf(1.f - powf(b2, step));
    } else {
        bc1 = (float)(1 - pow(beta1, (double)step));
        bc2s = (float)sqrt(1 - pow(beta2, (double)step));
    }
    switch (vc & 7) {
        case 0: {   // torch's adam_math as I remember it
            const float step_size = lr / bc1;
            const float denom = (sqrtf(exp_avg_sq) / bc2s) + eps;
            param -= step_size * exp_avg / denom;
        } break;
        case 1: {   // all float
            const float step_size = (float)lr / bc1;
            const float denom = (sqrtf(exp_avg_sq) / bc2s) + (float)eps;
            param -= step_size * exp_avg / denom;
        } break;
        case 2: {   // double where doubles appear
            const double step_size = lr / bc1;
            const double denom = (sqrtf(exp_avg_sq) / bc2s) + eps;
            param = (float)(param - step_size * exp_avg / denom);
        } break;
        case 3: {   // addcdiv form: param += (-step_size) * (exp_avg / denom)
            const float step_size = lr / bc1;
            const float denom = (sqrtf(exp_avg_sq) / bc2s) + eps;
            param = param + (-step_size) * (exp_avg / denom);
        } break;
        case 4: {
            const float step_size = lr / bc1;
            const float denom = (sqrtf(exp_avg_sq) / bc2s) + eps;
            param = __fsub_rn(param, __fdiv_rn(__fmul_rn(step_size, exp_avg), denom));
        } break;
    }
    p[e] = param;
    m[e] = exp_avg;
    v[e] = exp_avg_sq;
}

extern "C" int adam_variant(float* p, const float* g, float* m, float* v, int64_t n, double lr, double beta1, double beta2, double eps,
                            float step, int va, int vb, int vc, void* stream) {
    hipLaunchKernelGGL(adam_variant_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr, beta1,
                       beta2, eps, step, va, vb, vc);
    return (int)hipGetLastError();
}
