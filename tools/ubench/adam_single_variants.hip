// tools/ubench/adam_single_variants.hip -- which arithmetic does torch.optim.Adam(foreach=False, fused=False) -- torch's single-tensor
// implementation (torch/optim/adam.py::_single_tensor_adam; the reference's train.py:58,99,117,140 builds plain Adam objects, whose update
// this is on one tensor at a time) -- carry out on this build?  The update is a chain of ATen kernels (lerp_, mul_, addcmul_, sqrt, div by
// a host scalar, add_, addcdiv_); the expression of each link is selectable here and tools/adam_single_variants.py compares every
// candidate BITWISE with torch.  Experiment only -- not part of libnnr.so.
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ void adam_single_kernel(float* p, const float* g, float* m, float* v, int64_t n, double beta1, double beta2, double eps,
                                   double neg_step_size, double bc2_sqrt, int va, int vb, int vc, int vd) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    float param = p[e];
    const float grad = g[e];
    float exp_avg = m[e], exp_avg_sq = v[e];
    const float w = (float)(1 - beta1), d = __fsub_rn(grad, exp_avg);
    switch (va) {      // exp_avg.lerp_(grad, 1 - beta1): weight < 0.5 -> self + weight * (end - self)
        case 0: exp_avg = __builtin_fmaf(w, d, exp_avg); break;
        case 1: exp_avg = __fadd_rn(exp_avg, __fmul_rn(w, d)); break;
        case 2: exp_avg = __fadd_rn(__fmul_rn((float)beta1, exp_avg), __fmul_rn(w, grad)); break;      // pytorch 1.7: mul_(beta1).add_(grad, alpha = 1 - beta1), no fma
        case 3: exp_avg = __builtin_fmaf(w, grad, __fmul_rn((float)beta1, exp_avg)); break;            // the same with the add_ contracted
    }
    const float v1 = __fmul_rn(exp_avg_sq, (float)beta2), a2 = (float)(1 - beta2);
    switch (vb) {      // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value = 1 - beta2): a + alpha * (b * c)
        case 0: exp_avg_sq = __builtin_fmaf(a2, __fmul_rn(grad, grad), v1); break;
        case 1: exp_avg_sq = __fadd_rn(v1, __fmul_rn(a2, __fmul_rn(grad, grad))); break;
        case 2: exp_avg_sq = __builtin_fmaf(__fmul_rn(a2, grad), grad, v1); break;
        case 3: exp_avg_sq = __fadd_rn(v1, __fmul_rn(__fmul_rn(a2, grad), grad)); break;
    }
    float denom;
    const float sq = sqrtf(exp_avg_sq);
    switch (vc) {      // (exp_avg_sq.sqrt() / bias_correction2_sqrt).add_(eps): division by a host scalar
        case 0: denom = __fadd_rn(__fmul_rn(sq, 1.0f / (float)bc2_sqrt), (float)eps); break;      // a * (1 / b), the reciprocal in float
        case 1: denom = __fadd_rn(__fdiv_rn(sq, (float)bc2_sqrt), (float)eps); break;
        case 2: denom = __fadd_rn(__fmul_rn(sq, (float)(1.0 / bc2_sqrt)), (float)eps); break;     // the reciprocal in double
        default: denom = __fadd_rn(__fmul_rn(__fsqrt_rn(exp_avg_sq), 1.0f / (float)bc2_sqrt), (float)eps); break;
    }
    const float ss = (float)neg_step_size, q = __fdiv_rn(exp_avg, denom);
    switch (vd) {      // param.addcdiv_(exp_avg, denom, value = -step_size): a + alpha * (b / c)
        case 0: param = __builtin_fmaf(ss, q, param); break;
        case 1: param = __fadd_rn(param, __fmul_rn(ss, q)); break;
    }
    p[e] = param;
    m[e] = exp_avg;
    v[e] = exp_avg_sq;
}

extern "C" int adam_single(float* p, const float* g, float* m, float* v, int64_t n, double beta1, double beta2, double eps,
                           double neg_step_size, double bc2_sqrt, int va, int vb, int vc, int vd, void* stream) {
    hipLaunchKernelGGL(adam_single_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, beta1, beta2,
                       eps, neg_step_size, bc2_sqrt, va, vb, vc, vd);
    return (int)hipGetLastError();
}
