// nnr_optim.hip -- ONE launch for the Adam updates of a training step (reference model/training.py:90-96: up to four
// torch.optim.Adam.step() calls per iteration).  The trainer already selects torch's single-kernel implementation of that class
// (fused=True: 2 launches per optimiser -- the step-counter increment and the update -- 6 per step for the network, pose and
// distortion optimisers, 69 us of GPU time at 595 844 + 128 parameters); this kernel takes the tensors of ALL of them at once.
//
// The update is torch's, operation by operation and type by type (aten/src/ATen/native/cuda/fused_adam_utils.cuh, the non-amsgrad,
// non-capturable ADAM_MODE::ORIGINAL path, weight_decay == 0, maximize == false), because training must not depend on which of the
// two implementations stepped: lr, betas and eps are DOUBLES, the first / second moment updates are evaluated in double (one fma on the
// decayed moment) and rounded to float once, the bias corrections are 1 - pow(beta, step) in double rounded to float, the step size lr / bias_correction1 is a
// double division rounded to float, the denominator adds the double eps to a float quotient.  tests/test_gpu_optim.py compares the
// parameters and both moments BITWISE with torch.optim.Adam(fused=True) over hundreds of steps.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/nnr.h"
#include "nnr_kernels.h"

namespace nnr {

constexpr int kAdamBlock = 256, kAdamIlp = 4;   // elements per block = 1024
static_assert(sizeof(nnr_adam_table) <= 4096, "the table is passed by value: a kernel argument block holds 4 KiB");

__global__ __launch_bounds__(kAdamBlock) void adam_multi_kernel(nnr_adam_table t) {
    // which tensor does this block work on?  (block_first is a prefix table: tensor i owns blocks [block_first[i], block_first[i+1]))
    int i = 0;
    while (i + 1 < t.n_tensors && (int)blockIdx.x >= t.block_first[i + 1]) ++i;
    const int64_t n = t.numel[i];
    const int64_t base = (int64_t)((int)blockIdx.x - t.block_first[i]) * (kAdamBlock * kAdamIlp);
    float* __restrict__ p = t.param[i];
    const float* __restrict__ g = t.grad[i];
    float* __restrict__ m = t.exp_avg[i];
    float* __restrict__ v = t.exp_avg_sq[i];
    const double lr = t.lr[i], beta1 = t.beta1[i], beta2 = t.beta2[i], eps = t.eps[i];
    // torch increments the step counter first (_foreach_add_(state_steps, 1)) and the update reads the incremented value.  Here the
    // counters ping-pong between two arrays so that no block can read a counter another block has already advanced.
    const float step = t.step_in[i][0] + 1.0f;
    if (base == 0 && threadIdx.x == 0) t.step_out[i][0] = step;
    if (t.flavour == NNR_ADAM_SINGLE) {
        // torch/optim/adam.py::_single_tensor_adam on this build, link by link (tools/adam_single_variants.py compares every candidate
        // expression bitwise with torch's kernels, profiles/r05/b_adam_single_variants.txt): exp_avg.lerp_(grad, 1 - beta1) = one float fma
        // on the difference; exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value = 1 - beta2) = a float product and one fma; the denominator
        // divides by the HOST scalar bias_correction2 ** 0.5 through its reciprocal (taken in double, rounded to float), then adds float(eps)
        // -- two kernels in torch, so the product and the sum must NOT contract --; param.addcdiv_(exp_avg, denom, value = -step_size) = one
        // fma on the quotient.  The host scalars come from the caller as python computes them.
        const float w1 = (float)(1 - beta1), b2f = (float)beta2, w2 = (float)(1 - beta2), epsf = (float)eps;
        const float inv_bc2 = (float)(1.0 / t.bc2_sqrt[i]), nss = (float)lr;      // (this flavour's lr[i] = -(lr / bias_correction1), see nnr.h)
#pragma unroll
        for (int u = 0; u < kAdamIlp; ++u) {
            const int64_t e = base + (int64_t)u * kAdamBlock + threadIdx.x;
            if (e >= n) continue;
            const float grad = g[e];
            float exp_avg = m[e], exp_avg_sq = v[e];
            exp_avg = __builtin_fmaf(w1, grad - exp_avg, exp_avg);
            float decayed = exp_avg_sq * b2f, gg = grad * grad;
            asm volatile("" : "+v"(decayed), "+v"(gg));          // (products of their own kernels: rounded before the fma)
            exp_avg_sq = __builtin_fmaf(w2, gg, decayed);
            float scaled = sqrtf(exp_avg_sq) * inv_bc2;
            asm volatile("" : "+v"(scaled));
            const float denom = scaled + epsf;
            float quot = exp_avg / denom;
            asm volatile("" : "+v"(quot));
            p[e] = __builtin_fmaf(nss, quot, p[e]);
            m[e] = exp_avg;
            v[e] = exp_avg_sq;
        }
        return;
    }
    const float bias_correction1 = (float)(1 - pow(beta1, (double)step));
    const float bias_correction2_sqrt = (float)sqrt(1 - pow(beta2, (double)step));
#pragma unroll
    for (int u = 0; u < kAdamIlp; ++u) {
        const int64_t e = base + (int64_t)u * kAdamBlock + threadIdx.x;
        if (e >= n) continue;
        float param = p[e];
        const float grad = g[e];
        float exp_avg = m[e], exp_avg_sq = v[e];
        // (torch's `beta1 * exp_avg + (1 - beta1) * grad` is compiled with the first product contracted into an fma -- established
        // bitwise by tools/adam_variants.py, profiles/r03/f_adam_variants.txt; spelled out here so that it does not depend on this
        // translation unit's contraction decisions)
        exp_avg = (float)__fma_rn(beta1, (double)exp_avg, (1 - beta1) * (double)grad);
        exp_avg_sq = (float)__fma_rn(beta2, (double)exp_avg_sq, (1 - beta2) * (double)grad * (double)grad);
        const float step_size = lr / bias_correction1;
        const float denom = (sqrtf(exp_avg_sq) / bias_correction2_sqrt) + eps;
        param -= step_size * exp_avg / denom;
        p[e] = param;
        m[e] = exp_avg;
        v[e] = exp_avg_sq;
    }
}

hipError_t launch_adam_multi(const nnr_adam_table& t, hipStream_t st) {
    const int blocks = t.block_first[t.n_tensors];
    if (blocks <= 0) return hipSuccess;
    hipLaunchKernelGGL(adam_multi_kernel, dim3(blocks), dim3(kAdamBlock), 0, st, t);
    return hipGetLastError();
}

}  // namespace nnr
