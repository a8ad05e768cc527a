// The optimizer update of one pixel, shared by the stand-alone kernel (elementwise.hip: adam_kernel) and by the gather
// kernel of the GMM prior that applies it in its epilogue (gmm.hip): ONE device function, so both produce the same bits.
#pragma once
#include <hip/hip_runtime.h>

namespace jd {

struct AdamArgs {
  float* theta;
  const float* flux_in;
  float* flux_out;
  float* grad_flux;
  float* m;
  float* v;
  const float* mask;
  size_t n;
  float step_size, beta1, beta2, one_minus_beta1, one_minus_beta2, bias2_sqrt, eps, lr;
  int zero_grad, sgd;
  int linear;  // use_log_flux=False: theta IS the flux (models/core.py:586-594), no exp / chain rule
  // nullable, device [2] = {step_size, bias2_sqrt}: read instead of the two members above (use_device_bias) -- the step
  // count of a captured graph's optimizer step lives in device memory, its launch arguments never change
  const float* bias_dev;
};

__device__ __forceinline__ void use_device_bias(AdamArgs& a) {
  if (a.bias_dev) a.step_size = a.bias_dev[0], a.bias2_sqrt = a.bias_dev[1];
}

__device__ inline void adam_update(float& th, float& m, float& v, float g, const AdamArgs& a) {
  // every product and sum rounded on its own: whether the compiler fuses a multiply-add depends on the kernel around it
  // (adam_multi_kernel came out one ulp from adam_kernel), and every kernel that steps a parameter must give the same bits
#ifndef JD_ADAM_CONTRACT  // (A/B only: -DJD_ADAM_CONTRACT leaves the fusing to the compiler)
#pragma clang fp contract(off)
#endif
  if (a.sgd) {
    th = th - a.lr * g;
    return;
  }
  // exp_avg.lerp_(grad, 1 - beta1); exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
  m = m + a.one_minus_beta1 * (g - m);
  v = v * a.beta2 + a.one_minus_beta2 * (g * g);
  const float denom = sqrtf(v) / a.bias2_sqrt + a.eps;
  th = th - a.step_size * (m / denom);
}

// chain rule + update + new flux of one pixel: th, f (flux in -> flux out), m, v are updated in place; gf = d loss / d flux
__device__ __forceinline__ void adam_pixel(float& th, float& f, float& m, float& v, float gf, float mk, const AdamArgs& a) {
  // d flux / d theta = exp(theta) * mask = flux  (models/core.py:588-592); linear: = mask
  const float dfdth = a.linear ? (a.mask ? mk : 1.f) : f;
  adam_update(th, m, v, gf * dfdth, a);
  f = a.linear ? th : expf(th);
  if (a.mask) f *= mk;
}

}  // namespace jd
