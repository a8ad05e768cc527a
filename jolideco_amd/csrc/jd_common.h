// Shared host-side helpers for libjolideco_hip.so (error reporting, launch checks).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>

#include "../../include/jolideco_hip.h"

namespace jd {

inline char* error_buffer() {
  static thread_local char buf[512] = {0};
  return buf;
}

inline int fail(int status, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(error_buffer(), 512, fmt, ap);
  va_end(ap);
  return status;
}

#define JD_HIP(call)                                                                          \
  do {                                                                                        \
    hipError_t e_ = (call);                                                                   \
    if (e_ != hipSuccess)                                                                     \
      return jd::fail(JD_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_),      \
                      __FILE__, __LINE__);                                                    \
  } while (0)

#define JD_LAUNCH_CHECK()                                                                     \
  do {                                                                                        \
    hipError_t e_ = hipGetLastError();                                                        \
    if (e_ != hipSuccess)                                                                     \
      return jd::fail(JD_ERR_HIP, "kernel launch failed: %s (%s:%d)", hipGetErrorString(e_),  \
                      __FILE__, __LINE__);                                                    \
  } while (0)

#define JD_REQUIRE(cond, ...)                                                                 \
  do {                                                                                        \
    if (!(cond)) return jd::fail(JD_ERR_INVALID, __VA_ARGS__);                                \
  } while (0)

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// wave64 reductions ------------------------------------------------------------------------
__device__ inline double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ inline float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// Deterministic block reduction of a double: every thread passes its value, thread 0 gets the
// block total (fixed order: lanes within a wave by xor-butterfly, waves in index order).
template <int BLOCK>
__device__ inline double block_sum(double v, double* smem /* BLOCK/64 doubles */) {
  v = wave_sum(v);
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) smem[wave] = v;
  __syncthreads();
  double total = 0.0;
  if (threadIdx.x == 0) {
#pragma unroll
    for (int i = 0; i < BLOCK / 64; ++i) total += smem[i];
  }
  return total;
}

// One pixel of nn.PoissonNLLLoss(log_input=False, eps, full=True) without its Stirling term (jolideco/loss.py:35-37):
//   term = n - c log(n + eps),   g = d term / d n / N = (1 - c / (n + eps)) / N
// with the hardware reciprocal plus one Newton step (<= 1 ulp) and v_log_f32 (log2, ~1 ulp) in place of the IEEE
// division and the accurate logf: ~10 instead of ~25 VALU instructions per pixel in kernels that are issue bound.
// EVERY Poisson pass of the library goes through this function, so all paths (fused into the separable
// convolution, stand-alone, pooled, calibrated) produce the same bits for the same n, c.
__device__ __forceinline__ void poisson_point(float n, float c, float eps, float inv_n, float& term, float& g) {
  const float ne = n + eps;
  float r = __builtin_amdgcn_rcpf(ne);
  r = fmaf(fmaf(-ne, r, 1.f), r, r);
  term = fmaf(-c, __builtin_amdgcn_logf(ne) * 0.69314718055994531f, n);
  g = fmaf(-c, r, 1.f) * inv_n;
}

// Sub-pixel shift of the calibration (jolideco/models/npred.py:298-402, utils/torch.py:196-223: affine_grid + grid_sample,
// bilinear, zero padding, align_corners=False, pure translation): in pixel units the sample point of output pixel (i, j) is
// (i + scale * shift_y, j + scale * shift_x) -- the same integer offsets and bilinear weights for every pixel.
struct ShiftGeom {
  int fy, fx;                // integer parts
  float wy0, wy1, wx0, wx1;  // weights of rows fy, fy + 1 / columns fx, fx + 1
};

__device__ __forceinline__ ShiftGeom shift_geom(const float* shift_xy, float scale) {
  const float sx = scale * shift_xy[0], sy = scale * shift_xy[1];
  const float flx = floorf(sx), fly = floorf(sy);
  ShiftGeom g;
  g.fx = (int)flx, g.fy = (int)fly;
  g.wx1 = sx - flx, g.wx0 = 1.f - g.wx1;
  g.wy1 = sy - fly, g.wy0 = 1.f - g.wy1;
  return g;
}

struct __attribute__((packed, aligned(4))) F4U4 {  // 16 bytes at a 4-byte aligned address: one global_load_dwordx4
  float x, y, z, w;
};

// out = (accumulate ? out : 0) + scale * sum(partials[0..n)) + offset, summed in index order in fp64.
int launch_finalize_sum(const double* partials, int n, double scale, double offset, float* out,
                        int accumulate, hipStream_t stream);

}  // namespace jd
