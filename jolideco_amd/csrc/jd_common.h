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

__device__ __forceinline__ ShiftGeom shift_geom_of(float shift_x, float shift_y, float scale) {
  const float sx = scale * shift_x, sy = scale * shift_y;
  const float flx = floorf(sx), fly = floorf(sy);
  ShiftGeom g;
  g.fx = (int)flx, g.fy = (int)fly;
  g.wx1 = sx - flx, g.wx0 = 1.f - g.wx1;
  g.wy1 = sy - fly, g.wy0 = 1.f - g.wy1;
  return g;
}

// A uniform read-only value through the CONSTANT address space: a scalar load (s_load, SGPR result, lgkmcnt) instead of a
// flat vector load with a full wait.  For values no launch of the library writes while it reads them (the calibration
// parameters: the optimizer step that changes them is a launch of its own).
__device__ __forceinline__ float cld(const float* p) { return *(const __attribute__((address_space(4))) float*)p; }

__device__ __forceinline__ ShiftGeom shift_geom(const float* shift_xy, float scale) {
  const float sx = scale * shift_xy[0], sy = scale * shift_xy[1];
  const float flx = floorf(sx), fly = floorf(sy);
  ShiftGeom g;
  g.fx = (int)flx, g.fy = (int)fly;
  g.wx1 = sx - flx, g.wx0 = 1.f - g.wx1;
  g.wy1 = sy - fly, g.wy0 = 1.f - g.wy1;
  return g;
}

// GLOBAL-memory accessors.  A pointer read from memory (a per-dataset pointer table) or selected between such a pointer and
// a kernel argument is a GENERIC pointer to the compiler: every access through it becomes a flat_load / flat_store, which
// count on vmcnt AND lgkmcnt and may return out of order with LDS traffic, so each is followed by a full
// `s_waitcnt vmcnt(0) lgkmcnt(0)` -- the loads of a kernel that also works in LDS (every FFT kernel) run one at a time.
// These accessors go through address space 1 explicitly (global_load / global_store, counted waits).  Neither a cast to
// address space 1 and back nor __builtin_assume(!is_shared && !is_private) survives to the pass that would use it
// (measured on this toolchain, tools/asm notes in DESIGN_LOG.md).  Builtin vector types: HIP's float4 / float2 structs have
// no copy from an address-space-qualified reference.
#define JD_AS1 __attribute__((address_space(1)))
typedef float jd_v4f __attribute__((ext_vector_type(4)));
typedef float jd_v2f __attribute__((ext_vector_type(2)));
typedef jd_v4f jd_v4f_a4 __attribute__((aligned(4)));  // 16 bytes at a 4-byte aligned address: one global_load_dwordx4
__device__ __forceinline__ float gld(const float* p) { return *(const JD_AS1 float*)p; }
__device__ __forceinline__ void gst(float* p, float v) { *(JD_AS1 float*)p = v; }
__device__ __forceinline__ float4 gld4(const float* p) {  // 16-byte aligned
  const jd_v4f v = *(const JD_AS1 jd_v4f*)(const JD_AS1 float*)p;
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float4 gld4u(const float* p) {  // 4-byte aligned
  const jd_v4f v = *(const JD_AS1 jd_v4f_a4*)(const JD_AS1 float*)p;
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float4 gld4(const float2* p) {  // two complex numbers, 16-byte aligned
  const jd_v4f v = *(const JD_AS1 jd_v4f*)(const JD_AS1 float2*)p;
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float2 gld2(const float2* p) {
  const jd_v2f v = *(const JD_AS1 jd_v2f*)(const JD_AS1 float2*)p;
  return float2{v.x, v.y};
}
__device__ __forceinline__ jd_v2f gld2(const float* p) { return *(const JD_AS1 jd_v2f*)(const JD_AS1 float*)p; }  // 8-byte aligned
__device__ __forceinline__ void gst2(float* p, jd_v2f v) { *(JD_AS1 jd_v2f*)(JD_AS1 float*)p = v; }
__device__ __forceinline__ void gst4(float* p, float4 v) { *(JD_AS1 jd_v4f*)(JD_AS1 float*)p = jd_v4f{v.x, v.y, v.z, v.w}; }
__device__ __forceinline__ void gst4u(float* p, float4 v) { *(JD_AS1 jd_v4f_a4*)(JD_AS1 float*)p = jd_v4f{v.x, v.y, v.z, v.w}; }
__device__ __forceinline__ void gst4(float2* p, float4 v) { *(JD_AS1 jd_v4f*)(JD_AS1 float2*)p = jd_v4f{v.x, v.y, v.z, v.w}; }
__device__ __forceinline__ void gst2(float2* p, float2 v) { *(JD_AS1 jd_v2f*)(JD_AS1 float2*)p = jd_v2f{v.x, v.y}; }

// The same five pixels in two steps: `issue_row5` is ONE unconditional 16-byte load + one float at clamped coordinates
// (global address space), `finish_row5` puts the zeros of the outside in -- register moves only.  With load_row5's three
// exits the compiler gave every row's vector load a full `s_waitcnt vmcnt(0)` (the border exit's scalar loads write the
// same registers) and took the pointers of the batch table for generic ones (flat loads): the 14 row loads of a dataset,
// meant to be in flight together, ran as 14 dependent round trips.  Needs W >= 5.
struct Raw5 {
  float q[5];
  int delta;  // wanted first column minus loaded first column
  bool yin;   // the row exists
};

__device__ __forceinline__ Raw5 issue_row5(const float* img, int H, int W, int y, int x) {
  Raw5 r;
  const int yc = min(max(y, 0), H - 1), xc = max(min(x, W - 5), 0);  // (W >= 5: the callers' launch conditions ask for W >= 8)
  r.yin = y == yc, r.delta = x - xc;
  const float* row = img + ((size_t)yc * W + xc);
  const float4 q = gld4u(row);
  r.q[0] = q.x, r.q[1] = q.y, r.q[2] = q.z, r.q[3] = q.w, r.q[4] = gld(row + 4);
  return r;
}

struct Row5 {
  float v[5];
};

__device__ __forceinline__ Row5 finish_row5(const Raw5& r) {
  Row5 o;
  if (r.delta == 0) {  // (all but the threads at the left and right image border)
#pragma unroll
    for (int i = 0; i < 5; ++i) o.v[i] = r.yin ? r.q[i] : 0.f;
  } else {
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int j = i + r.delta;  // column x + i is element j of the loaded five, or outside the image
      float v = 0.f;
#pragma unroll
      for (int jj = 0; jj < 5; ++jj) v = j == jj ? r.q[jj] : v;
      o.v[i] = r.yin ? v : 0.f;
    }
  }
  return o;
}

struct __attribute__((packed, aligned(4))) F4U4 {  // 16 bytes at a 4-byte aligned address: one global_load_dwordx4
  float x, y, z, w;
};

// out = (accumulate ? out : 0) + scale * sum(partials[0..n)) + offset, summed in index order in fp64.
int launch_finalize_sum(const double* partials, int n, double scale, double offset, float* out,
                        int accumulate, hipStream_t stream);

}  // namespace jd
