// HBM-bound element-wise kernels of the MAP step (gfx950): pad*exposure (K1), k-space complex
// multiply (K2), fused clip + background + Poisson NLL + gradient (K3), adjoint epilogue (K5),
// exp/chain-rule + Adam/SGD (K6) and the element-wise priors.  All are streaming kernels:
// one pass over each operand, 16 B per lane where the row alignment allows it, fp64 block
// partials + a single fixed-order finalize for every scalar so results are run-to-run identical.
#include <cstdint>
#include <cstdlib>

#include "jd_common.h"
#include "kernels.h"
#include "jd_adam.h"

namespace jd {

constexpr int BLOCK = 256;

// ------------------------------------------------------------------------------------------
// finalize: out = [out +] scale * sum(partials) + offset   (one block, fixed order)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void finalize_sum_kernel(const double* __restrict__ partials, int n,
                                                            double scale, double offset,
                                                            float* __restrict__ out, int accumulate) {
  __shared__ double smem[BLOCK / 64];
  double acc = 0.0;
  for (int i = threadIdx.x; i < n; i += BLOCK) acc += partials[i];
  double total = block_sum<BLOCK>(acc, smem);
  if (threadIdx.x == 0) {
    double v = scale * total + offset;
    if (accumulate) v += (double)out[0];
    out[0] = (float)v;
  }
}

int launch_finalize_sum(const double* partials, int n, double scale, double offset, float* out,
                        int accumulate, hipStream_t stream) {
  finalize_sum_kernel<<<1, BLOCK, 0, stream>>>(partials, n, scale, offset, out, accumulate);
  JD_LAUNCH_CHECK();
  return JD_OK;
}

struct FinalizeRowsArgs {
  const double* partials;
  int n;
  double scale;
  float offset[SEP_MAX_BATCH];
  float* out[SEP_MAX_BATCH];
};

__global__ __launch_bounds__(BLOCK) void finalize_rows_kernel(FinalizeRowsArgs a) {
  __shared__ double smem[BLOCK / 64];
  const double* row = a.partials + (size_t)blockIdx.x * a.n;
  double acc = 0.0;
  for (int i = threadIdx.x; i < a.n; i += BLOCK) acc += row[i];
  const double total = block_sum<BLOCK>(acc, smem);
  if (threadIdx.x == 0) a.out[blockIdx.x][0] = (float)(a.scale * total + (double)a.offset[blockIdx.x]);
}

int launch_finalize_rows(const double* partials, int n, int n_out, double scale, const float* offset_host,
                         float* const* out, hipStream_t stream) {
  if (n_out < 1 || n_out > SEP_MAX_BATCH) return fail(JD_ERR_INVALID, "finalize_rows: %d outputs not in [1, %d]", n_out, SEP_MAX_BATCH);
  FinalizeRowsArgs a{};
  a.partials = partials, a.n = n, a.scale = scale;
  for (int d = 0; d < n_out; ++d) a.offset[d] = offset_host[d], a.out[d] = out[d];
  finalize_rows_kernel<<<n_out, BLOCK, 0, stream>>>(a);
  JD_LAUNCH_CHECK();
  return JD_OK;
}

// ------------------------------------------------------------------------------------------
// K1: padded[y][x] = image[y][x] * scale[y][x] inside (H, W), 0 in the padding
// ------------------------------------------------------------------------------------------
template <int VEC>
__global__ __launch_bounds__(BLOCK) void pad_mul_kernel(const float* __restrict__ image,
                                                       const float* __restrict__ scale,
                                                       float* __restrict__ padded, int H, int W, int Wp) {
  const int y = blockIdx.y;
  const int x0 = (blockIdx.x * BLOCK + threadIdx.x) * VEC;
  if (x0 >= Wp) return;
  float v[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) v[i] = 0.f;
  if (y < H && x0 < W) {
    const size_t off = (size_t)y * W + x0;
    if constexpr (VEC == 4) {
      // W % 4 == 0 on this path, so x0 < W implies x0 + 3 < W
      const float4 a = *reinterpret_cast<const float4*>(image + off);
      float4 s = make_float4(1.f, 1.f, 1.f, 1.f);
      if (scale) s = *reinterpret_cast<const float4*>(scale + off);
      v[0] = a.x * s.x, v[1] = a.y * s.y, v[2] = a.z * s.z, v[3] = a.w * s.w;
    } else {
      v[0] = image[off] * (scale ? scale[off] : 1.f);
    }
  }
  float* dst = padded + (size_t)y * Wp + x0;
  if constexpr (VEC == 4) {
    *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
  } else {
    dst[0] = v[0];
  }
}

int launch_pad_mul(const float* image, const float* scale, float* padded, int H, int W, int Hp, int Wp,
                   hipStream_t stream) {
  const bool vec = (W % 4 == 0) && (Wp % 4 == 0);
  const int per_block = BLOCK * (vec ? 4 : 1);
  dim3 grid((Wp + per_block - 1) / per_block, Hp);
  ProfScope prof(JD_KERNEL_PAD_MUL, stream);
  if (vec)
    pad_mul_kernel<4><<<grid, BLOCK, 0, stream>>>(image, scale, padded, H, W, Wp);
  else
    pad_mul_kernel<1><<<grid, BLOCK, 0, stream>>>(image, scale, padded, H, W, Wp);
  JD_LAUNCH_CHECK();
  return JD_OK;
}

// ------------------------------------------------------------------------------------------
// K2: spec *= khat   or   spec *= conj(khat)     (interleaved complex64)
// ------------------------------------------------------------------------------------------
template <bool CONJ>
__global__ __launch_bounds__(BLOCK) void cmul_kernel(float2* __restrict__ spec,
                                                    const float2* __restrict__ khat, size_t n) {
  const size_t stride = (size_t)gridDim.x * BLOCK * 2;
  for (size_t i = ((size_t)blockIdx.x * BLOCK + threadIdx.x) * 2; i < n; i += stride) {
    if (i + 1 < n) {
      float4 a = *reinterpret_cast<float4*>(spec + i);
      const float4 k = *reinterpret_cast<const float4*>(khat + i);
      const float k1 = CONJ ? -k.y : k.y, k3 = CONJ ? -k.w : k.w;
      float4 r;
      r.x = a.x * k.x - a.y * k1;
      r.y = a.x * k1 + a.y * k.x;
      r.z = a.z * k.z - a.w * k3;
      r.w = a.z * k3 + a.w * k.z;
      *reinterpret_cast<float4*>(spec + i) = r;
    } else {
      const float2 a = spec[i];
      const float2 k = khat[i];
      const float ky = CONJ ? -k.y : k.y;
      spec[i] = make_float2(a.x * k.x - a.y * ky, a.x * ky + a.y * k.x);
    }
  }
}

int launch_cmul(float2* spec, const float2* khat, size_t n, bool conj, hipStream_t stream) {
  size_t blocks = (n / 2 + BLOCK - 1) / BLOCK;
  if (blocks > 8192) blocks = 8192;
  if (blocks == 0) blocks = 1;
  ProfScope prof(JD_KERNEL_CMUL, stream);
  if (conj)
    cmul_kernel<true><<<(unsigned)blocks, BLOCK, 0, stream>>>(spec, khat, n);
  else
    cmul_kernel<false><<<(unsigned)blocks, BLOCK, 0, stream>>>(spec, khat, n);
  JD_LAUNCH_CHECK();
  return JD_OK;
}

// ------------------------------------------------------------------------------------------
// K3: fused clip + sum over components + background + Poisson NLL partial sums + gradient.
// Reads conv_c at the crop offset of the padded grid, writes g_c over the WHOLE padded grid
// (zeros outside (H, W)) so the buffer is directly the input of the adjoint R2C.
// Algorithmic bytes: 16 B/pixel for one component (conv, background, counts in; g out).
// ------------------------------------------------------------------------------------------
// CAL: a dataset calibration is present (background norm read from device memory, second partial sum
// for its gradient); a compile-time switch so that the plain pass carries none of it.
template <int VEC, int ROWS, bool CAL>
__global__ __launch_bounds__(BLOCK) void poisson_fused_kernel(PoissonArgs a) {
  __shared__ double smem[BLOCK / 64];
  const int x0 = (blockIdx.x * BLOCK + threadIdx.x) * VEC;
  double local = 0.0, local_b = 0.0;
  const float bkg_norm = (CAL && a.log_bkg_norm) ? expf(a.log_bkg_norm[0]) : 1.f;  // NPredCalibration.background_norm
  // ROWS rows per thread: every load of the thread is issued before the first one is consumed
  // (ROWS x (2 + n_comp) x 16 B in flight per lane)
  float b[ROWS][VEC], c[ROWS][VEC], conv[ROWS][JD_MAX_COMPONENTS][VEC];
  bool in_image[ROWS];
#pragma unroll
  for (int r = 0; r < ROWS; ++r) {
    const int y = blockIdx.y * ROWS + r;
    in_image[r] = (y < a.H) && (x0 < a.W);
    if (in_image[r]) {
      const size_t off = (size_t)y * a.W + x0;
      const size_t poff = (size_t)(y + a.oy) * a.Wp + (x0 + a.ox);
      if constexpr (VEC == 4) {
        const float4 bb = *reinterpret_cast<const float4*>(a.background + off);
        const float4 cc = *reinterpret_cast<const float4*>(a.counts + off);
        b[r][0] = bb.x, b[r][1] = bb.y, b[r][2] = bb.z, b[r][3] = bb.w;
        c[r][0] = cc.x, c[r][1] = cc.y, c[r][2] = cc.z, c[r][3] = cc.w;
      } else {
        b[r][0] = a.background[off];
        c[r][0] = a.counts[off];
      }
#pragma unroll
      for (int k = 0; k < JD_MAX_COMPONENTS; ++k) {
        if (k >= a.n_comp) break;
        if constexpr (VEC == 4) {
          const float4 v = *reinterpret_cast<const float4*>(a.conv[k] + poff);
          conv[r][k][0] = v.x, conv[r][k][1] = v.y, conv[r][k][2] = v.z, conv[r][k][3] = v.w;
        } else {
          conv[r][k][0] = a.conv[k][poff];
        }
      }
    }
  }

#pragma unroll
  for (int r = 0; r < ROWS; ++r) {
    const int y = blockIdx.y * ROWS + r;
    float g[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) g[i] = 0.f;
    if (in_image[r]) {
      float n[VEC];
#pragma unroll
      for (int i = 0; i < VEC; ++i) n[i] = 0.f;
#pragma unroll
      for (int k = 0; k < JD_MAX_COMPONENTS; ++k) {
        if (k >= a.n_comp) break;
#pragma unroll
        for (int i = 0; i < VEC; ++i) n[i] += fmaxf(conv[r][k][i], 0.f);  // clip per component, npred.py:191
      }
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        const float bi = (CAL && a.log_bkg_norm) ? b[r][i] * bkg_norm : b[r][i];
        n[i] += bi;  // background added last, un-convolved (npred.py:234-261)
        float term;
        poisson_point(n[i], c[r][i], a.eps, a.inv_n, term, g[i]);
        local += (double)term;
        if (CAL) local_b += (double)(g[i] * bi);
      }
      if (a.npred_out) {
        const size_t off = (size_t)y * a.W + x0;
        if constexpr (VEC == 4)
          *reinterpret_cast<float4*>(a.npred_out + off) = make_float4(n[0], n[1], n[2], n[3]);
        else
          a.npred_out[off] = n[0];
      }
    }
    if (a.write_grad && x0 < a.Wp && y < a.Hp) {
      const size_t goff = (size_t)y * a.Wp + x0;
#pragma unroll
      for (int k = 0; k < JD_MAX_COMPONENTS; ++k) {
        if (k >= a.n_comp) break;
        float gk[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i)
          gk[i] = (in_image[r] && conv[r][k][i] >= 0.f) ? g[i] : 0.f;  // clamp backward: passes where conv >= 0
        if constexpr (VEC == 4)
          *reinterpret_cast<float4*>(a.g[k] + goff) = make_float4(gk[0], gk[1], gk[2], gk[3]);
        else
          a.g[k][goff] = gk[0];
      }
    }
  }

  const double total = block_sum<BLOCK>(local, smem);
  if (threadIdx.x == 0) a.partials[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = total;
  if (CAL && a.partials_b) {  // wave-uniform
    __syncthreads();
    const double total_b = block_sum<BLOCK>(local_b, smem);
    if (threadIdx.x == 0) a.partials_b[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = total_b;
  }
}

int launch_poisson_fused(const PoissonArgs& a, int* n_partials, hipStream_t stream) {
  const bool vec = (a.W % 4 == 0) && (a.Wp % 4 == 0) && (a.ox % 4 == 0);
  const int per_block = BLOCK * (vec ? 4 : 1);
  const int span = a.write_grad ? a.Wp : a.W;
  const int rows = a.write_grad ? a.Hp : a.H;
  int rows_per_block = 1;  // measured on MI355X at 2048^2: 1 row 14.6 us, 2 rows 15.8 us, 4 rows 17.3 us
  {  // tuning override
    const int v = opt_value(OPT_POISSON_ROWS, 0);
    if (vec && (v == 1 || v == 2 || v == 4)) rows_per_block = v;
  }
  dim3 grid((span + per_block - 1) / per_block, (rows + rows_per_block - 1) / rows_per_block);
  *n_partials = grid.x * grid.y;
  ProfScope prof(JD_KERNEL_POISSON_FUSED, stream);
  const bool cal = a.log_bkg_norm != nullptr;
  if (!vec && cal)
    poisson_fused_kernel<1, 1, true><<<grid, BLOCK, 0, stream>>>(a);
  else if (!vec)
    poisson_fused_kernel<1, 1, false><<<grid, BLOCK, 0, stream>>>(a);
  else if (cal)
    poisson_fused_kernel<4, 1, true><<<grid, BLOCK, 0, stream>>>(a);
  else if (rows_per_block == 4)
    poisson_fused_kernel<4, 4, false><<<grid, BLOCK, 0, stream>>>(a);
  else if (rows_per_block == 2)
    poisson_fused_kernel<4, 2, false><<<grid, BLOCK, 0, stream>>>(a);
  else
    poisson_fused_kernel<4, 1, false><<<grid, BLOCK, 0, stream>>>(a);
  JD_LAUNCH_CHECK();
  return JD_OK;
}

// ------------------------------------------------------------------------------------------
// K3 with sum-pooling (upsampling_factor u > 1, jolideco/models/npred.py:181-184): one thread per
// COUNTS pixel sums the u x u block of every component's convolution, clips, adds the background,
// accumulates the Poisson NLL and replicates g = d loss / d npred into the u x u block of g_c (the
// adjoint of the sum-pool), masked where the pooled value was clipped.  The padding of the g buffers
// (FFT method) is already zero: K1 rewrote the whole padded grid before the forward transform.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void poisson_pooled_kernel(PoissonArgs a) {
  __shared__ double smem[BLOCK / 64];
  const int u = a.up;
  const int Hd = a.H / u, Wd = a.W / u;
  const int y = blockIdx.y;
  const int x = blockIdx.x * BLOCK + threadIdx.x;
  double local = 0.0, local_b = 0.0;
  if (y < Hd && x < Wd) {
    const size_t off = (size_t)y * Wd + x;
    float pooled[JD_MAX_COMPONENTS];
    float n = 0.f;
#pragma unroll
    for (int k = 0; k < JD_MAX_COMPONENTS; ++k) {
      if (k >= a.n_comp) break;
      float acc = 0.f;
      for (int dy = 0; dy < u; ++dy) {
        const float* row = a.conv[k] + (size_t)(y * u + dy + a.oy) * a.Wp + (x * u + a.ox);
        for (int dx = 0; dx < u; ++dx) acc += row[dx];
      }
      pooled[k] = acc;
      n += fmaxf(acc, 0.f);  // clip per component after pooling (npred.py:181-191)
    }
    const float bi = a.log_bkg_norm ? a.background[off] * expf(a.log_bkg_norm[0]) : a.background[off];
    n += bi;
    const float c = a.counts[off];
    float term, g;
    poisson_point(n, c, a.eps, a.inv_n, term, g);
    local = (double)term;
    local_b = (double)(g * bi);
    if (a.npred_out) a.npred_out[off] = n;
    if (a.write_grad) {
#pragma unroll
      for (int k = 0; k < JD_MAX_COMPONENTS; ++k) {
        if (k >= a.n_comp) break;
        const float gk = pooled[k] >= 0.f ? g : 0.f;
        for (int dy = 0; dy < u; ++dy) {
          float* row = a.g[k] + (size_t)(y * u + dy) * a.Wp + (size_t)x * u;
          for (int dx = 0; dx < u; ++dx) row[dx] = gk;
        }
      }
    }
  }
  const double total = block_sum<BLOCK>(local, smem);
  if (threadIdx.x == 0) a.partials[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = total;
  if (a.partials_b) {
    __syncthreads();
    const double total_b = block_sum<BLOCK>(local_b, smem);
    if (threadIdx.x == 0) a.partials_b[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = total_b;
  }
}

int launch_poisson_pooled(const PoissonArgs& a, int* n_partials, hipStream_t stream) {
  const int Hd = a.H / a.up, Wd = a.W / a.up;
  dim3 grid((Wd + BLOCK - 1) / BLOCK, Hd);
  *n_partials = grid.x * grid.y;
  ProfScope prof(JD_KERNEL_POISSON_FUSED, stream);
  poisson_pooled_kernel<<<grid, BLOCK, 0, stream>>>(a);
  JD_LAUNCH_CHECK();
  return JD_OK;
}

int poisson_fused_max_partials(int Hp, int Wp) { return ((Wp + BLOCK - 1) / BLOCK) * Hp; }

// ------------------------------------------------------------------------------------------
// K5: adjoint epilogue. grad[p][q] (+)= coef * scale[p][q] * corr[(p-oy) mod Hp][(q-ox) mod Wp]
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void adjoint_epilogue_kernel(const float* __restrict__ corr,
                                                                const float* __restrict__ scale,
                                                                float* __restrict__ grad, int H, int W,
                                                                int Hp, int Wp, int oy, int ox,
                                                                float coef, int accumulate) {
  const int p = blockIdx.y;
  const int q = blockIdx.x * BLOCK + threadIdx.x;
  if (q >= W) return;
  int sy = p - oy;
  if (sy < 0) sy += Hp;
  int sx = q - ox;
  if (sx < 0) sx += Wp;
  const size_t off = (size_t)p * W + q;
  float v = corr[(size_t)sy * Wp + sx] * coef;
  if (scale) v *= scale[off];
  if (accumulate) v += grad[off];
  grad[off] = v;
}

int launch_adjoint_epilogue(const float* corr, const float* scale, float* grad, int H, int W, int Hp,
                            int Wp, int oy, int ox, float coef, int accumulate, hipStream_t stream) {
  dim3 grid((W + BLOCK - 1) / BLOCK, H);
  ProfScope prof(JD_KERNEL_ADJOINT_EPILOGUE, stream);
  adjoint_epilogue_kernel<<<grid, BLOCK, 0, stream>>>(corr, scale, grad, H, W, Hp, Wp, oy, ox, coef,
                                                      accumulate);
  JD_LAUNCH_CHECK();
  return JD_OK;
}

// crop: out[y][x] = padded[y+oy][x+ox]
__global__ __launch_bounds__(BLOCK) void crop_kernel(const float* __restrict__ padded,
                                                    float* __restrict__ out, int H, int W, int Wp, int oy,
                                                    int ox) {
  const int y = blockIdx.y;
  const int x = blockIdx.x * BLOCK + threadIdx.x;
  if (x >= W) return;
  out[(size_t)y * W + x] = padded[(size_t)(y + oy) * Wp + (x + ox)];
}

int launch_crop(const float* padded, float* out, int H, int W, int Wp, int oy, int ox, hipStream_t stream) {
  dim3 grid((W + BLOCK - 1) / BLOCK, H);
  crop_kernel<<<grid, BLOCK, 0, stream>>>(padded, out, H, W, Wp, oy, ox);
  JD_LAUNCH_CHECK();
  return JD_OK;
}

// ------------------------------------------------------------------------------------------
// stand-alone Poisson NLL on a flat npred (autograd.Function seam of loss.py:35-37)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void poisson_nll_kernel(const float* __restrict__ npred,
                                                           const float* __restrict__ counts, size_t n,
                                                           float eps, float inv_n,
                                                           float* __restrict__ grad,
                                                           double* __restrict__ partials) {
  __shared__ double smem[BLOCK / 64];
  double local = 0.0;
  const size_t stride = (size_t)gridDim.x * BLOCK;
  for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride) {
    const float v = npred[i], c = counts[i];
    float term, g;
    poisson_point(v, c, eps, inv_n, term, g);
    local += (double)term;
    if (grad) grad[i] = g;
  }
  const double total = block_sum<BLOCK>(local, smem);
  if (threadIdx.x == 0) partials[blockIdx.x] = total;
}

// ------------------------------------------------------------------------------------------
// K6: chain rule + Adam (torch.optim.Adam single-tensor formula) + exp of the new theta
// ------------------------------------------------------------------------------------------
template <int VEC>
__global__ __launch_bounds__(BLOCK) void adam_kernel(AdamArgs a) {
  use_device_bias(a);
  const size_t stride = (size_t)gridDim.x * BLOCK * VEC;
  for (size_t i = ((size_t)blockIdx.x * BLOCK + threadIdx.x) * VEC; i < a.n; i += stride) {
    float th[VEC], f[VEC], gf[VEC], m[VEC], v[VEC], mk[VEC];
    if constexpr (VEC == 4) {
      const float4 t4 = *reinterpret_cast<const float4*>(a.theta + i);
      const float4 f4 = *reinterpret_cast<const float4*>(a.flux_in + i);
      const float4 g4 = *reinterpret_cast<const float4*>(a.grad_flux + i);
      th[0] = t4.x, th[1] = t4.y, th[2] = t4.z, th[3] = t4.w;
      f[0] = f4.x, f[1] = f4.y, f[2] = f4.z, f[3] = f4.w;
      gf[0] = g4.x, gf[1] = g4.y, gf[2] = g4.z, gf[3] = g4.w;
      if (!a.sgd) {
        const float4 m4 = *reinterpret_cast<const float4*>(a.m + i);
        const float4 v4 = *reinterpret_cast<const float4*>(a.v + i);
        m[0] = m4.x, m[1] = m4.y, m[2] = m4.z, m[3] = m4.w;
        v[0] = v4.x, v[1] = v4.y, v[2] = v4.z, v[3] = v4.w;
      }
      mk[0] = mk[1] = mk[2] = mk[3] = 1.f;
      if (a.mask) {
        const float4 k4 = *reinterpret_cast<const float4*>(a.mask + i);
        mk[0] = k4.x, mk[1] = k4.y, mk[2] = k4.z, mk[3] = k4.w;
      }
    } else {
      th[0] = a.theta[i], f[0] = a.flux_in[i], gf[0] = a.grad_flux[i];
      if (!a.sgd) m[0] = a.m[i], v[0] = a.v[i];
      mk[0] = a.mask ? a.mask[i] : 1.f;
    }
#pragma unroll
    for (int k = 0; k < VEC; ++k) adam_pixel(th[k], f[k], m[k], v[k], gf[k], mk[k], a);
    if constexpr (VEC == 4) {
      *reinterpret_cast<float4*>(a.theta + i) = make_float4(th[0], th[1], th[2], th[3]);
      *reinterpret_cast<float4*>(a.flux_out + i) = make_float4(f[0], f[1], f[2], f[3]);
      if (!a.sgd) {
        *reinterpret_cast<float4*>(a.m + i) = make_float4(m[0], m[1], m[2], m[3]);
        *reinterpret_cast<float4*>(a.v + i) = make_float4(v[0], v[1], v[2], v[3]);
      }
      if (a.zero_grad) *reinterpret_cast<float4*>(a.grad_flux + i) = make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
      a.theta[i] = th[0], a.flux_out[i] = f[0];
      if (!a.sgd) a.m[i] = m[0], a.v[i] = v[0];
      if (a.zero_grad) a.grad_flux[i] = 0.f;
    }
  }
}

static int launch_adam(const AdamArgs& a, hipStream_t stream) {
  const bool vec = (a.n % 4 == 0);
  const size_t per_block = (size_t)BLOCK * (vec ? 4 : 1);
  size_t blocks = (a.n + per_block - 1) / per_block;
  if (blocks > 4096) blocks = 4096;
  if (blocks == 0) blocks = 1;
  ProfScope prof(JD_KERNEL_ADAM, stream);
  if (vec)
    adam_kernel<4><<<(unsigned)blocks, BLOCK, 0, stream>>>(a);
  else
    adam_kernel<1><<<(unsigned)blocks, BLOCK, 0, stream>>>(a);
  JD_LAUNCH_CHECK();
  return JD_OK;
}

__global__ __launch_bounds__(BLOCK) void flux_from_theta_kernel(const float* __restrict__ theta,
                                                               const float* __restrict__ mask,
                                                               float* __restrict__ flux, size_t n, int linear) {
  const size_t stride = (size_t)gridDim.x * BLOCK;
  for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride) {
    float f = linear ? theta[i] : expf(theta[i]);
    if (mask) f *= mask[i];
    flux[i] = f;
  }
}

// ------------------------------------------------------------------------------------------
// element-wise priors
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void elementwise_prior_kernel(int kind, const float* __restrict__ flux,
                                                                 size_t n, float alpha, float beta,
                                                                 float coef, float* __restrict__ grad,
                                                                 double* __restrict__ partials) {
  __shared__ double smem[BLOCK / 64];
  double local = 0.0;
  const size_t stride = (size_t)gridDim.x * BLOCK;
  for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride) {
    const float f = flux[i];
    float value, dv;
    if (kind == 1) {  // inverse gamma: -beta/f + (-alpha-1)*log f      (priors/core.py:223-224)
      value = -beta / f + (-alpha - 1.f) * logf(f);
      dv = beta / (f * f) + (-alpha - 1.f) / f;
    } else {  // exponential: -alpha * f                                (priors/core.py:324)
      value = -alpha * f;
      dv = -alpha;
    }
    local += (double)value;
    if (grad) grad[i] += coef * dv;
  }
  const double total = block_sum<BLOCK>(local, smem);
  if (threadIdx.x == 0) partials[blockIdx.x] = total;
}

// small per-process scratch for the stand-alone reductions (grown on demand, never freed while
// the library is loaded; one per device would be needed for multi-device processes -- the
// framework runs one process per GPU)
static double* g_partials = nullptr;
static size_t g_partials_cap = 0;
static int ensure_partials(size_t n) {
  if (n <= g_partials_cap) return JD_OK;
  if (g_partials) (void)hipFree(g_partials);
  g_partials = nullptr;
  g_partials_cap = 0;
  JD_HIP(hipMalloc(&g_partials, n * sizeof(double)));
  g_partials_cap = n;
  return JD_OK;
}

}  // namespace jd

// ==========================================================================================
// C ABI
// ==========================================================================================
using namespace jd;

extern "C" int jd_poisson_nll(const float* npred, const float* counts, size_t n, float stirling_mean,
                              float eps, float* loss_out, float* grad_npred, void* stream) {
  JD_REQUIRE(npred && counts && loss_out && n > 0, "jd_poisson_nll: null argument or n == 0");
  size_t blocks = (n + BLOCK - 1) / BLOCK;
  if (blocks > 2048) blocks = 2048;
  int rc = ensure_partials(8192);
  if (rc) return rc;
  hipStream_t s = as_stream(stream);
  poisson_nll_kernel<<<(unsigned)blocks, BLOCK, 0, s>>>(npred, counts, n, eps, 1.f / (float)n, grad_npred,
                                                       g_partials);
  JD_LAUNCH_CHECK();
  return launch_finalize_sum(g_partials, (int)blocks, 1.0 / (double)n, (double)stirling_mean, loss_out, 0, s);
}

extern "C" int jd_elementwise_prior_fwd_bwd(int kind, const float* flux, size_t n, float alpha, float beta,
                                            float log_const, float* value_out, float grad_coef,
                                            float* grad_flux_accum, void* stream) {
  JD_REQUIRE(kind == 1 || kind == 2, "jd_elementwise_prior_fwd_bwd: kind must be 1 (inverse-gamma) or 2 (exponential)");
  JD_REQUIRE(flux && value_out && n > 0, "jd_elementwise_prior_fwd_bwd: null argument or n == 0");
  size_t blocks = (n + BLOCK - 1) / BLOCK;
  if (blocks > 2048) blocks = 2048;
  int rc = ensure_partials(8192);
  if (rc) return rc;
  hipStream_t s = as_stream(stream);
  elementwise_prior_kernel<<<(unsigned)blocks, BLOCK, 0, s>>>(kind, flux, n, alpha, beta, grad_coef,
                                                             grad_flux_accum, g_partials);
  JD_LAUNCH_CHECK();
  return launch_finalize_sum(g_partials, (int)blocks, 1.0 / (double)n, (double)log_const, value_out, 0, s);
}

extern "C" int jd_flux_from_theta(const float* theta, const float* mask, float* flux, size_t n, int use_log_flux,
                                  void* stream) {
  JD_REQUIRE(theta && flux && n > 0, "jd_flux_from_theta: null argument or n == 0");
  size_t blocks = (n + BLOCK - 1) / BLOCK;
  if (blocks > 8192) blocks = 8192;
  flux_from_theta_kernel<<<(unsigned)blocks, BLOCK, 0, as_stream(stream)>>>(theta, mask, flux, n, use_log_flux ? 0 : 1);
  JD_LAUNCH_CHECK();
  return JD_OK;
}

// Components that share one forward operator (models/npred.py:279-295 builds every component's model of a dataset from
// the SAME exposure and, unless `psf` is a dict, the same PSF): npred = sum_c clip(PSF * (flux_c E), 0) = PSF * ((sum_c
// flux_c) E) wherever no term is negative, and d loss / d flux_c is ONE image for all c.  The two helpers either side of
// the single-component launches (jolideco_amd/loss.py): the sum of the component fluxes, left to right, and the copy of
// the gradient image into the other components' gradient images.  16-byte accesses where n and the pointers allow.
struct ImagePtrs {
  const float* src[4];
  float* dst[4];
  int n_src, n_dst;
};

template <bool VEC>
__global__ __launch_bounds__(BLOCK) void sum_images_kernel(ImagePtrs p, size_t n) {
  const size_t stride = (size_t)gridDim.x * BLOCK;
  if (VEC) {
    const size_t n4 = n / 4;
    for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n4; i += stride) {
      float4 v = reinterpret_cast<const float4*>(p.src[0])[i];
      for (int c = 1; c < p.n_src; ++c) {
        const float4 w = reinterpret_cast<const float4*>(p.src[c])[i];
        v.x += w.x, v.y += w.y, v.z += w.z, v.w += w.w;
      }
      for (int d = 0; d < p.n_dst; ++d) reinterpret_cast<float4*>(p.dst[d])[i] = v;
    }
  } else {
    for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride) {
      float v = p.src[0][i];
      for (int c = 1; c < p.n_src; ++c) v += p.src[c][i];
      for (int d = 0; d < p.n_dst; ++d) p.dst[d][i] = v;
    }
  }
}

static int launch_sum_images(const ImagePtrs& p, size_t n, void* stream) {
  bool vec = n % 4 == 0;
  for (int c = 0; c < p.n_src; ++c) vec = vec && (reinterpret_cast<uintptr_t>(p.src[c]) & 15) == 0;
  for (int d = 0; d < p.n_dst; ++d) vec = vec && (reinterpret_cast<uintptr_t>(p.dst[d]) & 15) == 0;
  size_t blocks = ((vec ? n / 4 : n) + BLOCK - 1) / BLOCK;
  if (blocks > 8192) blocks = 8192;
  if (vec)
    sum_images_kernel<true><<<(unsigned)blocks, BLOCK, 0, as_stream(stream)>>>(p, n);
  else
    sum_images_kernel<false><<<(unsigned)blocks, BLOCK, 0, as_stream(stream)>>>(p, n);
  JD_LAUNCH_CHECK();
  return JD_OK;
}

extern "C" int jd_sum_images(float* out, const float* const* srcs, int n_srcs, size_t n, void* stream) {
  JD_REQUIRE(out && srcs && n > 0, "jd_sum_images: null argument or n == 0");
  JD_REQUIRE(n_srcs >= 1 && n_srcs <= 4, "jd_sum_images: %d source images (1 to 4)", n_srcs);
  ImagePtrs p{};
  for (int c = 0; c < n_srcs; ++c) {
    JD_REQUIRE(srcs[c], "jd_sum_images: source %d is null", c);
    p.src[c] = srcs[c];
  }
  p.n_src = n_srcs, p.dst[0] = out, p.n_dst = 1;
  return launch_sum_images(p, n, stream);
}

extern "C" int jd_copy_image_to(const float* src, float* const* dsts, int n_dsts, size_t n, void* stream) {
  JD_REQUIRE(src && dsts && n > 0, "jd_copy_image_to: null argument or n == 0");
  JD_REQUIRE(n_dsts >= 1 && n_dsts <= 4, "jd_copy_image_to: %d destination images (1 to 4)", n_dsts);
  ImagePtrs p{};
  p.src[0] = src, p.n_src = 1;
  for (int d = 0; d < n_dsts; ++d) {
    JD_REQUIRE(dsts[d] && dsts[d] != src, "jd_copy_image_to: destination %d is null or the source", d);
    p.dst[d] = dsts[d];
  }
  p.n_dst = n_dsts;
  return launch_sum_images(p, n, stream);
}

// The step scalars of an epoch from a pinned host row (device-accessible: zero-copy reads over the host link) into their
// device buffer: one small block instead of a hipMemcpyAsync, whose copy engine hand-over costs a stream 6-8 us between two
// epochs (tools/gpu/small_fits.py).
__global__ __launch_bounds__(256) void fetch_scalars_kernel(const int* __restrict__ src, int* __restrict__ dst, int n) {
  for (int i = threadIdx.x; i < n; i += 256) dst[i] = src[i];
}

extern "C" int jd_step_scalars_fetch(const int32_t* host_row, int32_t* dst, int n, void* stream) {
  JD_REQUIRE(host_row && dst && n > 0 && n <= (1 << 20), "jd_step_scalars_fetch: null argument or bad n");
  fetch_scalars_kernel<<<1, 256, 0, as_stream(stream)>>>(host_row, dst, n);
  JD_LAUNCH_CHECK();
  return JD_OK;
}

extern "C" int jd_adam_step(float* theta, const float* flux_in, float* flux_out, float* grad_flux,
                            float* exp_avg, float* exp_avg_sq, const float* mask, size_t n, float step_size,
                            float beta1, float beta2, float one_minus_beta1, float one_minus_beta2,
                            float bias2_sqrt, float eps, int zero_grad, int use_log_flux, const float* bias_dev,
                            void* stream) {
  JD_REQUIRE(theta && flux_in && flux_out && grad_flux && exp_avg && exp_avg_sq && n > 0,
             "jd_adam_step: null argument or n == 0");
  AdamArgs a{theta, flux_in, flux_out, grad_flux, exp_avg, exp_avg_sq, mask, n,
             step_size, beta1, beta2, one_minus_beta1, one_minus_beta2, bias2_sqrt, eps, 0.f, zero_grad, 0,
             use_log_flux ? 0 : 1, bias_dev};
  return launch_adam(a, as_stream(stream));
}

// Adam steps of MANY small parameter vectors in one launch (the calibration parameters of the datasets of a joint step:
// two tensors of 2 and 1 floats per dataset -- one launch each took 4.5 us, sixteen per step of eight observations).
// One block per tensor; the update is jd_adam_step's with use_log_flux = 0 (adam_update: the same device function).
constexpr int ADAM_MULTI_MAX = 64;
struct AdamMultiArgs {
  float* theta[ADAM_MULTI_MAX];
  const float* grad[ADAM_MULTI_MAX];
  float* m[ADAM_MULTI_MAX];
  float* v[ADAM_MULTI_MAX];
  int size[ADAM_MULTI_MAX];
  float step_size[ADAM_MULTI_MAX], bias2_sqrt[ADAM_MULTI_MAX];
  float beta1, beta2, one_minus_beta1, one_minus_beta2, eps;
  const float* bias_dev;  // nullable, device [2 n]: {step_size, bias2_sqrt} of tensor i at [2 i], read instead of the arrays above
};

__global__ __launch_bounds__(64) void adam_multi_kernel(AdamMultiArgs a) {
  float* theta = nullptr;
  const float* grad = nullptr;
  float *m = nullptr, *v = nullptr;
  int size = 0;
  AdamArgs s{};
  s.beta1 = a.beta1, s.beta2 = a.beta2, s.one_minus_beta1 = a.one_minus_beta1, s.one_minus_beta2 = a.one_minus_beta2, s.eps = a.eps;
  // (compile-time indices: indexing the by-value argument arrays with blockIdx would send them through scratch memory)
#pragma unroll
  for (int i = 0; i < ADAM_MULTI_MAX; ++i)
    if ((int)blockIdx.x == i) {
      theta = a.theta[i], grad = a.grad[i], m = a.m[i], v = a.v[i], size = a.size[i];
      s.step_size = a.step_size[i], s.bias2_sqrt = a.bias2_sqrt[i];
    }
  if (a.bias_dev) s.step_size = a.bias_dev[2 * blockIdx.x], s.bias2_sqrt = a.bias_dev[2 * blockIdx.x + 1];
  for (int j = threadIdx.x; j < size; j += 64) {
    float th = theta[j], mj = m[j], vj = v[j];
    adam_update(th, mj, vj, grad[j], s);
    theta[j] = th, m[j] = mj, v[j] = vj;
  }
}

extern "C" int jd_adam_step_multi(int n_tensors, float* const* theta, const float* const* grad, float* const* exp_avg,
                                  float* const* exp_avg_sq, const int* sizes, const float* step_size,
                                  const float* bias2_sqrt, float beta1, float beta2, float one_minus_beta1,
                                  float one_minus_beta2, float eps, const float* bias_dev, void* stream) {
  JD_REQUIRE(theta && grad && exp_avg && exp_avg_sq && sizes && (bias_dev || (step_size && bias2_sqrt)),
             "jd_adam_step_multi: null argument");
  JD_REQUIRE(n_tensors >= 1 && n_tensors <= ADAM_MULTI_MAX, "jd_adam_step_multi: n_tensors = %d not in [1, %d]", n_tensors,
             ADAM_MULTI_MAX);
  AdamMultiArgs a{};
  for (int i = 0; i < n_tensors; ++i) {
    JD_REQUIRE(theta[i] && grad[i] && exp_avg[i] && exp_avg_sq[i] && sizes[i] > 0, "jd_adam_step_multi: null tensor %d", i);
    a.theta[i] = theta[i], a.grad[i] = grad[i], a.m[i] = exp_avg[i], a.v[i] = exp_avg_sq[i], a.size[i] = sizes[i];
    if (!bias_dev) a.step_size[i] = step_size[i], a.bias2_sqrt[i] = bias2_sqrt[i];
  }
  a.bias_dev = bias_dev;
  a.beta1 = beta1, a.beta2 = beta2, a.one_minus_beta1 = one_minus_beta1, a.one_minus_beta2 = one_minus_beta2, a.eps = eps;
  ProfScope prof(JD_KERNEL_ADAM, as_stream(stream));
  adam_multi_kernel<<<n_tensors, 64, 0, as_stream(stream)>>>(a);
  JD_LAUNCH_CHECK();
  return JD_OK;
}

extern "C" int jd_sgd_step(float* theta, const float* flux_in, float* flux_out, float* grad_flux,
                           const float* mask, size_t n, float lr, int zero_grad, int use_log_flux, void* stream) {
  JD_REQUIRE(theta && flux_in && flux_out && grad_flux && n > 0, "jd_sgd_step: null argument or n == 0");
  AdamArgs a{theta, flux_in, flux_out, grad_flux, nullptr, nullptr, mask, n,
             0.f, 0.f, 0.f, 0.f, 0.f, 1.f, 0.f, lr, zero_grad, 1, use_log_flux ? 0 : 1, nullptr};
  return launch_adam(a, as_stream(stream));
}

extern "C" int jd_version(void) { return 100; }
extern "C" const char* jd_last_error(void) { return jd::error_buffer(); }
extern "C" const char* jd_target_arch(void) { return "gfx950"; }
