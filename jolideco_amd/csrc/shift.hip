// Sub-pixel shift of a flux image, the dataset calibration of jolideco/models/npred.py:298-402:
// `shift_image_torch` (jolideco/utils/torch.py:196-223) = affine_grid + grid_sample (bilinear, zero
// padding, align_corners=False) with a pure translation.  In pixel units the sample point of output
// pixel (i, j) is (i + scale * shift_y, j + scale * shift_x), so the integer offsets and the
// bilinear weights are the same for every pixel.  The shift is read from DEVICE memory (it is a
// trainable parameter that changes every step without the host looking at it).
#include "jd_common.h"
#include "kernels.h"

namespace jd {

constexpr int SHIFT_ROWS = 16;  // rows per block of the backward kernel

__device__ __forceinline__ float at(const float* img, int H, int W, int y, int x) {
  return (y >= 0 && y < H && x >= 0 && x < W) ? img[(size_t)y * W + x] : 0.f;
}

__global__ __launch_bounds__(256) void shift_fwd_kernel(const float* __restrict__ in, float* __restrict__ out, int H,
                                                        int W, const float* __restrict__ shift_xy, float scale) {
  const int y = blockIdx.y, x = blockIdx.x * 256 + threadIdx.x;
  if (x >= W) return;
  const ShiftGeom g = shift_geom(shift_xy, scale);
  const int y0 = y + g.fy, x0 = x + g.fx;
  // same association as grid_sample: nw * (w_x0 w_y0) + ne * (w_x1 w_y0) + sw * (w_x0 w_y1) + se * (w_x1 w_y1)
  out[(size_t)y * W + x] = at(in, H, W, y0, x0) * (g.wx0 * g.wy0) + at(in, H, W, y0, x0 + 1) * (g.wx1 * g.wy0) +
                           at(in, H, W, y0 + 1, x0) * (g.wx0 * g.wy1) + at(in, H, W, y0 + 1, x0 + 1) * (g.wx1 * g.wy1);
}

// Backward of the shift for one image:
//   grad_in[p][q] (+)= sum_{a,b} w_a^y w_b^x gs[p - fy - a][q - fx - b]         (adjoint, gather form)
//   d/d shift_x = scale * sum_ij gs[i][j] * ((in[y0][x0+1] - in[y0][x0]) w_y0 + (in[y0+1][x0+1] - in[y0+1][x0]) w_y1)
//   d/d shift_y analogous; zero padding outside the image (grid_sample's in-bounds checks).
__global__ __launch_bounds__(256) void shift_bwd_kernel(const float* __restrict__ in, const float* __restrict__ gs,
                                                        float* __restrict__ grad_in, int accumulate, int H, int W,
                                                        const float* __restrict__ shift_xy, float scale,
                                                        double* __restrict__ partials) {
#pragma clang fp contract(off)  // (every operation rounded on its own: the batched form below must give the same bits)
  __shared__ double smem[256 / 64];
  const int x = blockIdx.x * 256 + threadIdx.x;
  const ShiftGeom g = shift_geom(shift_xy, scale);
  double dsx = 0.0, dsy = 0.0;
  // a block owns SHIFT_ROWS rows of its 256 columns: one pair of partial sums per 16 rows (at 4096^2 the one-block
  // finalize kernel behind this one summed 131 072 partials in 92 us -- 0.7 ms per step of eight calibrated observations)
  for (int y = blockIdx.y * SHIFT_ROWS; y < min((int)(blockIdx.y + 1) * SHIFT_ROWS, H); ++y) {
    if (x >= W) break;
    const int py = y - g.fy, px = x - g.fx;
    float v = at(gs, H, W, py, px) * (g.wx0 * g.wy0) + at(gs, H, W, py, px - 1) * (g.wx1 * g.wy0) +
              at(gs, H, W, py - 1, px) * (g.wx0 * g.wy1) + at(gs, H, W, py - 1, px - 1) * (g.wx1 * g.wy1);
    const size_t off = (size_t)y * W + x;
    if (accumulate) v += grad_in[off];
    grad_in[off] = v;
    // derivative of OUTPUT pixel (y, x) w.r.t. the sample position
    const int y0 = y + g.fy, x0 = x + g.fx;
    const float nw = at(in, H, W, y0, x0), ne = at(in, H, W, y0, x0 + 1);
    const float sw = at(in, H, W, y0 + 1, x0), se = at(in, H, W, y0 + 1, x0 + 1);
    const float go = gs[off];
    dsx += (double)(go * ((ne - nw) * g.wy0 + (se - sw) * g.wy1));
    dsy += (double)(go * ((sw - nw) * g.wx0 + (se - ne) * g.wx1));
  }
  const double tx = block_sum<256>(dsx, smem);
  __syncthreads();
  const double ty = block_sum<256>(dsy, smem);
  if (threadIdx.x == 0) {
    const size_t b = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
    partials[2 * b] = tx * (double)scale;
    partials[2 * b + 1] = ty * (double)scale;
  }
}

// The transposed shifts of SEVERAL datasets in one launch (calibrated batched step): per pixel the datasets' terms are
// added in dataset order -- v = [grad_in +] t_0, then v = t_d + v -- exactly as the per-dataset launches leave them when each
// accumulates onto its predecessor; a dataset without a shift contributes its image as it is (what its adjoint launch
// would have accumulated directly).  Partial sums of d loss / d shift_xy per dataset: partials + d * partials_stride.
__global__ __launch_bounds__(256) void shift_bwd_batch_kernel(const float* __restrict__ in, const FftBatch* __restrict__ batch,
                                                              int n_datasets, float* __restrict__ grad_in, int accumulate, int H,
                                                              int W, float scale, double* __restrict__ partials,
                                                              size_t partials_stride) {
#pragma clang fp contract(off)
  __shared__ double smem[256 / 64];
  const int x = blockIdx.x * 256 + threadIdx.x;
  const size_t b = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
  const int y_begin = blockIdx.y * SHIFT_ROWS, y_end = min((int)(blockIdx.y + 1) * SHIFT_ROWS, H);
  // (rows outer, datasets inner would keep v in a register; datasets outer keeps the per-dataset sums in registers --
  // the per-dataset block sums need the latter: v goes through the gradient image, which this block owns)
  for (int d = 0; d < n_datasets; ++d) {
    const float* gs = batch->gshift[d];
    const float* shift_xy = batch->shift_xy[d];
    const bool add = accumulate || d > 0;
    double dsx = 0.0, dsy = 0.0;
    if (shift_xy) {
      const ShiftGeom g = shift_geom(shift_xy, scale);
      for (int y = y_begin; y < y_end; ++y) {
        if (x >= W) break;
        const int py = y - g.fy, px = x - g.fx;
        float v = at(gs, H, W, py, px) * (g.wx0 * g.wy0) + at(gs, H, W, py, px - 1) * (g.wx1 * g.wy0) +
                  at(gs, H, W, py - 1, px) * (g.wx0 * g.wy1) + at(gs, H, W, py - 1, px - 1) * (g.wx1 * g.wy1);
        const size_t off = (size_t)y * W + x;
        if (add) v += grad_in[off];
        grad_in[off] = v;
        const int y0 = y + g.fy, x0 = x + g.fx;
        const float nw = at(in, H, W, y0, x0), ne = at(in, H, W, y0, x0 + 1);
        const float sw = at(in, H, W, y0 + 1, x0), se = at(in, H, W, y0 + 1, x0 + 1);
        const float go = gs[off];
        dsx += (double)(go * ((ne - nw) * g.wy0 + (se - sw) * g.wy1));
        dsy += (double)(go * ((sw - nw) * g.wx0 + (se - ne) * g.wx1));
      }
    } else {
      for (int y = y_begin; y < y_end; ++y) {
        if (x >= W) break;
        const size_t off = (size_t)y * W + x;
        float v = gs[off];
        if (add) v += grad_in[off];
        grad_in[off] = v;
      }
    }
    if (shift_xy) {  // (uniform)
      const double tx = block_sum<256>(dsx, smem);
      __syncthreads();
      const double ty = block_sum<256>(dsy, smem);
      __syncthreads();
      if (threadIdx.x == 0) {
        partials[(size_t)d * partials_stride + 2 * b] = tx * (double)scale;
        partials[(size_t)d * partials_stride + 2 * b + 1] = ty * (double)scale;
      }
    }
  }
}


// ---- the same two kernels on four pixels per thread (W % 4 == 0, 16-byte aligned images) -------------------------------
// A thread owns R rows of four columns and keeps their gradient in REGISTERS over the datasets of the launch: per dataset
// it loads R + 1 rows of five source columns of gs and of the flux (a 16-byte load at a 4-byte aligned address + one float
// each -- independent loads, all in flight together) and the R centre rows of gs; the gradient is read once (if the launch
// accumulates) and written once, where rounds 4-5 sent it through memory per dataset (a read-modify-write chain: 8
// calibrated observations at 4096^2 390 us, and at 512^2 -- 32 blocks walking 16 rows x 8 datasets each -- 137 of the step's
// 253 us).  The per-pixel arithmetic (order of the four products and three sums, v = t_d + v in dataset order, every
// operation rounded on its own) is that of the scalar kernels above: the same v, bit for bit; the shift-gradient partial
// sums are per block and dataset (wave sums by xor-butterfly, waves in index order: `block_sum`'s order), so their
// grouping -- not their terms -- follows the tiling below.
// (`issue_row5` / `finish_row5`: the same five pixels by unconditional loads, jd_common.h)

// Tiling of an H x W image: a block's 256 threads are `segs` row segments of `tpr` threads (a power of two <= 256: narrow
// images put several row segments into a block instead of idle threads), a thread owns R rows: the block covers
// segs * R rows of 4 * tpr columns.  R: 4 while that leaves >= 4 blocks per CU, else 2, else 1 (a 512^2 image: 256 blocks).
struct ShiftTiling {
  int tpr_log2, segs, R;
  dim3 grid;
};

static ShiftTiling shift_tiling(int H, int W) {
  ShiftTiling t;
  t.tpr_log2 = 4;
  while (t.tpr_log2 < 8 && (4 << t.tpr_log2) < W) ++t.tpr_log2;
  const int tpr = 1 << t.tpr_log2;
  t.segs = 256 / tpr;
  const int gx = (W + 4 * tpr - 1) / (4 * tpr);
  auto blocks = [&](int r) { return gx * ((H + t.segs * r - 1) / (t.segs * r)); };
  t.R = blocks(4) >= 1024 ? 4 : blocks(2) >= 1024 ? 2 : 1;
  t.grid = dim3(gx, (H + t.segs * t.R - 1) / (t.segs * t.R));
  return t;
}

struct ShiftBwdArgs {
  const float* in;
  const FftBatch* batch;  // the datasets' images and shifts (device table), or nullptr: the ONE dataset below
  const float* gs0;
  const float* shift0;
  int n_datasets;
  float* grad_in;
  int accumulate, H, W, tpr_log2;
  float scale;
  double* partials;
  size_t partials_stride;
};

constexpr int SHIFT_CHUNK = 16;  // datasets between two barriers (their wave sums wait in LDS)

template <int R>
__global__ __launch_bounds__(256) void shift_bwd4_kernel(ShiftBwdArgs a) {
#pragma clang fp contract(off)
  __shared__ double wsum[SHIFT_CHUNK][256 / 64][2];
  const int H = a.H, W = a.W;
  const int tpr = 1 << a.tpr_log2, seg = threadIdx.x >> a.tpr_log2, segs = 256 >> a.tpr_log2;
  const int x = 4 * (blockIdx.x * tpr + (threadIdx.x & (tpr - 1)));
  const int y0 = ((int)blockIdx.y * segs + seg) * R;
  const bool live = x < W && y0 < H;
  const size_t b = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float acc[R][4];
#pragma unroll
  for (int k = 0; k < R; ++k) {
    float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.accumulate && live && y0 + k < H) q = *reinterpret_cast<const float4*>(a.grad_in + (size_t)(y0 + k) * W + x);
    acc[k][0] = q.x, acc[k][1] = q.y, acc[k][2] = q.z, acc[k][3] = q.w;
  }
  for (int d0 = 0; d0 < a.n_datasets; d0 += SHIFT_CHUNK) {
    const int dn = min(SHIFT_CHUNK, a.n_datasets - d0);
    // the shift of dataset d + 1 is requested (two scalar loads: pointer, then value) while dataset d is processed: read at
    // the top of its own iteration it was a flat load with a full wait in front of the iteration's 24 row loads
    const float* next_xy = a.batch ? a.batch->shift_xy[d0] : a.shift0;
    float next_sx = next_xy ? cld(next_xy) : 0.f, next_sy = next_xy ? cld(next_xy + 1) : 0.f;
    for (int dd = 0; dd < dn; ++dd) {
      const int d = d0 + dd;
      const float* gs = a.batch ? a.batch->gshift[d] : a.gs0;
      const float* shift_xy = next_xy;
      const float shift_x = next_sx, shift_y = next_sy;
      if (dd + 1 < dn) {
        next_xy = a.batch ? a.batch->shift_xy[d + 1] : a.shift0;
        next_sx = next_xy ? cld(next_xy) : 0.f, next_sy = next_xy ? cld(next_xy + 1) : 0.f;
      }
      const bool add = a.accumulate || d > 0;
      if (shift_xy) {  // (uniform)
        double dsx = 0.0, dsy = 0.0;
        if (live) {
          const ShiftGeom g = shift_geom_of(shift_x, shift_y, a.scale);
          const float w00 = g.wx0 * g.wy0, w10 = g.wx1 * g.wy0, w01 = g.wx0 * g.wy1, w11 = g.wx1 * g.wy1;
          // gs rows y - fy - 1 (prev) and y - fy (cur) at columns x - fx - 1 ..; flux rows y + fy (north), y + fy + 1 (south)
          Row5 gr[R + 1], fl[R + 1];
          float4 go[R];
          {
            Raw5 gr_raw[R + 1], fl_raw[R + 1];
#pragma unroll
            for (int k = 0; k <= R; ++k) {
              gr_raw[k] = issue_row5(gs, H, W, y0 + k - g.fy - 1, x - g.fx - 1);
              fl_raw[k] = issue_row5(a.in, H, W, y0 + k + g.fy, x + g.fx);
            }
#pragma unroll
            for (int k = 0; k < R; ++k) {  // (unconditional too: the last existing row stands in for one below the image)
              const float4 q = gld4(gs + ((size_t)min(y0 + k, H - 1) * W + x));
              go[k] = y0 + k < H ? q : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int k = 0; k <= R; ++k) gr[k] = finish_row5(gr_raw[k]), fl[k] = finish_row5(fl_raw[k]);
          }
#pragma unroll
          for (int k = 0; k < R; ++k) {
            const float gov[4] = {go[k].x, go[k].y, go[k].z, go[k].w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              float v = gr[k + 1].v[i + 1] * w00 + gr[k + 1].v[i] * w10 + gr[k].v[i + 1] * w01 + gr[k].v[i] * w11;
              if (add) v += acc[k][i];
              acc[k][i] = v;
              const float nw = fl[k].v[i], ne = fl[k].v[i + 1], sw = fl[k + 1].v[i], se = fl[k + 1].v[i + 1];
              dsx += (double)(gov[i] * ((ne - nw) * g.wy0 + (se - sw) * g.wy1));
              dsy += (double)(gov[i] * ((sw - nw) * g.wx0 + (se - ne) * g.wx1));
            }
          }
        }
        dsx = wave_sum(dsx), dsy = wave_sum(dsy);
        if (lane == 0) wsum[dd][wave][0] = dsx, wsum[dd][wave][1] = dsy;
      } else if (live) {  // a dataset without a shift contributes its image as it is
#pragma unroll
        for (int k = 0; k < R; ++k) {
          if (y0 + k >= H) continue;
          const float4 q = gld4(gs + (size_t)(y0 + k) * W + x);
          const float qv[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            float v = qv[i];
            if (add) v += acc[k][i];
            acc[k][i] = v;
          }
        }
      }
    }
    __syncthreads();
    if ((int)threadIdx.x < 2 * dn) {  // thread 2 dd + i: sum i of dataset d0 + dd over the waves, in wave order
      const int dd = threadIdx.x >> 1, i = threadIdx.x & 1, d = d0 + dd;
      const float* shift_xy = a.batch ? a.batch->shift_xy[d] : a.shift0;
      if (shift_xy) {
        double total = 0.0;
#pragma unroll
        for (int w = 0; w < 256 / 64; ++w) total += wsum[dd][w][i];
        a.partials[(size_t)d * a.partials_stride + 2 * b + i] = total * (double)a.scale;
      }
    }
    __syncthreads();
  }
  if (live) {
#pragma unroll
    for (int k = 0; k < R; ++k)
      if (y0 + k < H)
        *reinterpret_cast<float4*>(a.grad_in + (size_t)(y0 + k) * W + x) = make_float4(acc[k][0], acc[k][1], acc[k][2], acc[k][3]);
  }
}

static int launch_shift_bwd4(const ShiftBwdArgs& base, int* n_blocks, hipStream_t stream) {
  ShiftBwdArgs a = base;
  const ShiftTiling t = shift_tiling(a.H, a.W);
  a.tpr_log2 = t.tpr_log2;
  *n_blocks = t.grid.x * t.grid.y;
  if (t.R == 4) shift_bwd4_kernel<4><<<t.grid, 256, 0, stream>>>(a);
  else if (t.R == 2) shift_bwd4_kernel<2><<<t.grid, 256, 0, stream>>>(a);
  else shift_bwd4_kernel<1><<<t.grid, 256, 0, stream>>>(a);
  JD_LAUNCH_CHECK();
  return JD_OK;
}

static bool shift_vec_ok(const void* a, const void* b, const void* c, int W) {
  return W % 4 == 0 && W >= 8 && ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c)) & 15) == 0;
}

int launch_shift_bwd_batch(const float* in, const FftBatch* batch, int n_datasets, float* grad_in, int accumulate, int H, int W,
                           float scale, double* partials, size_t partials_stride, int* n_blocks, hipStream_t stream) {
  // (the batch table's images are hipMalloc'ed by the plan: 256-byte aligned)
  ProfScope prof(JD_KERNEL_SHIFT, stream);
  if (shift_vec_ok(in, grad_in, nullptr, W)) {
    ShiftBwdArgs a{in, batch, nullptr, nullptr, n_datasets, grad_in, accumulate, H, W, 0, scale, partials, partials_stride};
    return launch_shift_bwd4(a, n_blocks, stream);
  }
  dim3 grid((W + 255) / 256, (H + SHIFT_ROWS - 1) / SHIFT_ROWS);
  *n_blocks = grid.x * grid.y;
  shift_bwd_batch_kernel<<<grid, 256, 0, stream>>>(in, batch, n_datasets, grad_in, accumulate, H, W, scale, partials, partials_stride);
  JD_LAUNCH_CHECK();
  return JD_OK;
}

__global__ __launch_bounds__(256) void finalize_multi_kernel(const double* __restrict__ partials, int n_blocks, int n_out,
                                                             double scale, float* __restrict__ out, int accumulate) {
  __shared__ double smem[256 / 64];
  for (int i = 0; i < n_out; ++i) {
    double acc = 0.0;
    for (int b = threadIdx.x; b < n_blocks; b += 256) acc += partials[(size_t)n_out * b + i];
    const double total = block_sum<256>(acc, smem);
    if (threadIdx.x == 0) {
      double v = scale * total;
      if (accumulate) v += (double)out[i];
      out[i] = (float)v;
    }
    __syncthreads();
  }
}

// finalize_multi_kernel for SEVERAL datasets in one launch: block d sums the n_blocks x 2 partial sums of dataset d
// (partials + d * stride) into out[d][0 .. 1] -- the arithmetic and order of finalize_multi_kernel per dataset.
constexpr int FINALIZE_BATCH_MAX = 16;
struct FinalizeBatchArgs {
  const double* partials;
  size_t stride;  // doubles between the partial sums of consecutive datasets
  int n_blocks;
  float* out[FINALIZE_BATCH_MAX];  // nullable entries
};

__global__ __launch_bounds__(256) void finalize_multi_batch_kernel(FinalizeBatchArgs a) {
  __shared__ double smem[256 / 64];
  float* out = nullptr;
#pragma unroll
  for (int d = 0; d < FINALIZE_BATCH_MAX; ++d)  // (compile-time indices into the by-value argument array)
    if ((int)blockIdx.x == d) out = a.out[d];
  if (!out) return;  // (block-uniform)
  const double* partials = a.partials + (size_t)blockIdx.x * a.stride;
  for (int i = 0; i < 2; ++i) {
    double acc = 0.0;
    for (int b = threadIdx.x; b < a.n_blocks; b += 256) acc += partials[(size_t)2 * b + i];
    const double total = block_sum<256>(acc, smem);
    if (threadIdx.x == 0) out[i] = (float)(1.0 * total);
    __syncthreads();
  }
}

int launch_finalize_multi_batch(const double* partials, size_t stride, int n_blocks, int n_datasets, float* const* out,
                                hipStream_t stream) {
  if (n_datasets < 1 || n_datasets > FINALIZE_BATCH_MAX) return fail(JD_ERR_INVALID, "finalize_multi_batch: %d datasets", n_datasets);
  FinalizeBatchArgs a{};
  a.partials = partials, a.stride = stride, a.n_blocks = n_blocks;
  bool any = false;
  for (int d = 0; d < n_datasets; ++d) a.out[d] = out[d], any = any || out[d];
  if (!any) return JD_OK;
  finalize_multi_batch_kernel<<<n_datasets, 256, 0, stream>>>(a);
  JD_LAUNCH_CHECK();
  return JD_OK;
}

int launch_shift_fwd(const float* in, float* out, int H, int W, const float* shift_xy, float scale, hipStream_t stream) {
  dim3 grid((W + 255) / 256, H);
  ProfScope prof(JD_KERNEL_SHIFT, stream);
  shift_fwd_kernel<<<grid, 256, 0, stream>>>(in, out, H, W, shift_xy, scale);
  JD_LAUNCH_CHECK();
  return JD_OK;
}

int shift_bwd_max_blocks(int H, int W) {  // partial-sum pairs a launch may write: the larger of the two tilings
  const ShiftTiling t = shift_tiling(H, W);
  const int scalar = ((W + 255) / 256) * ((H + SHIFT_ROWS - 1) / SHIFT_ROWS), vec = (int)(t.grid.x * t.grid.y);
  return scalar > vec ? scalar : vec;
}

int launch_shift_bwd(const float* in, const float* gs, float* grad_in, int accumulate, int H, int W,
                     const float* shift_xy, float scale, double* partials, int* n_blocks, hipStream_t stream) {
  ProfScope prof(JD_KERNEL_SHIFT, stream);
  if (shift_vec_ok(in, gs, grad_in, W)) {
    ShiftBwdArgs a{in, nullptr, gs, shift_xy, 1, grad_in, accumulate, H, W, 0, scale, partials, 0};
    return launch_shift_bwd4(a, n_blocks, stream);
  }
  dim3 grid((W + 255) / 256, (H + SHIFT_ROWS - 1) / SHIFT_ROWS);
  *n_blocks = grid.x * grid.y;
  shift_bwd_kernel<<<grid, 256, 0, stream>>>(in, gs, grad_in, accumulate, H, W, shift_xy, scale, partials);
  JD_LAUNCH_CHECK();
  return JD_OK;
}

int launch_finalize_multi(const double* partials, int n_blocks, int n_out, double scale, float* out, int accumulate,
                          hipStream_t stream) {
  finalize_multi_kernel<<<1, 256, 0, stream>>>(partials, n_blocks, n_out, scale, out, accumulate);
  JD_LAUNCH_CHECK();
  return JD_OK;
}

}  // namespace jd
