// Separable ("low-rank PSF") 'same' convolution: when the PSF is a short sum of outer products
//     psf[i][j] = sum_{r < R} u_r[i] * v_r[j],    R <= 3
// (every sampled Gaussian is R = 1, a core + wing double Gaussian R = 2), the 2-D sum over kh*kw taps
//     out[y][x] = sum_ij psf[i][j] * in[y + oy - i][x + ox - j]                 (utils/torch.py:347-370)
// factors into a row pass and a column pass of kh + kw taps per rank.  At 17x17 that is 34 instead of 289
// multiply-adds per pixel: the kernel stops being arithmetic bound and runs at the HBM rate of one read and one
// write of the image (the MFMA Toeplitz kernel of directconv.hip is bound by the matrix cores).
//
// The factors come from a host-side cross approximation of the PSF (`sep_factorize`, exact for exactly
// low-rank kernels; it refuses anything whose residual exceeds a few fp32 ulps of the PSF sum, so a PSF that is
// not separable never takes this path).
//
// Kernel: one 32 x 128 output tile per block.  The input window (tile + halo, times the exposure, zero outside
// the image) is staged once in LDS; the row pass writes an intermediate (window rows x 128) image back to LDS, the
// column pass reads it and adds into per-thread accumulators that persist over the ranks; the epilogue applies
// coef * out_scale and stores / accumulates.  Both passes are register blocked 8 outputs x 4 taps, so one LDS
// read feeds ~3 FMAs and the taps stay runtime values (no template per PSF size).
#include <cmath>
#include <vector>

#include "jd_common.h"
#include "kernels.h"

namespace jd {

namespace {

constexpr int TY = 32, TX = 64, THREADS = 256, STAGE_BATCH = 5;

struct SepGeom {
  int khp, kwp;    // padded tap counts (multiples of 4): rows / columns
  int oy0, ox0;    // image offset of window (row 0, col 0) relative to the tile origin
  int shiftx;      // zero taps prepended to the column taps so that ox0 is a multiple of 4
  int rows, pitch; // LDS window: rows = TY + khp - 1, pitch = TX + kwp rounded so that pitch / 4 is odd
};

inline SepGeom sep_geom(int kh, int kw, int oy, int ox, bool adjoint) {
  SepGeom g{};
  // forward: out[y] = sum_t u[kh-1-t] in[y + oy - (kh-1) + t];  adjoint: out[y] = sum_t u[t] in[y - oy + t]
  g.oy0 = adjoint ? -oy : oy - (kh - 1);
  const int ox0 = adjoint ? -ox : ox - (kw - 1);
  g.shiftx = ((ox0 % 4) + 4) % 4;
  g.ox0 = ox0 - g.shiftx;
  g.khp = (kh + 3) / 4 * 4;
  g.kwp = (kw + g.shiftx + 3) / 4 * 4;
  g.rows = TY + g.khp - 1;
  g.pitch = TX + g.kwp;
  if ((g.pitch / 4) % 2 == 0) g.pitch += 4;
  return g;
}

struct SepArgs {
  const float* in;
  const float* in_scale;
  const float* op;    // operator buffer: [0] = rank, taps of this direction start at `taps_off`
  float* out;
  const float* out_scale;
  int H, W, tiles_x, n_tiles;
  int khp, kwp, oy0, ox0, rows, pitch, taps_off;
  float coef;
  int accumulate;
};

template <bool VEC>
__global__ __launch_bounds__(THREADS) void sep_conv_kernel(SepArgs a) {
  extern __shared__ float4 lds4[];
  float* win = reinterpret_cast<float*>(lds4);
  float* hbuf = win + a.rows * a.pitch;
  float* taps = hbuf + a.rows * TX;  // per rank: khp row taps then kwp column taps
  const int tid = threadIdx.x;
  const int rank = (int)a.op[0];
  const int tap_stride = a.khp + a.kwp;

  // consecutive tiles on one XCD (blockIdx % 8) are neighbours in the image: their halos hit in that XCD's L2
  const int per_xcd = (a.n_tiles + 7) / 8;
  const int tile = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
  if (tile >= a.n_tiles) return;
  const int Y0 = (tile / a.tiles_x) * TY, X0 = (tile % a.tiles_x) * TX;

  for (int i = tid; i < rank * tap_stride; i += THREADS) taps[i] = a.op[a.taps_off + i];

  // ---- stage the window: rows x (pitch) floats, image * in_scale, zero outside ---------------------------
  const int gy0 = Y0 + a.oy0, gx0 = X0 + a.ox0;
  if (VEC) {
    // all loads of a batch are issued before the first LDS store: one exposed memory latency per batch, not per load
    const int nv = (TX + a.kwp) / 4;  // float4 per row actually needed
    const int total = a.rows * nv;
    for (int base = tid; base < total; base += STAGE_BATCH * THREADS) {
      float4 v[STAGE_BATCH], sc[STAGE_BATCH];
      int dst[STAGE_BATCH];
#pragma unroll
      for (int b = 0; b < STAGE_BATCH; ++b) {
        const int i = base + b * THREADS;
        const int r = i / nv, cv = i - r * nv;
        const int gy = gy0 + r, gx = gx0 + 4 * cv;
        dst[b] = i < total ? r * a.pitch + 4 * cv : -1;
        v[b] = make_float4(0.f, 0.f, 0.f, 0.f);
        sc[b] = make_float4(1.f, 1.f, 1.f, 1.f);
        if (i < total && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) {  // gx, W multiples of 4: never partial
          const size_t off = (size_t)gy * a.W + gx;
          v[b] = *reinterpret_cast<const float4*>(a.in + off);
          if (a.in_scale) sc[b] = *reinterpret_cast<const float4*>(a.in_scale + off);
        }
      }
#pragma unroll
      for (int b = 0; b < STAGE_BATCH; ++b)
        if (dst[b] >= 0)
          *reinterpret_cast<float4*>(win + dst[b]) =
              make_float4(v[b].x * sc[b].x, v[b].y * sc[b].y, v[b].z * sc[b].z, v[b].w * sc[b].w);
    }
  } else {
    const int nc = TX + a.kwp;
    const int total = a.rows * nc;
    for (int i = tid; i < total; i += THREADS) {
      const int r = i / nc, c = i - r * nc;
      const int gy = gy0 + r, gx = gx0 + c;
      float v = 0.f;
      if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) {
        const size_t off = (size_t)gy * a.W + gx;
        v = a.in[off];
        if (a.in_scale) v *= a.in_scale[off];
      }
      win[r * a.pitch + c] = v;
    }
  }

  constexpr int COL_ITEMS = TX * (TY / 8) / THREADS;  // column-pass items (one x, 8 rows) per thread
  static_assert(COL_ITEMS * THREADS == TX * (TY / 8), "tile / block shape");
  float acc[COL_ITEMS][8];
#pragma unroll
  for (int m = 0; m < COL_ITEMS; ++m)
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[m][c] = 0.f;

  // epilogue operands are requested now, so that their latency hides behind the two passes
  float oscale[COL_ITEMS][8], oprev[COL_ITEMS][8];
#pragma unroll
  for (int m = 0; m < COL_ITEMS; ++m) {
    const int item = tid + m * THREADS;
    const int gx = X0 + item % TX, y0 = Y0 + (item / TX) * 8;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const bool ok = gx < a.W && y0 + c < a.H;
      const size_t off = ok ? (size_t)(y0 + c) * a.W + gx : 0;
      oscale[m][c] = (ok && a.out_scale) ? a.out_scale[off] : 1.f;
      oprev[m][c] = (ok && a.accumulate) ? a.out[off] : 0.f;
    }
  }

  for (int r = 0; r < rank; ++r) {
    __syncthreads();  // window (r == 0) / previous column pass done with hbuf (r > 0); taps visible
    const float* tu = taps + r * tap_stride;
    const float* tv = tu + a.khp;
    // ---- row pass: hbuf[row][x] = sum_t tv[t] * win[row][x + t] -----------------------------------------
    for (int item = tid; item < a.rows * (TX / 8); item += THREADS) {
      const int row = item / (TX / 8), x0 = (item % (TX / 8)) * 8;
      const float* w = win + row * a.pitch + x0;
      float h[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) h[c] = 0.f;
      for (int q = 0; q < a.kwp; q += 4) {
        const float4 t4 = *reinterpret_cast<const float4*>(tv + q);
        const float4 w0 = *reinterpret_cast<const float4*>(w + q);
        const float4 w1 = *reinterpret_cast<const float4*>(w + q + 4);
        const float4 w2 = *reinterpret_cast<const float4*>(w + q + 8);
        const float ww[12] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w, w2.x, w2.y, w2.z, w2.w};
        const float tt[4] = {t4.x, t4.y, t4.z, t4.w};
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int c = 0; c < 8; ++c) h[c] = fmaf(tt[e], ww[c + e], h[c]);
      }
      float4* dst = reinterpret_cast<float4*>(hbuf + row * TX + x0);
      dst[0] = make_float4(h[0], h[1], h[2], h[3]);
      dst[1] = make_float4(h[4], h[5], h[6], h[7]);
    }
    __syncthreads();
    // ---- column pass: acc[y][x] += sum_t tu[t] * hbuf[y + t][x] ------------------------------------------
#pragma unroll
    for (int m = 0; m < COL_ITEMS; ++m) {
      const int item = tid + m * THREADS;
      const int x = item % TX, y0 = (item / TX) * 8;
      const float* hcol = hbuf + y0 * TX + x;
      for (int q = 0; q < a.khp; q += 4) {
        const float4 t4 = *reinterpret_cast<const float4*>(tu + q);
        const float tt[4] = {t4.x, t4.y, t4.z, t4.w};
        float hh[11];
#pragma unroll
        for (int k = 0; k < 11; ++k) hh[k] = hcol[(q + k) * TX];
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int c = 0; c < 8; ++c) acc[m][c] = fmaf(tt[e], hh[c + e], acc[m][c]);
      }
    }
  }

  // ---- epilogue: out (+)= coef * out_scale * acc ---------------------------------------------------------
#pragma unroll
  for (int m = 0; m < COL_ITEMS; ++m) {
    const int item = tid + m * THREADS;
    const int gx = X0 + item % TX, y0 = Y0 + (item / TX) * 8;
    if (gx >= a.W) continue;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const int gy = y0 + c;
      if (gy >= a.H) break;
      const size_t off = (size_t)gy * a.W + gx;
      a.out[off] = oprev[m][c] + a.coef * acc[m][c] * oscale[m][c];
    }
  }
}

}  // namespace

bool sep_conv_supported(int kh, int kw) { return kh >= 1 && kw >= 1 && kh <= SEP_MAX_K && kw <= SEP_MAX_K; }

size_t sep_conv_operator_floats() {
  // [rank, 3 unused] + 2 directions x SEP_MAX_RANK x (row taps + column taps), taps padded to (SEP_MAX_K + 4)
  return 4 + 2 * (size_t)SEP_MAX_RANK * 2 * (SEP_MAX_K + 4);
}

// Cross approximation with full pivoting in double precision: psf ~= sum_r u_r v_r^T.  Returns the smallest
// rank R <= SEP_MAX_RANK whose residual satisfies sum|psf - approx| <= tol * sum|psf|, or 0 if there is none.
int sep_factorize(const float* psf, int kh, int kw, double tol, std::vector<double>* u_out, std::vector<double>* v_out) {
  std::vector<double> res((size_t)kh * kw);
  double norm = 0.0;
  for (size_t i = 0; i < res.size(); ++i) {
    res[i] = psf[i];
    norm += std::fabs(res[i]);
  }
  if (!(norm > 0.0) || !std::isfinite(norm)) return 0;
  std::vector<double> us, vs;
  for (int r = 0; r < SEP_MAX_RANK; ++r) {
    size_t piv = 0;
    for (size_t i = 1; i < res.size(); ++i)
      if (std::fabs(res[i]) > std::fabs(res[piv])) piv = i;
    const double p = res[piv];
    if (p == 0.0) break;
    const int pi = (int)(piv / kw), pj = (int)(piv % kw);
    std::vector<double> u(kh), v(kw);
    for (int i = 0; i < kh; ++i) u[i] = res[(size_t)i * kw + pj];
    for (int j = 0; j < kw; ++j) v[j] = res[(size_t)pi * kw + j] / p;
    double left = 0.0;
    for (int i = 0; i < kh; ++i)
      for (int j = 0; j < kw; ++j) {
        res[(size_t)i * kw + j] -= u[i] * v[j];
        left += std::fabs(res[(size_t)i * kw + j]);
      }
    us.insert(us.end(), u.begin(), u.end());
    vs.insert(vs.end(), v.begin(), v.end());
    if (left <= tol * norm) {
      if (u_out) *u_out = us;
      if (v_out) *v_out = vs;
      return r + 1;
    }
  }
  return 0;
}

// Host image of the operator buffer for a factorised PSF (see sep_conv_operator_floats).
int sep_build_operator(const float* psf, int kh, int kw, int oy, int ox, double tol, std::vector<float>* op) {
  std::vector<double> u, v;
  const int rank = sep_factorize(psf, kh, kw, tol, &u, &v);
  if (rank == 0) return 0;
  op->assign(sep_conv_operator_floats(), 0.f);
  (*op)[0] = (float)rank;
  const size_t half = (op->size() - 4) / 2;
  for (int adjoint = 0; adjoint < 2; ++adjoint) {
    const SepGeom g = sep_geom(kh, kw, oy, ox, adjoint != 0);
    float* dst = op->data() + 4 + adjoint * half;
    for (int r = 0; r < rank; ++r) {
      float* tu = dst + (size_t)r * (g.khp + g.kwp);
      float* tv = tu + g.khp;
      for (int t = 0; t < kh; ++t) tu[t] = (float)u[(size_t)r * kh + (adjoint ? t : kh - 1 - t)];
      for (int t = 0; t < kw; ++t) tv[g.shiftx + t] = (float)v[(size_t)r * kw + (adjoint ? t : kw - 1 - t)];
    }
  }
  return rank;
}

// adjoint == 0: out (+)= coef * out_scale * conv_same(in * in_scale, psf)    [crop offset (oy, ox)]
// adjoint != 0: out (+)= coef * out_scale * corr_same(in * in_scale, psf)    (the transpose of the above)
int launch_sep_conv(const float* in, const float* in_scale, const float* op, float* out, const float* out_scale, int H,
                    int W, int kh, int kw, int oy, int ox, int adjoint, float coef, int accumulate,
                    hipStream_t stream) {
  if (!sep_conv_supported(kh, kw))
    return fail(JD_ERR_INVALID, "separable convolution: PSF %dx%d exceeds %dx%d", kh, kw, SEP_MAX_K, SEP_MAX_K);
  const SepGeom g = sep_geom(kh, kw, oy, ox, adjoint != 0);
  SepArgs a{};
  a.in = in, a.in_scale = in_scale, a.op = op, a.out = out, a.out_scale = out_scale;
  a.H = H, a.W = W;
  a.tiles_x = (W + TX - 1) / TX;
  a.n_tiles = a.tiles_x * ((H + TY - 1) / TY);
  a.khp = g.khp, a.kwp = g.kwp, a.oy0 = g.oy0, a.ox0 = g.ox0, a.rows = g.rows, a.pitch = g.pitch;
  a.taps_off = 4 + (adjoint ? (int)((sep_conv_operator_floats() - 4) / 2) : 0);
  a.coef = coef, a.accumulate = accumulate;
  const size_t lds = ((size_t)g.rows * (g.pitch + TX) + (size_t)SEP_MAX_RANK * (g.khp + g.kwp)) * sizeof(float);
  const int blocks = ((a.n_tiles + 7) / 8) * 8;
  auto aligned = [](const void* p) { return p == nullptr || (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  const bool vec = W % 4 == 0 && aligned(in) && aligned(in_scale);
  auto kernel = vec ? sep_conv_kernel<true> : sep_conv_kernel<false>;
  static size_t lds_set[2] = {0, 0};
  if (lds > 64 * 1024 && lds > lds_set[vec]) {
    JD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    lds_set[vec] = lds;
  }
  ProfScope prof(JD_KERNEL_SEP_CONV, stream);
  hipLaunchKernelGGL(kernel, dim3(blocks), dim3(THREADS), lds, stream, a);
  JD_LAUNCH_CHECK();
  return JD_OK;
}

}  // namespace jd
