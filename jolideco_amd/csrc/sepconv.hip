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
// Kernel: one 32 x 64 output tile per block of 256 threads, ~30 KB of LDS (5 blocks per CU).  The input window
// (tile + halo, times the exposure, zero outside the image) is staged once in LDS; the row pass writes an
// intermediate (window rows x 64) image back to LDS, the column pass reads it and adds into per-thread
// accumulators that persist over the ranks; the epilogue applies coef * out_scale and stores / accumulates.  Both
// passes are register blocked (8 outputs x 4 taps / 4 x 4) and use packed fp32 FMAs (v_pk_fma_f32: the row pass on
// two image rows, the column pass on two neighbouring columns), so one LDS read feeds ~5 FMAs; the taps stay
// runtime values (no template per PSF size).
#include <cmath>
#include <vector>

#include "jd_common.h"
#include "kernels.h"

namespace jd {

namespace {

constexpr int TY = 32, TX = 64, THREADS = 256, STAGE_BATCH = 3;
typedef float v2f __attribute__((ext_vector_type(2)));

struct SepGeom {
  int khp, kwp;    // padded tap counts (multiples of 4): rows / columns
  int oy0, ox0;    // image offset of window (row 0, col 0) relative to the tile origin
  int shiftx;      // zero taps prepended to the column taps so that ox0 is a multiple of 4
  int rpairs;      // window row PAIRS: ceil((TY + khp - 1) / 2)
  int pitch;       // window columns per row (>= TX + kwp), pitch % 4 == 2 (bank spread of the row-pair reads)
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
  g.rpairs = (TY + g.khp) / 2;  // khp is a multiple of 4: (TY + khp - 1 + 1) / 2
  g.pitch = TX + g.kwp + 2;
  return g;
}

struct SepArgs {
  const float* in;
  const float* in_scale;
  const float* op;    // operator buffer: [0] = rank, taps of this direction start at `taps_off`
  float* out;
  const float* out_scale;
  int H, W, tiles_x, n_tiles;
  int khp, kwp, oy0, ox0, rpairs, pitch, taps_off;
  float coef;
  int accumulate;
  // fused Poisson epilogue (POISSON kernels only): `out` receives g = d loss / d conv instead of the convolution
  const float* background;
  const float* counts;
  float* npred_out;   // nullable
  double* partials;   // one per tile
  float eps, inv_n;
  int write_grad;
  // batch of datasets (n_batch > 0): per-dataset exposure (in_scale of the forward model = out_scale of the adjoint),
  // operator, background, counts, g work image (forward: output, adjoint: input)
  // (a pointer table in DEVICE memory: indexing arrays inside the by-value kernel argument with a run-time dataset
  // index makes hipcc copy the whole argument block to scratch, which halved the speed of every variant)
  int n_batch;
  const SepBatchTable* table;
};

// LDS images:
//   win  [rpairs][pitch][2]  the input window with two image rows interleaved per column, so that one ds_read_b128
//                            yields the operand pairs (row 2p, row 2p+1) of two columns for v_pk_fma_f32
//   hbuf [2 * rpairs][TX]    the row-pass result, plain row-major: the column pass packs two neighbouring x
// POISSON: the forward model of ONE component with no up-sampling ends here -- clip, + background, Poisson NLL and
// its gradient are computed from the convolution while it is still in registers (the arithmetic of
// poisson_fused_kernel, statement for statement), so the convolution image is neither written nor read back.
// Batches (a.n_batch > 0, several datasets that share the geometry and the input image layout):
//   * POISSON: blockIdx.y selects the dataset -- one launch for all forward models of a joint step;
//   * otherwise (the adjoint): every block walks over ALL datasets and adds their contributions in dataset order in
//     registers, so the gradient image is read and written once instead of once per dataset (same additions in the
//     same order as the per-dataset launches: same bits).
template <bool VEC, bool IN_SCALE, bool POISSON>
__global__ __launch_bounds__(THREADS) void sep_conv_kernel(SepArgs a) {
  extern __shared__ float4 lds4[];
  float* win = reinterpret_cast<float*>(lds4);
  float* hbuf = win + a.rpairs * a.pitch * 2;
  float* taps = hbuf + 2 * a.rpairs * TX;  // per rank: khp row taps then kwp column taps
  const int tid = threadIdx.x;
  const int tap_stride = a.khp + a.kwp;
  const int nrows = 2 * a.rpairs;

  // consecutive tiles on one XCD (blockIdx % 8) are neighbours in the image: their halos hit in that XCD's L2
  const int per_xcd = (a.n_tiles + 7) / 8;
  const int tile = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
  if (tile >= a.n_tiles) return;
  const int Y0 = (tile / a.tiles_x) * TY, X0 = (tile % a.tiles_x) * TX;
  const int gy0 = Y0 + a.oy0, gx0 = X0 + a.ox0;

  // column-pass item of this thread: outputs (y0 .. y0+3, x and x+1)
  static_assert((TX / 2) * (TY / 4) == THREADS, "tile / block shape");
  const int cx = (tid % (TX / 2)) * 2, cy = (tid / (TX / 2)) * 4;
  const int gx = X0 + cx;

  const int d_begin = a.n_batch > 0 ? (POISSON ? (int)blockIdx.y : 0) : 0;
  const int d_end = a.n_batch > 0 ? (POISSON ? d_begin + 1 : a.n_batch) : 1;
  // a pair of pixels of an (H, W) image at `off`: aligned float2 on the VEC path
  auto load2 = [&](const float* img, size_t off, float other) {
    if (VEC) return *reinterpret_cast<const v2f*>(img + off);
    return v2f{img[off], gx + 1 < a.W ? img[off + 1] : other};
  };

  v2f res[4];  // running output of the plain / adjoint epilogue: [out +] sum_d coef * out_scale_d * conv_d
  if (!POISSON) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      res[c] = v2f{0.f, 0.f};
      const int gy = Y0 + cy + c;
      if (a.accumulate && gy < a.H && gx < a.W) res[c] = load2(a.out, (size_t)gy * a.W + gx, 0.f);
    }
  }

  for (int d = d_begin; d < d_end; ++d) {
    const float* in = a.n_batch > 0 && !POISSON ? a.table->g[d] : a.in;
    const float* in_scale = a.n_batch > 0 ? a.table->scale[d] : a.in_scale;
    const float* op = a.n_batch > 0 ? a.table->op[d] : a.op;
    const float* out_scale = a.n_batch > 0 ? a.table->scale[d] : a.out_scale;  // adjoint batch: the exposure of dataset d
    const float* background = a.n_batch > 0 ? a.table->bkg[d] : a.background;
    const float* counts = a.n_batch > 0 ? a.table->cnt[d] : a.counts;
    const int rank = (int)op[0];
    if (d > d_begin) __syncthreads();  // the previous dataset is done with the LDS images
    for (int i = tid; i < rank * tap_stride; i += THREADS) taps[i] = op[a.taps_off + i];

    // ---- stage the window: image * in_scale, zero outside ---------------------------------------------------
    if (VEC) {
      // One item = 4 columns of BOTH rows of a row pair: two float4 loads (+ two of the exposure), two
      // ds_write_b128 of the interleaved (row 2p, row 2p+1) pairs.  All loads of a thread are issued before its
      // first LDS store, so a tile exposes one memory latency; lanes outside the image load element 0 and select
      // zero (no branches).
      const int nv = (TX + a.kwp) / 4;  // float4 per row actually needed
      const int total = a.rpairs * nv;
      const int step_r = THREADS / nv, step_c = THREADS % nv;  // (row pair, float4 column) advance of i += THREADS
      int rp = tid / nv, cv = tid - rp * nv;
      for (int base = tid; base < total; base += STAGE_BATCH * THREADS) {
        float4 va[STAGE_BATCH], vb[STAGE_BATCH], sa[STAGE_BATCH], sb[STAGE_BATCH];
        int dst[STAGE_BATCH];
        bool oka[STAGE_BATCH], okb[STAGE_BATCH];
#pragma unroll
        for (int b = 0; b < STAGE_BATCH; ++b) {
          const int gy = gy0 + 2 * rp, gxs = gx0 + 4 * cv;
          const bool inx = rp < a.rpairs && gxs >= 0 && gxs < a.W;  // gxs, W multiples of 4: never partial
          dst[b] = rp < a.rpairs ? (rp * a.pitch + 4 * cv) * 2 : -1;
          oka[b] = inx && gy >= 0 && gy < a.H;
          okb[b] = inx && gy + 1 >= 0 && gy + 1 < a.H;
          const size_t offa = oka[b] ? (size_t)gy * a.W + gxs : 0, offb = okb[b] ? (size_t)(gy + 1) * a.W + gxs : 0;
          va[b] = *reinterpret_cast<const float4*>(in + offa);
          vb[b] = *reinterpret_cast<const float4*>(in + offb);
          if (IN_SCALE) {
            sa[b] = *reinterpret_cast<const float4*>(in_scale + offa);
            sb[b] = *reinterpret_cast<const float4*>(in_scale + offb);
          }
          rp += step_r, cv += step_c;
          if (cv >= nv) cv -= nv, ++rp;
        }
#pragma unroll
        for (int b = 0; b < STAGE_BATCH; ++b) {
          float4 x = va[b], y = vb[b];
          if (IN_SCALE) {
            x.x *= sa[b].x, x.y *= sa[b].y, x.z *= sa[b].z, x.w *= sa[b].w;
            y.x *= sb[b].x, y.y *= sb[b].y, y.z *= sb[b].z, y.w *= sb[b].w;
          }
          if (!oka[b]) x = make_float4(0.f, 0.f, 0.f, 0.f);
          if (!okb[b]) y = make_float4(0.f, 0.f, 0.f, 0.f);
          if (dst[b] >= 0) {
            float4* dq = reinterpret_cast<float4*>(win + dst[b]);
            dq[0] = make_float4(x.x, y.x, x.y, y.y);
            dq[1] = make_float4(x.z, y.z, x.w, y.w);
          }
        }
      }
    } else {
      const int nc = TX + a.kwp;
      const int total = nrows * nc;
      for (int i = tid; i < total; i += THREADS) {
        const int r = i / nc, c = i - r * nc;
        const int gy = gy0 + r, gxs = gx0 + c;
        float v = 0.f;
        if (gy >= 0 && gy < a.H && gxs >= 0 && gxs < a.W) {
          const size_t off = (size_t)gy * a.W + gxs;
          v = in[off];
          if (IN_SCALE) v *= in_scale[off];
        }
        win[((r >> 1) * a.pitch + c) * 2 + (r & 1)] = v;
      }
    }

    // epilogue operands are requested now, so that their latency hides behind the two passes:
    // plain / adjoint: (out_scale, -);  POISSON: (background, counts)
    v2f acc[4], opa[4], opb[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      acc[c] = v2f{0.f, 0.f};
      opa[c] = v2f{1.f, 1.f};
      opb[c] = v2f{0.f, 0.f};
      const int gy = Y0 + cy + c;
      if (gy >= a.H || gx >= a.W) continue;
      const size_t off = (size_t)gy * a.W + gx;
      if (POISSON) {
        opa[c] = load2(background, off, 0.f);
        opb[c] = load2(counts, off, 0.f);
      } else if (out_scale) {
        opa[c] = load2(out_scale, off, 1.f);
      }
    }

    for (int r = 0; r < rank; ++r) {
      __syncthreads();  // window (r == 0) / previous column pass done with hbuf (r > 0); taps visible
      const float* tu = taps + r * tap_stride;
      const float* tv = tu + a.khp;
      // ---- row pass, two rows at once: hbuf[row][x] = sum_t tv[t] * win[row][x + t] -------------------------
      for (int item = tid; item < a.rpairs * (TX / 8); item += THREADS) {
        const int rp = item / (TX / 8), x0 = (item % (TX / 8)) * 8;
        const float* w = win + (rp * a.pitch + x0) * 2;
        v2f h[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) h[c] = v2f{0.f, 0.f};
        for (int q = 0; q < a.kwp; q += 4) {
          const float4 t4 = *reinterpret_cast<const float4*>(tv + q);
          const float tt[4] = {t4.x, t4.y, t4.z, t4.w};
          v2f ww[12];
#pragma unroll
          for (int k = 0; k < 6; ++k) {
            const float4 two = *reinterpret_cast<const float4*>(w + (q + 2 * k) * 2);
            ww[2 * k] = v2f{two.x, two.y};
            ww[2 * k + 1] = v2f{two.z, two.w};
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const v2f t2 = v2f{tt[e], tt[e]};
#pragma unroll
            for (int c = 0; c < 8; ++c) h[c] = __builtin_elementwise_fma(t2, ww[c + e], h[c]);
          }
        }
        float* d0 = hbuf + (2 * rp) * TX + x0;
#pragma unroll
        for (int c = 0; c < 8; ++c) d0[c] = h[c].x, d0[TX + c] = h[c].y;
      }
      __syncthreads();
      // ---- column pass, two columns at once: acc[y][x] += sum_t tu[t] * hbuf[y + t][x] -----------------------
      const float* hcol = hbuf + cy * TX + cx;
      for (int q = 0; q < a.khp; q += 4) {
        const float4 t4 = *reinterpret_cast<const float4*>(tu + q);
        const float tt[4] = {t4.x, t4.y, t4.z, t4.w};
        v2f hh[7];
#pragma unroll
        for (int k = 0; k < 7; ++k) hh[k] = *reinterpret_cast<const v2f*>(hcol + (q + k) * TX);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const v2f t2 = v2f{tt[e], tt[e]};
#pragma unroll
          for (int c = 0; c < 4; ++c) acc[c] = __builtin_elementwise_fma(t2, hh[c + e], acc[c]);
        }
      }
    }

    if (POISSON) {
      // ---- epilogue: n = max(conv, 0) + b;  loss += n - c log(n + eps);  g = (1 - c / (n + eps)) / N where conv >= 0
      __shared__ double red[THREADS / 64];
      float* g_out = a.n_batch > 0 ? a.table->g[d] : a.out;
      double local = 0.0;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int gy = Y0 + cy + c;
        if (gy >= a.H || gx >= a.W) continue;
        const size_t off = (size_t)gy * a.W + gx;
        const bool two = gx + 1 < a.W;
        float n2[2], g2[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const float conv = acc[c][e], b = opa[c][e], cnt = opb[c][e];
          const float n = fmaxf(conv, 0.f) + b;  // clip, then the un-convolved background (npred.py:191,254-261)
          const float ne = n + a.eps;
          if (e == 0 || two) local += (double)(n - cnt * logf(ne));
          const float g = (1.f - cnt / ne) * a.inv_n;
          n2[e] = n;
          g2[e] = conv >= 0.f ? g : 0.f;  // clamp backward: passes where conv >= 0
        }
        if (VEC) {
          if (a.write_grad) *reinterpret_cast<v2f*>(g_out + off) = v2f{g2[0], g2[1]};
          if (a.npred_out) *reinterpret_cast<v2f*>(a.npred_out + off) = v2f{n2[0], n2[1]};
        } else {
          if (a.write_grad) {
            g_out[off] = g2[0];
            if (two) g_out[off + 1] = g2[1];
          }
          if (a.npred_out) {
            a.npred_out[off] = n2[0];
            if (two) a.npred_out[off + 1] = n2[1];
          }
        }
      }
      local = wave_sum(local);
      if ((tid & 63) == 0) red[tid >> 6] = local;
      __syncthreads();
      if (tid == 0) a.partials[(size_t)d * a.n_tiles + tile] = (red[0] + red[1]) + (red[2] + red[3]);
    } else {
#pragma unroll
      for (int c = 0; c < 4; ++c) res[c] = res[c] + a.coef * acc[c] * opa[c];
    }
  }
  if (POISSON) return;

  // ---- epilogue: out = [out +] sum_d coef * out_scale_d * conv_d ------------------------------------------------
  if (gx >= a.W) return;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int gy = Y0 + cy + c;
    if (gy >= a.H) break;
    const size_t off = (size_t)gy * a.W + gx;
    if (VEC) {
      *reinterpret_cast<v2f*>(a.out + off) = res[c];
    } else {
      a.out[off] = res[c].x;
      if (gx + 1 < a.W) a.out[off + 1] = res[c].y;
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

namespace {
int launch_sep(SepArgs a, int kh, int kw, int oy, int ox, int adjoint, bool poisson, hipStream_t stream,
               bool batch_aligned = true) {
  if (!sep_conv_supported(kh, kw))
    return fail(JD_ERR_INVALID, "separable convolution: PSF %dx%d exceeds %dx%d", kh, kw, SEP_MAX_K, SEP_MAX_K);
  const SepGeom g = sep_geom(kh, kw, oy, ox, adjoint != 0);
  a.tiles_x = (a.W + TX - 1) / TX;
  a.n_tiles = a.tiles_x * ((a.H + TY - 1) / TY);
  a.khp = g.khp, a.kwp = g.kwp, a.oy0 = g.oy0, a.ox0 = g.ox0, a.rpairs = g.rpairs, a.pitch = g.pitch;
  a.taps_off = 4 + (adjoint ? (int)((sep_conv_operator_floats() - 4) / 2) : 0);
  const size_t lds = ((size_t)2 * g.rpairs * (g.pitch + TX) + (size_t)SEP_MAX_RANK * (g.khp + g.kwp)) * sizeof(float);
  const int blocks = ((a.n_tiles + 7) / 8) * 8;
  auto aligned = [](const void* p) { return p == nullptr || (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  bool vec = a.W % 4 == 0 && aligned(a.in) && aligned(a.in_scale) && aligned(a.out) && aligned(a.out_scale) &&
             aligned(a.background) && aligned(a.counts) && aligned(a.npred_out);
  vec = vec && (a.n_batch == 0 || batch_aligned);
  const bool in_scale = a.n_batch > 0 ? poisson : a.in_scale != nullptr;  // batches: forward scales its input, adjoint its output
  const int variant = (poisson ? 4 : 0) + (vec ? 2 : 0) + (in_scale ? 1 : 0);
  void (*const kernels[8])(SepArgs) = {
      sep_conv_kernel<false, false, false>, sep_conv_kernel<false, true, false>, sep_conv_kernel<true, false, false>,
      sep_conv_kernel<true, true, false>,   sep_conv_kernel<false, false, true>, sep_conv_kernel<false, true, true>,
      sep_conv_kernel<true, false, true>,   sep_conv_kernel<true, true, true>};
  auto kernel = kernels[variant];
  static size_t lds_set[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (lds > 64 * 1024 && lds > lds_set[variant]) {
    JD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    lds_set[variant] = lds;
  }
  ProfScope prof(poisson ? JD_KERNEL_POISSON_FUSED : JD_KERNEL_SEP_CONV, stream);  // the fused launch IS the Poisson pass
  hipLaunchKernelGGL(kernel, dim3(blocks, poisson && a.n_batch > 0 ? a.n_batch : 1), dim3(THREADS), lds, stream, a);
  JD_LAUNCH_CHECK();
  return JD_OK;
}
}  // namespace

int sep_conv_tiles(int H, int W) { return ((W + TX - 1) / TX) * ((H + TY - 1) / TY); }

// adjoint == 0: out (+)= coef * out_scale * conv_same(in * in_scale, psf)    [crop offset (oy, ox)]
// adjoint != 0: out (+)= coef * out_scale * corr_same(in * in_scale, psf)    (the transpose of the above)
int launch_sep_conv(const float* in, const float* in_scale, const float* op, float* out, const float* out_scale, int H,
                    int W, int kh, int kw, int oy, int ox, int adjoint, float coef, int accumulate,
                    hipStream_t stream) {
  SepArgs a{};
  a.in = in, a.in_scale = in_scale, a.op = op, a.out = out, a.out_scale = out_scale;
  a.H = H, a.W = W, a.coef = coef, a.accumulate = accumulate;
  return launch_sep(a, kh, kw, oy, ox, adjoint, false, stream);
}

// Forward model of one component fused with the Poisson pass: conv = conv_same(in * in_scale, psf) stays in
// registers; g_out (if write_grad) = masked d loss / d conv, npred_out (nullable) = clip(conv) + background,
// partials[tile] = block sums of n - c log(n + eps) (*n_partials of them).
int launch_sep_conv_poisson(const float* in, const float* in_scale, const float* op, float* g_out, int H, int W, int kh,
                            int kw, int oy, int ox, const float* background, const float* counts, float* npred_out,
                            double* partials, float eps, float inv_n, int write_grad, int* n_partials,
                            hipStream_t stream) {
  SepArgs a{};
  a.in = in, a.in_scale = in_scale, a.op = op, a.out = g_out;
  a.H = H, a.W = W, a.coef = 1.f;
  a.background = background, a.counts = counts, a.npred_out = npred_out, a.partials = partials;
  a.eps = eps, a.inv_n = inv_n, a.write_grad = write_grad;
  *n_partials = sep_conv_tiles(H, W);
  return launch_sep(a, kh, kw, oy, ox, 0, true, stream);
}

static bool table_aligned(const SepBatchTable& t, int n) {
  auto ok = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  bool all = true;
  for (int d = 0; d < n; ++d) all = all && ok(t.scale[d]) && ok(t.bkg[d]) && ok(t.cnt[d]) && ok(t.g[d]);
  return all;
}

// All forward models + Poisson passes of a joint step in ONE launch (grid.y = dataset): dataset d reads `flux` and
// table.scale[d], writes g into table.g[d] (if write_grad) and its block sums into partials[d * tiles + tile].
// `table_dev` is the device copy of `table`.
int launch_sep_conv_poisson_batch(int n, const float* flux, const SepBatchTable& table, const SepBatchTable* table_dev,
                                  int H, int W, int kh, int kw, int oy, int ox, double* partials, float eps, float inv_n,
                                  int write_grad, hipStream_t stream) {
  if (n < 1 || n > SEP_MAX_BATCH) return fail(JD_ERR_INVALID, "separable batch: %d datasets not in [1, %d]", n, SEP_MAX_BATCH);
  SepArgs a{};
  a.in = flux, a.H = H, a.W = W, a.coef = 1.f, a.partials = partials;
  a.eps = eps, a.inv_n = inv_n, a.write_grad = write_grad, a.n_batch = n, a.table = table_dev;
  a.op = table.op[0], a.in_scale = table.scale[0];
  return launch_sep(a, kh, kw, oy, ox, 0, true, stream, table_aligned(table, n));
}

// grad (+)= coef * sum_d scale[d] * corr_same(g[d], psf_d): one launch, the datasets are added in order in
// registers (bit-identical to n accumulate launches), the gradient image is read and written once.
int launch_sep_conv_adjoint_batch(int n, const SepBatchTable& table, const SepBatchTable* table_dev, float* grad, int H,
                                  int W, int kh, int kw, int oy, int ox, float coef, int accumulate, hipStream_t stream) {
  if (n < 1 || n > SEP_MAX_BATCH) return fail(JD_ERR_INVALID, "separable batch: %d datasets not in [1, %d]", n, SEP_MAX_BATCH);
  SepArgs a{};
  a.out = grad, a.H = H, a.W = W, a.coef = coef, a.accumulate = accumulate, a.n_batch = n, a.table = table_dev;
  a.in = table.g[0], a.op = table.op[0];
  return launch_sep(a, kh, kw, oy, ox, 1, false, stream, table_aligned(table, n));
}

}  // namespace jd
