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
#include <algorithm>
#include <cstring>
#include <cmath>
#include <cstdlib>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "jd_common.h"
#include "kernels.h"

namespace jd {

namespace {

#ifndef JD_SEP_WAVES_POISSON
#define JD_SEP_WAVES_POISSON 6  // waves per SIMD the fused forward + Poisson kernel is compiled for (80 registers)
#endif
#ifndef JD_SEP_WAVES_MULTI
#define JD_SEP_WAVES_MULTI 4    // the multi-component variant: 128 registers (132 uncapped = three waves; 96 spill badly)
#endif
#ifndef JD_SEP_WAVES_OTHER
#define JD_SEP_WAVES_OTHER 1    // (no register cap for the other variants)
#endif
constexpr int TY = 32, TX = 64, THREADS = 256, STAGE_BATCH = 3;
typedef float v2f __attribute__((ext_vector_type(2)));

struct SepArgs {
  const float* in;
  const float* in_scale;
  const float* op;    // operator buffer: [0] = rank, taps of this direction start at `taps_off`
  float* out;
  const float* out_scale;
  int H, W, tiles_x, n_tiles;
  int khp, kwp, oy0, ox0, rpairs, pitch, taps_off;  // (khp, kwp, oy0, ox0: of the TRIMMED taps, see launch_sep)
  int tap_stride, tu_off, tv_off;  // stored taps per rank (row + column block); first trimmed row / column tap within them
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
  // batches: flux components per dataset (POISSON: summed after the per-component clip, npred.py:191,254-261) and the
  // component whose gradient an adjoint launch produces
  int n_comp, comp;
  int alias;       // the row-pass image shares the window's LDS (see the kernel)
  // adjoint batch, first component: block d < n_batch also finalises the loss of dataset d from the forward launch's
  // partial sums (fin_partials[d * n_tiles + tile]) -- finalize_rows_kernel's summation order, no launch of its own
  const double* fin_partials;
  double fin_scale;
  // single dataset (n_batch == 0): block 0 of the adjoint launch finalises its loss the same way (finalize_sum_kernel's
  // summation order): *fin_out = fin_scale * sum(fin_partials[0 .. n_tiles)) + fin_offset
  float* fin_out;
  double fin_offset;
  int interleave;  // POISSON batches: the datasets of a tile are neighbours in the launch order (else dataset-major)
  // POISSON batches: the flux images of the components.  Kernel arguments, not table entries: a fit alternates between
  // two flux buffers, and a table that changes every step would be re-uploaded (synchronously) every step.  Read with
  // constant indices only (see above).
  const float* in_c1;
  const float* in_c2;
  const float* in_c3;
  int* guard;  // host-mapped flag: an operator contradicted the rank (or the trimmed tap window) the host launched it for
  unsigned window;  // sep_pack_support of the trimmed tap window [tu, tu + khp) x [tv, tv + kwp) of the launch; 0: the plan's full window
  int dir;          // 0 forward taps, 1 adjoint taps (header word 2 + dir holds the operator's own support)
};

// LDS images:
//   win  [rpairs][pitch][2]  the input window with two image rows interleaved per column, so that one ds_read_b128
//                            yields the operand pairs (row 2p, row 2p+1) of two columns for v_pk_fma_f32
//   hbuf [2 * rpairs][TX]    the row-pass result, plain row-major: the column pass packs two neighbouring x
// POISSON: the forward model of ONE component with no up-sampling ends here -- clip, + background, Poisson NLL and
// its gradient are computed from the convolution while it is still in registers (the arithmetic of
// poisson_fused_kernel, statement for statement), so the convolution image is neither written nor read back.
// Batches (a.n_batch > 0, several datasets that share the geometry and the input image layout):
//   * POISSON: the block index selects (tile, dataset) -- one launch for all forward models of a joint step; the block walks
//     over the dataset's flux components (own flux image, exposure and PSF each), clips each convolution, adds them
//     up in component order and writes one masked gradient image per component;
//   * otherwise (the adjoint): every block walks over ALL datasets and adds their contributions in dataset order in
//     registers, so the gradient image is read and written once instead of once per dataset (same additions in the
//     same order as the per-dataset launches: same bits).
// MULTI (POISSON batches only): more than one flux component per dataset; a compile-time switch so that the common
// one-component launch carries none of the component loop.
template <bool VEC, bool IN_SCALE, bool POISSON, bool MULTI = false>
__global__ __launch_bounds__(THREADS, POISSON ? (MULTI ? JD_SEP_WAVES_MULTI : JD_SEP_WAVES_POISSON) : JD_SEP_WAVES_OTHER) void sep_conv_kernel(SepArgs a) {
  extern __shared__ float4 lds4[];
  float* win = reinterpret_cast<float*>(lds4);
  // a.alias: the row-pass image overwrites the window (rank-1 operators, at most one row-pass item per thread): the
  // items are computed into registers, a barrier retires the window, then they are stored -- 19 KB instead of 32 KB of
  // LDS per block
  float* hbuf = a.alias ? win : win + a.rpairs * a.pitch * 2;
  float* taps = win + a.rpairs * a.pitch * 2 + (a.alias ? 0 : 2 * a.rpairs * TX);  // per rank: the stored row taps, then the column taps
  const int tid = threadIdx.x;
  const int tap_stride = a.tap_stride;
  const int nrows = 2 * a.rpairs;

  // consecutive tiles on one XCD (blockIdx % 8) are neighbours in the image: their halos hit in that XCD's L2.
  // POISSON batch: dataset-major order, or (a.interleave, tuning) the datasets of one tile as neighbours
  const int per_xcd = (a.n_tiles + 7) / 8;
  const int in_xcd = blockIdx.x / 8;
  int dsel = 0, t_in_xcd = in_xcd;  // dataset of a POISSON batch block, tile within the XCD's share
  if (POISSON && a.n_batch > 0) {
    if (a.interleave)
      dsel = in_xcd % a.n_batch, t_in_xcd = in_xcd / a.n_batch;
    else
      dsel = in_xcd / per_xcd, t_in_xcd = in_xcd % per_xcd;  // (tuning: dataset-major order)
  }
  const int tile = (blockIdx.x % 8) * per_xcd + t_in_xcd;
  if (!POISSON && a.fin_partials && (int)blockIdx.x < (a.n_batch > 0 ? a.n_batch : 1)) {  // (block-uniform)
    __shared__ double fin_red[THREADS / 64];
    const double* row = a.fin_partials + (size_t)blockIdx.x * a.n_tiles;
    double acc = 0.0;
    for (int i = tid; i < a.n_tiles; i += THREADS) acc += row[i];
    const double total = block_sum<THREADS>(acc, fin_red);
    if (tid == 0) {
      if (a.n_batch > 0)
        a.table->loss_out[blockIdx.x][0] = (float)(a.fin_scale * total + (double)a.table->loss_offset[blockIdx.x]);
      else
        a.fin_out[0] = (float)(a.fin_scale * total + a.fin_offset);
    }
    __syncthreads();
  }
  if (tile >= a.n_tiles) return;
  const int Y0 = (tile / a.tiles_x) * TY, X0 = (tile % a.tiles_x) * TX;
  const int gy0 = Y0 + a.oy0, gx0 = X0 + a.ox0;

  // column-pass item of this thread: outputs (y0 .. y0+3, x and x+1)
  static_assert((TX / 2) * (TY / 4) == THREADS, "tile / block shape");
  const int cx = (tid % (TX / 2)) * 2, cy = (tid / (TX / 2)) * 4;
  const int gx = X0 + cx;

  // units of work of this block: POISSON batch -- the components of dataset blockIdx.y; adjoint batch -- the datasets
  const int n_comp = (MULTI || !POISSON) && a.n_batch > 0 ? a.n_comp : 1;
  const int u_end = a.n_batch > 0 ? (POISSON ? n_comp : a.n_batch) : 1;
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

  v2f nsum[4];        // POISSON: sum over the components of clip(conv_c, 0)
  unsigned mask = 0;  // POISSON: bit 8 c + 2 row + e = (conv_c >= 0) of this thread's pixel (row, e)
  v2f opa[4], opb[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) nsum[c] = v2f{0.f, 0.f};

  for (int u = 0; u < u_end; ++u) {
    const int d = a.n_batch > 0 ? (POISSON ? dsel : u) : 0;
    const int slot = d * n_comp + (POISSON ? u : a.comp);  // table entry of (dataset, component)
    const float* in = a.n_batch > 0 && !POISSON ? a.table->g[slot] : a.in;
    if (MULTI) in = u == 0 ? a.in : u == 1 ? a.in_c1 : u == 2 ? a.in_c2 : a.in_c3;
    const float* in_scale = a.n_batch > 0 ? a.table->scale[slot] : a.in_scale;
    const float* op = a.n_batch > 0 ? a.table->op[slot] : a.op;
    const float* out_scale = a.n_batch > 0 ? a.table->scale[slot] : a.out_scale;  // adjoint batch: the exposure of (d, c)
    const float* background = a.n_batch > 0 ? a.table->bkg[d] : a.background;
    const float* counts = a.n_batch > 0 ? a.table->cnt[d] : a.counts;
    const int rank = (int)op[0];
    if (a.alias && rank != 1 && tid == 0) *a.guard = 1;  // launched for rank 1 (host registry): reported at the next call
    if (a.window && tid == 0) {  // launched on a trimmed window: the operator's own support must lie inside it
      const unsigned own = __float_as_uint(op[2 + a.dir]), w = a.window;
      if ((own & 255u) < (w & 255u) || (own >> 8 & 255u) > (w >> 8 & 255u) || (own >> 16 & 255u) < (w >> 16 & 255u) ||
          (own >> 24) > (w >> 24))
        *a.guard = 1;
    }
    if (u > 0) __syncthreads();  // the previous unit is done with the LDS images
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
    v2f acc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      acc[c] = v2f{0.f, 0.f};
      if (POISSON && u > 0) continue;  // background and counts: once per dataset
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
      const float* tu = taps + r * tap_stride + a.tu_off;
      const float* tv = taps + r * tap_stride + a.tv_off;
      // ---- row pass, two rows at once: hbuf[row][x] = sum_t tv[t] * win[row][x + t] -------------------------
      const int n_items = a.rpairs * (TX / 8);
      for (int item0 = 0; item0 < n_items; item0 += THREADS) {  // (alias: one trip)
        const int item = item0 + tid;
        const bool has = item < n_items;
        const int rp = item / (TX / 8), x0 = (item % (TX / 8)) * 8;
        v2f h[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) h[c] = v2f{0.f, 0.f};
        if (has) {
          const float* w = win + (rp * a.pitch + x0) * 2;
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
        }
        if (a.alias) __syncthreads();  // every item has read its part of the window: the results may overwrite it
        if (has) {
          float* d0 = hbuf + (2 * rp) * TX + x0;
#pragma unroll
          for (int c = 0; c < 8; ++c) d0[c] = h[c].x, d0[TX + c] = h[c].y;
        }
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
      // clip per component, sum in component order (0 + clip(conv_0) + clip(conv_1) + ...), remember where conv_c >= 0
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          nsum[c][e] += fmaxf(acc[c][e], 0.f);
          mask |= (acc[c][e] >= 0.f ? 1u : 0u) << (8 * u + 2 * c + e);
        }
      if (u + 1 < u_end) continue;
      // ---- epilogue: n = sum_c max(conv_c, 0) + b;  loss += n - c log(n + eps);  g_c = (1 - c / (n + eps)) / N where
      // conv_c >= 0
      __shared__ double red[THREADS / 64];
      double local = 0.0;
      v2f n4[4], g4[4];
      // the thread's eight loss terms are summed in fp32 before they join the fp64 block sum (poisson_point: the
      // arithmetic every Poisson pass of the library shares)
      float local_f = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int gy = Y0 + cy + c;
        n4[c] = g4[c] = v2f{0.f, 0.f};
        if (gy >= a.H || gx >= a.W) continue;
        const bool two = gx + 1 < a.W;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const float n = nsum[c][e] + opa[c][e];  // the un-convolved background comes last (npred.py:191,254-261)
          float term, g;
          poisson_point(n, opb[c][e], a.eps, a.inv_n, term, g);
          if (e == 0 || two) local_f += term;
          n4[c][e] = n;
          g4[c][e] = g;
        }
      }
      local = (double)local_f;
      for (int k = 0; k < n_comp; ++k) {
        float* g_out = a.n_batch > 0 ? a.table->g[d * n_comp + k] : a.out;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int gy = Y0 + cy + c;
          if (gy >= a.H || gx >= a.W) continue;
          const size_t off = (size_t)gy * a.W + gx;
          const bool two = gx + 1 < a.W;
          // clamp backward: the gradient passes where conv_k >= 0
          const float g0 = (mask >> (8 * k + 2 * c)) & 1u ? g4[c][0] : 0.f, g1 = (mask >> (8 * k + 2 * c + 1)) & 1u ? g4[c][1] : 0.f;
          if (VEC) {
            if (a.write_grad) *reinterpret_cast<v2f*>(g_out + off) = v2f{g0, g1};
            if (k == 0 && a.npred_out) *reinterpret_cast<v2f*>(a.npred_out + off) = n4[c];
          } else {
            if (a.write_grad) {
              g_out[off] = g0;
              if (two) g_out[off + 1] = g1;
            }
            if (k == 0 && a.npred_out) {
              a.npred_out[off] = n4[c][0];
              if (two) a.npred_out[off + 1] = n4[c][1];
            }
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

SepGeom sep_geom(int kh, int kw, int oy, int ox, bool adjoint) {
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

// Rank of every operator buffer this process has built, by device address (jd_conv_psf_spectrum registers it after the
// upload).  The kernels that assume rank 1 (LDS aliasing of the tile kernel, the walk kernels) are only launched for
// buffers found here with rank 1 and still check op[0] on the device: a mismatch (the buffer was overwritten behind the
// library's back) raises a host-mapped flag that the next launch reports as an error.
static std::mutex g_rank_mutex;
static std::unordered_map<const void*, SepOpInfo> g_op_info;
static int* g_guard_host = nullptr;
static int* g_guard_dev = nullptr;

void sep_register_operator(const void* op_dev, const SepOpInfo& info) {
  std::lock_guard<std::mutex> lock(g_rank_mutex);
  g_op_info[op_dev] = info;
}

void sep_forget_operator(const void* op_dev) {
  std::lock_guard<std::mutex> lock(g_rank_mutex);
  g_op_info.erase(op_dev);
}

int sep_operator_rank(const void* op_dev) {
  std::lock_guard<std::mutex> lock(g_rank_mutex);
  auto it = g_op_info.find(op_dev);
  return it == g_op_info.end() ? 0 : it->second.rank;
}

bool sep_operator_info(const void* op_dev, SepOpInfo* info) {
  std::lock_guard<std::mutex> lock(g_rank_mutex);
  auto it = g_op_info.find(op_dev);
  if (it == g_op_info.end()) return false;
  *info = it->second;
  return true;
}

int sep_guard_check(int** guard_dev) {
  if (!g_guard_host) {
    JD_HIP(hipHostMalloc(reinterpret_cast<void**>(&g_guard_host), sizeof(int), hipHostMallocMapped));
    *g_guard_host = 0;
    void* mapped = nullptr;
    JD_HIP(hipHostGetDevicePointer(&mapped, g_guard_host, 0));
    g_guard_dev = static_cast<int*>(mapped);
  }
  *guard_dev = g_guard_dev;
  if (*g_guard_host) {
    *g_guard_host = 0;
    return fail(JD_ERR_INVALID, "separable convolution: an operator buffer held another rank, or taps outside the support / "
                "frame it was registered with, on the device (operator buffers are immutable: was it overwritten after "
                "jd_conv_psf_spectrum?  jd_conv_operator_forget() the address, or build the new operator into it with "
                "jd_conv_psf_spectrum)");
  }
  return JD_OK;
}

// Host image of the operator buffer for a factorised PSF (see sep_conv_operator_floats); *info <- rank and the support
// of the stored taps
int sep_build_operator(const float* psf, int kh, int kw, int oy, int ox, double tol, std::vector<float>* op, SepOpInfo* info) {
  std::vector<double> u, v;
  const int rank = sep_factorize(psf, kh, kw, tol, &u, &v);
  if (rank == 0) return 0;
  op->assign(sep_conv_operator_floats(), 0.f);
  (*op)[0] = (float)rank;
  const size_t half = (op->size() - 4) / 2;
  SepOpInfo oi;
  oi.rank = rank;
  for (int adjoint = 0; adjoint < 2; ++adjoint) {
    const SepGeom g = sep_geom(kh, kw, oy, ox, adjoint != 0);
    float* dst = op->data() + 4 + adjoint * half;
    int ulo = g.khp, uhi = 0, vlo = g.kwp, vhi = 0;
    for (int r = 0; r < rank; ++r) {
      float* tu = dst + (size_t)r * (g.khp + g.kwp);
      float* tv = tu + g.khp;
      for (int t = 0; t < kh; ++t) tu[t] = (float)u[(size_t)r * kh + (adjoint ? t : kh - 1 - t)];
      for (int t = 0; t < kw; ++t) tv[g.shiftx + t] = (float)v[(size_t)r * kw + (adjoint ? t : kw - 1 - t)];
      for (int t = 0; t < g.khp; ++t)
        if (tu[t] != 0.f) ulo = t < ulo ? t : ulo, uhi = t + 1 > uhi ? t + 1 : uhi;
      for (int t = 0; t < g.kwp; ++t)
        if (tv[t] != 0.f) vlo = t < vlo ? t : vlo, vhi = t + 1 > vhi ? t + 1 : vhi;
    }
    if (uhi <= ulo) ulo = 0, uhi = 1;  // (all taps rounded to zero in fp32: an empty support is still one tap)
    if (vhi <= vlo) vlo = 0, vhi = 1;
    oi.ulo[adjoint] = ulo, oi.uhi[adjoint] = uhi, oi.vlo[adjoint] = vlo, oi.vhi[adjoint] = vhi;
  }
  // the operator's own record in its header (kernels.h: SepOpInfo): what the kernels' guard compares with the window /
  // frame the host chose from the registry
  const int frame = walk_info_frame(oi, kh, kw, oy, ox);
  (*op)[1] = frame ? (float)frame : 99.f;
  for (int adjoint = 0; adjoint < 2; ++adjoint) {
    const unsigned packed = sep_pack_support(oi.ulo[adjoint], oi.uhi[adjoint], oi.vlo[adjoint], oi.vhi[adjoint]);
    memcpy(&(*op)[2 + adjoint], &packed, sizeof(packed));
  }
  if (info) *info = oi;
  return rank;
}

namespace {
// rank1: every operator of the launch is registered with rank 1 (sep_operator_rank)
// ops[0 .. n_ops): the operators the launch reads.  Their NON-ZERO taps (SepOpInfo, union over the operators; an operator
// the library does not know: all of them) decide the window: PSFs that were embedded in a larger array of zeros cost
// what their own size costs.  Trimmed in steps of 4 taps (the kernel reads taps and window in 16-byte pieces); skipping
// a zero tap changes no bit (h + 0 * w = h).  Option JD_SEP_NO_TRIM: the plan's full (kh, kw).
int launch_sep(SepArgs a, int kh, int kw, int oy, int ox, int adjoint, bool poisson, hipStream_t stream, bool rank1,
               const float* const* ops, int n_ops, bool batch_aligned = true) {
  if (!sep_conv_supported(kh, kw))
    return fail(JD_ERR_INVALID, "separable convolution: PSF %dx%d exceeds %dx%d", kh, kw, SEP_MAX_K, SEP_MAX_K);
  SepGeom g = sep_geom(kh, kw, oy, ox, adjoint != 0);
  a.tiles_x = (a.W + TX - 1) / TX;
  a.n_tiles = a.tiles_x * ((a.H + TY - 1) / TY);
  a.tap_stride = g.khp + g.kwp, a.tu_off = 0, a.tv_off = g.khp;
  if (!opt_is_set(OPT_SEP_NO_TRIM)) {
    const int dir = adjoint ? 1 : 0;
    int ulo = g.khp, uhi = 0, vlo = g.kwp, vhi = 0;
    bool known = n_ops > 0;
    for (int i = 0; i < n_ops && known; ++i) {
      SepOpInfo info;
      known = sep_operator_info(ops[i], &info);
      if (!known) break;
      ulo = std::min(ulo, info.ulo[dir]), uhi = std::max(uhi, info.uhi[dir]);
      vlo = std::min(vlo, info.vlo[dir]), vhi = std::max(vhi, info.vhi[dir]);
    }
    if (known && uhi > ulo && vhi > vlo && uhi <= g.khp && vhi <= g.kwp) {
      const int au = ulo / 4 * 4, av = vlo / 4 * 4;
      const int khp = (uhi + 3) / 4 * 4 - au, kwp = (vhi + 3) / 4 * 4 - av;
      a.tu_off = au, a.tv_off = g.khp + av;
      a.window = sep_pack_support(au, au + khp, av, av + kwp), a.dir = dir;
      g.oy0 += au, g.ox0 += av, g.khp = khp, g.kwp = kwp;
      g.rpairs = (TY + g.khp) / 2, g.pitch = TX + g.kwp + 2;
    }
  }
  a.khp = g.khp, a.kwp = g.kwp, a.oy0 = g.oy0, a.ox0 = g.ox0, a.rpairs = g.rpairs, a.pitch = g.pitch;
  a.taps_off = 4 + (adjoint ? (int)((sep_conv_operator_floats() - 4) / 2) : 0);
  // rank-1 operators only and at most one row-pass item per thread: the row-pass image may share the window's LDS ->
  // 19 KB instead of 32 KB per block, a sixth block per CU for the fused forward + Poisson launch (option
  // JD_SEP_NO_ALIAS: testing)
  int rc = sep_guard_check(&a.guard);
  if (rc) return rc;
  a.alias = rank1 && g.rpairs * (TX / 8) <= THREADS && !opt_is_set(OPT_SEP_NO_ALIAS) ? 1 : 0;
  size_t lds = ((size_t)2 * g.rpairs * (g.pitch + (a.alias ? 0 : TX)) + (size_t)SEP_MAX_RANK * a.tap_stride) * sizeof(float);
  {  // tuning: caps the blocks per CU
    const size_t want = (size_t)opt_value(poisson ? OPT_SEP_FWD_MIN_LDS : OPT_SEP_ADJ_MIN_LDS, 0);
    if (want > lds && want <= 160 * 1024) lds = want;
  }
  const int blocks = ((a.n_tiles + 7) / 8) * 8;
  auto aligned = [](const void* p) { return p == nullptr || (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  bool vec = a.W % 4 == 0 && aligned(a.in) && aligned(a.in_scale) && aligned(a.out) && aligned(a.out_scale) &&
             aligned(a.background) && aligned(a.counts) && aligned(a.npred_out);
  vec = vec && (a.n_batch == 0 || batch_aligned);
  const bool in_scale = a.n_batch > 0 ? poisson : a.in_scale != nullptr;  // batches: forward scales its input, adjoint its output
  const int variant1 = (poisson ? 4 : 0) + (vec ? 2 : 0) + (in_scale ? 1 : 0);
  void (*const kernels[10])(SepArgs) = {
      sep_conv_kernel<false, false, false>, sep_conv_kernel<false, true, false>, sep_conv_kernel<true, false, false>,
      sep_conv_kernel<true, true, false>,   sep_conv_kernel<false, false, true>, sep_conv_kernel<false, true, true>,
      sep_conv_kernel<true, false, true>,   sep_conv_kernel<true, true, true>,
      sep_conv_kernel<false, true, true, true>, sep_conv_kernel<true, true, true, true>};
  const bool multi = poisson && a.n_batch > 0 && a.n_comp > 1;  // batches scale their input: IN_SCALE is set
  const int variant = multi ? (vec ? 9 : 8) : variant1;
  auto kernel = kernels[variant];
  static size_t lds_set[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  if (lds > 64 * 1024 && lds > lds_set[variant]) {
    JD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    lds_set[variant] = lds;
  }
  ProfScope prof(poisson ? JD_KERNEL_POISSON_FUSED : JD_KERNEL_SEP_CONV, stream);  // the fused launch IS the Poisson pass
  hipLaunchKernelGGL(kernel, dim3(blocks * (poisson && a.n_batch > 0 ? a.n_batch : 1)), dim3(THREADS), lds, stream, a);
  JD_LAUNCH_CHECK();
  return JD_OK;
}
}  // namespace

int sep_conv_tiles(int H, int W) { return ((W + TX - 1) / TX) * ((H + TY - 1) / TY); }

// adjoint == 0: out (+)= coef * out_scale * conv_same(in * in_scale, psf)    [crop offset (oy, ox)]
// adjoint != 0: out (+)= coef * out_scale * corr_same(in * in_scale, psf)    (the transpose of the above)
int launch_sep_conv(const float* in, const float* in_scale, const float* op, float* out, const float* out_scale, int H,
                    int W, int kh, int kw, int oy, int ox, int adjoint, float coef, int accumulate,
                    hipStream_t stream, const SepLossFold* fold, int* fold_done) {
  if (fold_done) *fold_done = 0;
  {
    const int rc = walk_conv(in, in_scale, op, out, out_scale, H, W, kh, kw, oy, ox, adjoint, coef, accumulate, stream);
    if (rc != JD_WALK_NOT_TAKEN) return rc;
  }
  SepArgs a{};
  a.in = in, a.in_scale = in_scale, a.op = op, a.out = out, a.out_scale = out_scale;
  a.H = H, a.W = W, a.coef = coef, a.accumulate = accumulate;
  // (the fold sums n_tiles partial sums: only when the forward launch was this kernel's own)
  if (fold && adjoint && fold->partials && fold->count == sep_conv_tiles(H, W)) {
    a.fin_partials = fold->partials, a.fin_scale = fold->scale, a.fin_out = fold->out, a.fin_offset = fold->offset;
    if (fold_done) *fold_done = 1;
  }
  return launch_sep(a, kh, kw, oy, ox, adjoint, false, stream, sep_operator_rank(op) == 1, &op, 1);
}

// Forward model of one component fused with the Poisson pass: conv = conv_same(in * in_scale, psf) stays in
// registers; g_out (if write_grad) = masked d loss / d conv, npred_out (nullable) = clip(conv) + background,
// partials[tile] = block sums of n - c log(n + eps) (*n_partials of them).
int launch_sep_conv_poisson(const float* in, const float* in_scale, const float* op, float* g_out, int H, int W, int kh,
                            int kw, int oy, int ox, const float* background, const float* counts, float* npred_out,
                            double* partials, float eps, float inv_n, int write_grad, int* n_partials,
                            hipStream_t stream) {
  {
    const int rc = walk_conv_poisson(in, in_scale, op, g_out, H, W, kh, kw, oy, ox, background, counts, npred_out, partials,
                                     eps, inv_n, write_grad, n_partials, stream);
    if (rc != JD_WALK_NOT_TAKEN) return rc;
  }
  SepArgs a{};
  a.in = in, a.in_scale = in_scale, a.op = op, a.out = g_out;
  a.H = H, a.W = W, a.coef = 1.f;
  a.background = background, a.counts = counts, a.npred_out = npred_out, a.partials = partials;
  a.eps = eps, a.inv_n = inv_n, a.write_grad = write_grad;
  *n_partials = sep_conv_tiles(H, W);
  return launch_sep(a, kh, kw, oy, ox, 0, true, stream, sep_operator_rank(op) == 1, &op, 1);
}

static bool table_aligned(const SepBatchTable& t, int n, int n_comp) {
  auto ok = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  bool all = true;
  for (int d = 0; d < n; ++d) all = all && ok(t.bkg[d]) && ok(t.cnt[d]);
  for (int i = 0; i < n * n_comp; ++i) all = all && ok(t.scale[i]) && ok(t.g[i]);
  return all;
}
static bool table_rank1(const SepBatchTable& t, int entries) {
  for (int i = 0; i < entries; ++i)
    if (sep_operator_rank(t.op[i]) != 1) return false;
  return true;
}
static int check_batch(int n, int n_comp) {
  if (n < 1 || n > SEP_MAX_BATCH) return fail(JD_ERR_INVALID, "separable batch: %d datasets not in [1, %d]", n, SEP_MAX_BATCH);
  if (n_comp < 1 || n_comp > SEP_BATCH_MAX_COMP)
    return fail(JD_ERR_INVALID, "separable batch: %d components not in [1, %d]", n_comp, SEP_BATCH_MAX_COMP);
  return JD_OK;
}

// All forward models + Poisson passes of a joint step in ONE launch (grid.y = dataset): dataset d convolves the
// n_comp flux images flux[c] (x table.scale[d * n_comp + c]), writes the masked g of component c into
// table.g[d * n_comp + c] (if write_grad) and its block sums into partials[d * tiles + tile].
// `table_dev` is the device copy of `table`.
int launch_sep_conv_poisson_batch(int n, int n_comp, const float* const* flux, const SepBatchTable& table,
                                  const SepBatchTable* table_dev, int H, int W, int kh, int kw, int oy, int ox,
                                  double* partials, float eps, float inv_n, int write_grad, int* n_partials,
                                  hipStream_t stream) {
  int rc = check_batch(n, n_comp);
  if (rc) return rc;
  rc = n_comp == 1 ? walk_conv_poisson_batch(n, flux[0], table, table_dev, H, W, kh, kw, oy, ox, partials, eps, inv_n,
                                             write_grad, n_partials, stream)
                   : walk_conv_poisson_batch_multi(n, n_comp, flux, table, table_dev, H, W, kh, kw, oy, ox, partials, eps,
                                                   inv_n, write_grad, n_partials, stream);
  if (rc != JD_WALK_NOT_TAKEN) return rc;
  *n_partials = sep_conv_tiles(H, W);
  SepArgs a{};
  a.in = flux[0], a.in_c1 = n_comp > 1 ? flux[1] : nullptr, a.in_c2 = n_comp > 2 ? flux[2] : nullptr;
  a.in_c3 = n_comp > 3 ? flux[3] : nullptr;
  a.H = H, a.W = W, a.coef = 1.f, a.partials = partials;
  a.eps = eps, a.inv_n = inv_n, a.write_grad = write_grad, a.n_batch = n, a.n_comp = n_comp, a.table = table_dev;
  a.op = table.op[0], a.in_scale = table.scale[0];
  // dataset-major launch order by default; JD_SEP_INTERLEAVE=1 makes the datasets of a tile neighbours in an XCD's
  // launch order (the flux window then comes from L2 for all but the first): measured neutral at 8 observations
  // (152-154 us either way) -- the Infinity Cache already serves the repeated flux reads
  a.interleave = opt_is_set(OPT_SEP_INTERLEAVE) ? 1 : 0;
  bool flux_aligned = true;
  for (int c = 0; c < n_comp; ++c) flux_aligned = flux_aligned && (reinterpret_cast<uintptr_t>(flux[c]) & 15) == 0;
  return launch_sep(a, kh, kw, oy, ox, 0, true, stream, table_rank1(table, n * n_comp), table.op, n * n_comp,
                    flux_aligned && table_aligned(table, n, n_comp));
}

// grad (+)= coef * sum_d scale[d, comp] * corr_same(g[d, comp], psf_(d, comp)): one launch per component, the datasets
// are added in order in registers (bit-identical to n accumulate launches), the gradient image is read and written once.
int launch_sep_conv_adjoint_batch(int n, int n_comp, int comp, const SepBatchTable& table, const SepBatchTable* table_dev,
                                  float* grad, int H, int W, int kh, int kw, int oy, int ox, float coef, int accumulate,
                                  hipStream_t stream, const double* fin_partials, double fin_scale, int fin_count,
                                  int* fin_done) {
  int rc = check_batch(n, n_comp);
  if (rc) return rc;
  if (comp < 0 || comp >= n_comp) return fail(JD_ERR_INVALID, "separable batch: component %d not in [0, %d)", comp, n_comp);
  if (fin_done) *fin_done = 0;
  {
    int folded = 0;
    rc = walk_conv_adjoint_batch(n, n_comp, comp, table, table_dev, grad, H, W, kh, kw, oy, ox, coef, accumulate, stream,
                                 fin_partials, fin_scale, fin_count, &folded);
    if (rc != JD_WALK_NOT_TAKEN) {
      if (fin_done) *fin_done = folded;
      return rc;
    }
  }
  SepArgs a{};
  a.out = grad, a.H = H, a.W = W, a.coef = coef, a.accumulate = accumulate, a.n_batch = n, a.table = table_dev;
  a.n_comp = n_comp, a.comp = comp;
  // (the fold of the tile kernel sums a.n_tiles partial sums per dataset: only when the forward launch was its own)
  if (fin_partials && fin_count == sep_conv_tiles(H, W)) {
    a.fin_partials = fin_partials, a.fin_scale = fin_scale;
    if (fin_done) *fin_done = 1;
  }
  a.in = table.g[comp], a.op = table.op[comp];
  // (the launch reads the operators of component `comp` only, but trims by all of them: one window geometry per step)
  return launch_sep(a, kh, kw, oy, ox, 1, false, stream, table_rank1(table, n * n_comp), table.op, n * n_comp,
                    table_aligned(table, n, n_comp));
}

}  // namespace jd
