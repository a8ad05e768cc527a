// Internal launch prototypes shared between the translation units of libjolideco_hip.so.
#pragma once
#include <vector>
#include "jd_common.h"

namespace jd {

struct PoissonArgs {
  const float* conv[JD_MAX_COMPONENTS];  // padded (Hp, Wp) convolution results, read at the crop offset
  float* g[JD_MAX_COMPONENTS];           // padded (Hp, Wp) outputs: masked d loss / d conv_c
  const float* background;
  const float* counts;
  float* npred_out;  // nullable
  double* partials;
  int n_comp, H, W, Hp, Wp, oy, ox;  // (H, W): the conv / flux grid; counts live on (H / up, W / up)
  float eps, inv_n;
  int write_grad;
  int up;  // up-sampling factor of the flux grid w.r.t. the counts grid (1 = none)
  // calibration (models/npred.py:234-237): background * exp(*log_bkg_norm); partials_b receives the
  // block sums of g * background * norm = d loss / d log_bkg_norm.  Both null without a calibration.
  const float* log_bkg_norm;
  double* partials_b;
};

int launch_pad_mul(const float* image, const float* scale, float* padded, int H, int W, int Hp, int Wp,
                   hipStream_t stream);
int launch_cmul(float2* spec, const float2* khat, size_t n, bool conj, hipStream_t stream);
int launch_poisson_fused(const PoissonArgs& a, int* n_partials, hipStream_t stream);
int launch_poisson_pooled(const PoissonArgs& a, int* n_partials, hipStream_t stream);
int poisson_fused_max_partials(int Hp, int Wp);
int launch_adjoint_epilogue(const float* corr, const float* scale, float* grad, int H, int W, int Hp,
                            int Wp, int oy, int ox, float coef, int accumulate, hipStream_t stream);
int launch_crop(const float* padded, float* out, int H, int W, int Wp, int oy, int ox, hipStream_t stream);

// sub-pixel shift of the calibration (shift.hip): out = bilinear(in, y + scale * shift[1], x + scale * shift[0])
int launch_shift_fwd(const float* in, float* out, int H, int W, const float* shift_xy, float scale, hipStream_t stream);
// grad_in (+)= adjoint(gs);  partials[2 * b + {0, 1}] = block sums of d/d shift_{x, y} (already times scale)
int launch_shift_bwd(const float* in, const float* gs, float* grad_in, int accumulate, int H, int W,
                     const float* shift_xy, float scale, double* partials, int* n_blocks, hipStream_t stream);
int shift_bwd_max_blocks(int H, int W);
// out[i] = [out[i] +] scale * sum_b partials[n_out * b + i]   (i < n_out; one fixed-order pass)
int launch_finalize_multi(const double* partials, int n_blocks, int n_out, double scale, float* out, int accumulate,
                          hipStream_t stream);

// direct (MFMA Toeplitz) convolution for small PSFs (directconv.hip)
enum { JD_CONV_FFT = 0, JD_CONV_DIRECT = 1, JD_CONV_SEPARABLE = 2 };
bool direct_conv_supported(int kh, int kw);
// split: the fp16 x 3 kernel (two-term fp16 split of both operands, 22 significant bits) instead of the fp32 MFMA one
bool direct_conv_split_supported(int kh, int kw);
size_t direct_conv_fragment_floats(int kh, int kw, int split);
int launch_toeplitz_fragments(const float* psf, float* afrag_fwd, float* afrag_adj, int kh, int kw, int split,
                              hipStream_t stream);
int launch_direct_conv(const float* in, const float* in_scale, const float* afrag, float* out, const float* out_scale,
                       int H, int W, int kh, int kw, int oy, int ox, int adjoint, float coef, int accumulate, int split,
                       hipStream_t stream);
int direct_conv_tiles(int H, int W);
int launch_direct_conv_poisson(const float* in, const float* in_scale, const float* afrag, float* g_out, int H, int W,
                               int kh, int kw, int oy, int ox, const float* background, const float* counts,
                               float* npred_out, double* partials, float eps, float inv_n, int write_grad,
                               int* n_partials, int split, hipStream_t stream);

// separable (low-rank PSF) convolution (sepconv.hip)
constexpr int SEP_MAX_K = 68, SEP_MAX_RANK = 3, SEP_MAX_BATCH = 16;
constexpr double SEP_DEFAULT_TOL = 3e-7;  // residual sum|psf - sum_r u_r v_r^T| <= tol * sum|psf|  (~5 fp32 ulps)
bool sep_conv_supported(int kh, int kw);
size_t sep_conv_operator_floats();
int sep_factorize(const float* psf_host, int kh, int kw, double tol, std::vector<double>* u, std::vector<double>* v);
int sep_build_operator(const float* psf_host, int kh, int kw, int oy, int ox, double tol, std::vector<float>* op);
int launch_sep_conv(const float* in, const float* in_scale, const float* op, float* out, const float* out_scale, int H,
                    int W, int kh, int kw, int oy, int ox, int adjoint, float coef, int accumulate, hipStream_t stream);
int sep_conv_tiles(int H, int W);
// per-dataset pointers of a batched joint step: exposure (input scale of the forward model, output scale of the
// adjoint), operator, background, counts, g work image
constexpr int SEP_BATCH_MAX_COMP = 4;  // flux components of a batched joint step (8 pixels x 4 clip masks = 32 bits per thread)
// Pointer table of a batched joint step (device memory; everything that stays the same from step to step).  Per
// (dataset d, component c) entries sit at d * n_comp + c.
struct SepBatchTable {
  const float* scale[SEP_MAX_BATCH * SEP_BATCH_MAX_COMP];   // exposure of (d, c)
  const float* op[SEP_MAX_BATCH * SEP_BATCH_MAX_COMP];      // factorised PSF of (d, c)
  float* g[SEP_MAX_BATCH * SEP_BATCH_MAX_COMP];             // masked d loss / d conv_(d, c) work image
  const float* bkg[SEP_MAX_BATCH];
  const float* cnt[SEP_MAX_BATCH];
  // loss of dataset d = loss_scale * sum(partials of d) + loss_offset[d] -> loss_out[d]: summed by block d of the first
  // adjoint launch of the step (the partial sums come from the forward launch before it), which saves the launch of a
  // finalize kernel; forward-only calls use launch_finalize_rows
  float* loss_out[SEP_MAX_BATCH];
  float loss_offset[SEP_MAX_BATCH];
};
int launch_sep_conv_poisson_batch(int n, int n_comp, const float* const* flux, const SepBatchTable& table,
                                  const SepBatchTable* table_dev, int H, int W, int kh, int kw, int oy, int ox,
                                  double* partials, float eps, float inv_n, int write_grad, hipStream_t stream);
// fin_partials != nullptr: block d < n also turns the n_tiles partial sums of dataset d into its loss (see SepBatchTable)
int launch_sep_conv_adjoint_batch(int n, int n_comp, int comp, const SepBatchTable& table, const SepBatchTable* table_dev,
                                  float* grad, int H, int W, int kh, int kw, int oy, int ox, float coef, int accumulate,
                                  hipStream_t stream, const double* fin_partials = nullptr, double fin_scale = 0.0);
// out[d][0] = scale * sum(partials[d * n .. d * n + n - 1]) + offset[d]   (one block per output, fixed order)
int launch_finalize_rows(const double* partials, int n, int n_out, double scale, const float* offset_host,
                         float* const* out, hipStream_t stream);
int launch_sep_conv_poisson(const float* in, const float* in_scale, const float* op, float* g_out, int H, int W, int kh,
                            int kw, int oy, int ox, const float* background, const float* counts, float* npred_out,
                            double* partials, float eps, float inv_n, int write_grad, int* n_partials,
                            hipStream_t stream);

// kernel timers (profile.hip): RAII bracket around one launch
int prof_begin(int kernel, hipStream_t s);
void prof_end(int slot, hipStream_t s);
struct ProfScope {
  int slot;
  hipStream_t s;
  ProfScope(int kernel, hipStream_t stream) : slot(prof_begin(kernel, stream)), s(stream) {}
  ~ProfScope() { prof_end(slot, s); }
};

}  // namespace jd
