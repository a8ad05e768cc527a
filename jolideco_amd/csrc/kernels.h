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
// grad_in (+)= sum_d shift_d^T(gs_d) in dataset order (datasets without a shift: + gs_d), and per dataset the partial sums of
// d loss / d shift_xy; batch (device memory): gshift, shift_xy per dataset
struct FftBatch;
int launch_shift_bwd_batch(const float* in, const FftBatch* batch, int n_datasets, float* grad_in, int accumulate, int H, int W,
                           float scale, double* partials, size_t partials_stride, int* n_blocks, hipStream_t stream);
int launch_finalize_multi_batch(const double* partials, size_t stride, int n_blocks, int n_datasets, float* const* out,
                                hipStream_t stream);
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

// tuning / test switches (options.hip): read from the environment once at load time, changed by jd_set_option()
enum JdOption {
  OPT_SEP_NO_ALIAS, OPT_SEP_FWD_MIN_LDS, OPT_SEP_ADJ_MIN_LDS, OPT_SEP_INTERLEAVE, OPT_SEP_NO_FUSION, OPT_SEP_WALK,
  OPT_SEP_WALK_COLS, OPT_SEP_WALK_ROWS, OPT_SEP_WALK_ADJ_COLS, OPT_SEP_WALK_ADJ_ROWS, OPT_DIRECT_FP32,
  OPT_CONV_BLOCKS_PER_CU, OPT_POISSON_ROWS, OPT_GMM_NO_HOST_STATS, OPT_GMM_BLOCK_TILES, OPT_GMM_DENSE, OPT_GMM_KSPLIT,
  OPT_GMM_SCREEN_NP, OPT_GMM_SCREEN_NO_LDS_CONSTS, OPT_GMM_SCREEN_DEBUG, OPT_GMM_SCREEN, OPT_GMM_FUSED_BWD,
  OPT_GMM_GATHER_TILED, OPT_GMM_LSE_SCREEN, OPT_GMM_WINNER_ROWS, OPT_SEP_JOINT, OPT_SEP_JOINT_ROWS, OPT_SEP_JOINT_CHUNK,
  OPT_SEP_WALK_ADJ_ALL, OPT_SEP_WALK_COST33, OPT_SEP_WALK_ROWS33, OPT_SEP_NO_TRIM, OPT_SEP_WALK_ADJ_ROWS33,
  OPT_SEP_WALK_ADJ33, OPT_FFT_NATIVE, OPT_DIRECT_AUTO_ALL, OPT_FFT_BATCH, OPT_FFT_TINY, OPT_FFT_POOL_IO, OPT_GMM_SORT_BLOCKS, OPT_GMM_GATHER_PRELOAD, OPT_COUNT
};
bool opt_is_set(int id);
int opt_value(int id, int unset_value);

// separable (low-rank PSF) convolution (sepconv.hip)
constexpr int SEP_MAX_K = 68, SEP_MAX_RANK = 3, SEP_MAX_BATCH = 16;
struct SepGeom {
  int khp, kwp;    // padded tap counts (multiples of 4): rows / columns
  int oy0, ox0;    // image offset of window (row 0, col 0) relative to the tile origin
  int shiftx;      // zero taps prepended to the column taps so that ox0 is a multiple of 4
  int rpairs;      // window row PAIRS of the tile kernel: ceil((TY + khp - 1) / 2)
  int pitch;       // window columns per row (>= TX + kwp), pitch % 4 == 2 (bank spread of the row-pair reads)
};
SepGeom sep_geom(int kh, int kw, int oy, int ox, bool adjoint);
// What the library knows about an operator buffer it built (jd_conv_psf_spectrum registers it by device address, the
// caller's jd_conv_operator_forget removes it): the rank and, per direction (0 forward, 1 adjoint), the index range
// [lo, hi) of the NON-ZERO stored taps -- row taps u within their khp block, column taps v within their kwp block (the
// shiftx zeros in front included), union over the ranks.  A PSF embedded in a larger array of zeros (datasets with
// different PSF sizes share one plan that way) has a support smaller than the plan's (kh, kw): the strip-walk kernels
// choose their frame (17 or 33 taps) per operator from it, the tile kernel trims its window per launch.
// rank 0 = a buffer the library has not seen (a copy made by the caller): such a buffer never takes a path that
// assumes its rank or support.
// An operator buffer is IMMUTABLE once built: the registry is keyed by its device address.  The buffer carries its own
// record in its four header floats -- [0] rank, [1] the strip-walk frame its taps need (17 / 33; 99: none), [2] / [3] the
// support of the forward / adjoint taps as the BIT PATTERN ulo | uhi << 8 | vlo << 16 | vhi << 24 (sep_pack_support) --
// and every kernel that trusted the registry for a narrower window or frame compares:
// a registered buffer overwritten in place by an operator of wider support (khat.copy_(other), a state load) trips the
// guard flag, and the next library call reports JD_ERR_INVALID instead of silently dropping taps.
struct SepOpInfo {
  int rank = 0;
  int ulo[2] = {0, 0}, uhi[2] = {0, 0}, vlo[2] = {0, 0}, vhi[2] = {0, 0};
};
inline unsigned sep_pack_support(int ulo, int uhi, int vlo, int vhi) {
  return (unsigned)ulo | (unsigned)uhi << 8 | (unsigned)vlo << 16 | (unsigned)vhi << 24;
}
void sep_register_operator(const void* op_dev, const SepOpInfo& info);
void sep_forget_operator(const void* op_dev);
int sep_operator_rank(const void* op_dev);
bool sep_operator_info(const void* op_dev, SepOpInfo* info);
constexpr double SEP_DEFAULT_TOL = 3e-7;  // residual sum|psf - sum_r u_r v_r^T| <= tol * sum|psf|  (~5 fp32 ulps)
bool sep_conv_supported(int kh, int kw);
size_t sep_conv_operator_floats();
int sep_factorize(const float* psf_host, int kh, int kw, double tol, std::vector<double>* u, std::vector<double>* v);
int sep_build_operator(const float* psf_host, int kh, int kw, int oy, int ox, double tol, std::vector<float>* op,
                       SepOpInfo* info = nullptr);
// (The strip-walk kernels take a launch wherever walk_conv() accepts it.  With option JD_SEP_WALK unset that depends on
// the number of (pixel, dataset, component) triples of the LAUNCH, so a batched step and the per-dataset calls it
// stands for may run different kernels and then agree to rounding only; JD_SEP_WALK = 0 / 1 makes both choose alike.)
// fold (adjoint launches of a single dataset): block 0 also turns the `count` partial sums of the forward launch into
// the dataset's loss, *out = scale * sum(partials) + offset, in finalize_sum_kernel's summation order -- one dependent
// launch less per step; *fold_done <- whether this launch did (the strip-walk kernel and a forward launch of another
// kernel's partial sums leave it to launch_finalize_sum)
struct SepLossFold {
  const double* partials;
  int count;
  double scale, offset;
  float* out;
};
int launch_sep_conv(const float* in, const float* in_scale, const float* op, float* out, const float* out_scale, int H,
                    int W, int kh, int kw, int oy, int ox, int adjoint, float coef, int accumulate, hipStream_t stream,
                    const SepLossFold* fold = nullptr, int* fold_done = nullptr);
int sep_guard_check(int** guard_dev);
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
  // one component: the datasets whose operators walk in the 17-tap frame (n17 of them, in dataset order), then those of
  // the 33-tap frame -- the order of the batched forward launch of the strip-walk kernels (walk_batch_order fills it)
  int order[SEP_MAX_BATCH];
  int n17;
  // bit d * n_comp + c: the operator of (dataset d, component c) walks in the 33-tap frame (several components: the
  // waves of a block choose their frame by it)
  unsigned long long frame33;
};
void walk_batch_order(SepBatchTable& table, int n, int n_comp, int kh, int kw, int oy, int ox);
// *n_partials <- partial sums written per dataset: partials[d * *n_partials + i]
int launch_sep_conv_poisson_batch(int n, int n_comp, const float* const* flux, const SepBatchTable& table,
                                  const SepBatchTable* table_dev, int H, int W, int kh, int kw, int oy, int ox,
                                  double* partials, float eps, float inv_n, int write_grad, int* n_partials,
                                  hipStream_t stream);
// fin_partials != nullptr: block d < n also turns the fin_count partial sums of dataset d into its loss (see
// SepBatchTable); *fin_done <- whether the launch did (a launch that cannot leaves the losses to launch_finalize_rows)
int launch_sep_conv_adjoint_batch(int n, int n_comp, int comp, const SepBatchTable& table, const SepBatchTable* table_dev,
                                  float* grad, int H, int W, int kh, int kw, int oy, int ox, float coef, int accumulate,
                                  hipStream_t stream, const double* fin_partials = nullptr, double fin_scale = 0.0,
                                  int fin_count = 0, int* fin_done = nullptr);
// true when the walk kernels take some but not all operators of a batch: the caller then runs the per-dataset calls
// (the batched and the per-dataset step must choose the same kernel for every dataset to stay bit-identical)
bool sep_batch_is_mixed(int n, int n_comp, const SepBatchTable& table, int H, int W, int kh, int kw, int oy, int ox);

// strip-walk form of the separable convolution (walkconv.hip; rank-1 PSFs whose non-zero taps fit a frame of 17 or 33
// taps): same contracts as the launch_sep_* functions above; JD_WALK_NOT_TAKEN when the case is not theirs (nothing was
// launched)
constexpr int JD_WALK_NOT_TAKEN = 1;
// geometry, size and option JD_SEP_WALK only (rank-1 operators and aligned images assumed): forward and adjoint
bool walk_takes_launch(int H, int W, int n_datasets, int kh, int kw, int oy, int ox);
// frame (17 / 33 taps) a registered operator of this plan geometry walks in, 0: none
int walk_operator_frame(const float* op, int kh, int kw, int oy, int ox);
// the same from a support record (sep_build_operator writes it into the operator's header, the kernels' guard reads it):
// 17 / 33, 0: rank > 1 or a support no frame holds
int walk_info_frame(const SepOpInfo& info, int kh, int kw, int oy, int ox);
int walk_conv(const float* in, const float* in_scale, const float* op, float* out, const float* out_scale, int H, int W,
              int kh, int kw, int oy, int ox, int adjoint, float coef, int accumulate, hipStream_t stream);
int walk_conv_poisson(const float* in, const float* in_scale, const float* op, float* g_out, int H, int W, int kh, int kw,
                      int oy, int ox, const float* background, const float* counts, float* npred_out, double* partials,
                      float eps, float inv_n, int write_grad, int* n_partials, hipStream_t stream);
int walk_conv_poisson_batch(int n, const float* flux, const SepBatchTable& table, const SepBatchTable* table_dev, int H,
                            int W, int kh, int kw, int oy, int ox, double* partials, float eps, float inv_n,
                            int write_grad, int* n_partials, hipStream_t stream);
// forward models + Poisson passes + adjoints + the sum over the datasets of a joint step in one launch (per 8 datasets)
int walk_joint_step(int n, const float* flux, const SepBatchTable& table, const SepBatchTable* table_dev, float* grad, int H,
                    int W, int kh, int kw, int oy, int ox, double* partials, float eps, float inv_n, float coef,
                    int accumulate, int* n_partials, hipStream_t stream);
int walk_conv_poisson_batch_multi(int n, int n_comp, const float* const* flux, const SepBatchTable& table,
                                  const SepBatchTable* table_dev, int H, int W, int kh, int kw, int oy, int ox,
                                  double* partials, float eps, float inv_n, int write_grad, int* n_partials,
                                  hipStream_t stream);
int walk_conv_adjoint_batch(int n, int n_comp, int comp, const SepBatchTable& table, const SepBatchTable* table_dev,
                            float* grad, int H, int W, int kh, int kw, int oy, int ox, float coef, int accumulate,
                            hipStream_t stream, const double* fin_partials, double fin_scale, int fin_count, int* fin_done);
// the same for ALL components of up to 16 datasets in ONE launch (grads[c] (+)= ...); JD_WALK_NOT_TAKEN where the
// per-component launches above are the better (or the only) choice
int walk_conv_adjoint_batch_all(int n, int n_comp, const SepBatchTable& table, const SepBatchTable* table_dev,
                                float* const* grads, int H, int W, int kh, int kw, int oy, int ox, float coef, int accumulate,
                                hipStream_t stream, const double* fin_partials, double fin_scale, int fin_count,
                                int* fin_done);
// out[d][0] = scale * sum(partials[d * n .. d * n + n - 1]) + offset[d]   (one block per output, fixed order)
int launch_finalize_rows(const double* partials, int n, int n_out, double scale, const float* offset_host,
                         float* const* out, hipStream_t stream);
int launch_sep_conv_poisson(const float* in, const float* in_scale, const float* op, float* g_out, int H, int W, int kh,
                            int kw, int oy, int ox, const float* background, const float* counts, float* npred_out,
                            double* partials, float eps, float inv_n, int write_grad, int* n_partials,
                            hipStream_t stream);

// native FFT convolution (fftnative.hip): three launches per convolution on hand-written complex FFTs of lengths
// 2^a * {1, 3, 9}; image rows packed pairwise (upper half real, lower half imaginary)
struct FftNative {
  int H = 0, W = 0, kh = 0, kw = 0, oy = 0, ox = 0, Hh = 0, Nx = 0, Ny = 0;
  float2* spec = nullptr;  // (Hh, Nx) row spectra
  float2* work = nullptr;  // (Ny, Nx) after the column pass
  float2* tw_x = nullptr;  // exp(-2 pi i m / Nx)
  float2* tw_y = nullptr;
};
bool fftn_supported(int H, int W, int kh, int kw);
int fftn_create(FftNative* n, int H, int W, int kh, int kw);
void fftn_destroy(FftNative* n);
size_t fftn_spectrum_elements(const FftNative& n);
int fftn_spectrum(const FftNative& n, const float* psf, float2* khat, hipStream_t stream);
int fftn_conv(const FftNative& n, const float* in, const float* in_scale, const float2* khat, float* out, const float* out_scale,
              int adjoint, float coef, int accumulate, hipStream_t stream);
int fftn_poisson_step(const FftNative& n, const float* flux, const float* exposure, const float2* khat, const float* background,
                      const float* counts, double* partials, int* n_partials, float eps, float inv_n, float* grad, float coef,
                      int accumulate, hipStream_t stream, double loss_scale, double loss_offset, float* loss_out);

constexpr int FFT_MAX_BATCH = 16;
struct FftBatch {  // per-dataset pointers of a batched likelihood step on the native FFT path (fftnative.hip); device memory
  int n;
  const float* exposure[FFT_MAX_BATCH];
  const float2* khat[FFT_MAX_BATCH];
  const float* background[FFT_MAX_BATCH];
  const float* counts[FFT_MAX_BATCH];
  float2* spec[FFT_MAX_BATCH];
  float2* work[FFT_MAX_BATCH];
  float* loss_out[FFT_MAX_BATCH];
  float loss_offset[FFT_MAX_BATCH];
  // calibrated steps (jd_npred_poisson_calibrated_batch_fwd_bwd): nullable entries
  const float* shift_xy[FFT_MAX_BATCH];
  const float* log_bkg_norm[FFT_MAX_BATCH];
  float* grad_shift_xy[FFT_MAX_BATCH];
  float* grad_log_bkg_norm[FFT_MAX_BATCH];
  float* gshift[FFT_MAX_BATCH];             // exposure x corr of the dataset: the transposed shift's input
};
int fftn_poisson_step_batch(const FftNative& n, int nd, const FftBatch* batch_dev, const float* flux, double* partials, float eps,
                            float inv_n, float* grad, float coef, int accumulate, hipStream_t stream, double loss_scale);
int fftn_poisson_step_pooled_batch(const FftNative& n, int upsampling, int nd, const FftBatch* batch_dev, const FftBatch& host,
                                   const float* flux, double* partials, double* partials_b, float eps, float inv_n, float* grad,
                                   double* partials_shift, float coef, int accumulate, hipStream_t stream, double loss_scale,
                                   double norm_grad_scale, int sequential);
bool fftn_pooled_supported(const FftNative& n, int upsampling);
int fftn_poisson_step_pooled(const FftNative& n, int upsampling, const float* flux, const float* exposure, const float2* khat,
                             const float* background, const float* counts, const float* log_bkg_norm, double* partials,
                             double* partials_b, float eps, float inv_n, float* target, float coef, int accumulate,
                             hipStream_t stream, double loss_scale, double loss_offset, float* loss_out, double norm_grad_scale,
                             float* norm_grad_out, const float* shift_xy, float shift_scale);

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
