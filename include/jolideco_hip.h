/*
 * jolideco_hip.h -- C-ABI of libjolideco_hip.so: the MI355X (gfx950) implementation of the
 * Jolideco MAP-deconvolution inner loop.
 *
 * The reference (pure Python/PyTorch, /root/reference) has no FFI for this path; its seam is the
 * Python object protocol.  Each entry point below names the reference code it replaces
 * (file:line relative to the reference root).  INTEGRATION.md shows the ctypes binding a
 * maintainer would add on the reference side.
 *
 * Conventions
 *   - every function returns 0 on success, a negative jd_status otherwise; the message of the
 *     last failure on the calling thread is available from jd_last_error()
 *   - all `float*` image arguments are DEVICE pointers to contiguous row-major fp32 (H, W)
 *     images borrowed from the caller (PyTorch owns them); handles own their workspaces
 *   - all work is enqueued asynchronously on the `stream` argument (a hipStream_t passed as
 *     void*); no call synchronises except *_create and *_destroy
 *   - scalar outputs (`*_out`) are DEVICE pointers to one float
 *   - no C++ exceptions cross this boundary; handles are not thread-safe, and a handle belongs to ONE stream at a
 *     time (its work buffers and cached device tables are reused from call to call without cross-stream events)
 */
#ifndef JOLIDECO_HIP_H
#define JOLIDECO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  JD_OK = 0,
  JD_ERR_INVALID = -1, /* bad argument / unsupported shape */
  JD_ERR_HIP = -2,     /* a HIP runtime call failed */
  JD_ERR_FFT = -3,     /* a rocFFT call failed */
  JD_ERR_ALLOC = -4
} jd_status;

#define JD_MAX_COMPONENTS 8

typedef struct jd_conv_plan jd_conv_plan; /* FFT-convolution geometry + rocFFT plans + workspaces */
typedef struct jd_gmm jd_gmm;             /* GMM constants in MFMA fragment order + workspaces */

/* library ------------------------------------------------------------------------------- */
int jd_version(void);
const char* jd_last_error(void);
/* number of HIP devices visible / name of the compiled target ("gfx950") */
const char* jd_target_arch(void);

/* Tuning / test switches (new; the reference has none).  Every switch has the name of the environment variable that
 * sets its initial value when the library is loaded (JD_GMM_SCREEN, JD_SEP_WALK, JD_SEP_NO_ALIAS, ...; the full list is
 * the table in csrc/options.hip); after loading, the environment is never read again by a launch path -- a test or an
 * A/B tool that wants another variant calls jd_set_option(key, "value"), or (key, NULL) to return to the default.
 * Unknown keys are an error.  jd_get_option: *is_set / *value of the switch as the library sees it now. */
int jd_set_option(const char* key, const char* value);
int jd_get_option(const char* key, int* is_set, int* value);

/* Convolution plan --------------------------------------------------------------------------
 * Replaces the per-call shape logic of jolideco/utils/torch.py:363-370 (`convolve_fft_torch`):
 * linear "same" convolution of an (H, W) image with a (kh, kw) kernel, zero outside the image; the
 * crop offset is the reference's `_centered` offset ((kh-1)/2, (kw-1)/2) (utils/torch.py:337-344).
 * Two methods compute the same function:
 *   FFT    where H is even, W % 4 == 0 and the padded lengths are of the form 2^a * {1, 3, 9} (<= 4608 columns, <= 2304 rows
 *          per image half): hand-written transforms on the un-padded grid (csrc/fftnative.hip: rows, columns with the
 *          k-space product, rows^-1 with the epilogue; the likelihood step of a dataset in five launches, with
 *          up-sampling and calibration too); otherwise rocFFT R2C / k-space multiply / C2R on a zero padded (Hp, Wp)
 *          grid, Hp >= H+kh-1, Wp >= W+kw-1 rounded up to FFT-friendly (2,3,5-smooth, Wp % 4 == 0) sizes
 *          (JD_CONV_MODE_FFT_EXACT forces rocFFT on the reference's own grid (H+kh-1, W+kw-1));
 *   DIRECT the sum over PSF taps as Toeplitz products on the matrix cores, PSFs up to 33x33; padding, exposure
 *          scaling and crop are folded into the kernel (csrc/directconv.hip).  Default where the operands fit in LDS
 *          (up to ~25x33): both operands split into two fp16 terms (22 significant bits, three fp16 MFMAs per
 *          product, fp32 accumulation; relative error of a convolution ~1e-6 of its maximum); otherwise, or with
 *          option JD_DIRECT_FP32=1, the fp32-input MFMA kernel (an exact fmaf chain).
 *   SEPARABLE for a PSF that is a sum of at most 3 outer products u_r v_r^T (a sampled Gaussian is 1, a
 *          double Gaussian 2): a row pass and a column pass of kh + kw taps per rank instead of kh * kw,
 *          PSFs up to 68x68 (csrc/sepconv.hip).  Whether a PSF qualifies is decided by
 *          jd_psf_separable_rank() / jd_conv_psf_spectrum(); a PSF that does not is an error for such a plan.
 * JD_CONV_MODE_AUTO picks DIRECT when the PSF is small enough for it to be faster than FFT -- up to 17 taps; up to 33
 * where the native FFT sizes do not fit, on images below 2^20 pixels, or with option JD_DIRECT_AUTO_ALL --; it never picks
 * SEPARABLE, because the choice depends on the PSF values, which a plan does not know: the caller asks
 * jd_psf_separable_rank() and requests JD_CONV_MODE_SEPARABLE (jolideco_amd.NPredModel does). */
enum {
  JD_CONV_MODE_AUTO = 0,
  JD_CONV_MODE_FFT_EXACT = 1,
  JD_CONV_MODE_FFT = 2,
  JD_CONV_MODE_DIRECT = 3,
  JD_CONV_MODE_SEPARABLE = 4
};
/* Rank R (1..3) of the outer-product decomposition of a (kh, kw) PSF given in HOST memory, or 0 when the PSF
 * is not that low-rank to within `tol` (sum |psf - sum_r u_r v_r^T| <= tol * sum |psf|; tol <= 0 selects the
 * default 3e-7, a few fp32 ulps) or larger than 68x68.  Host-only analysis, no device work. */
int jd_psf_separable_rank(const float* psf_host, int kh, int kw, float tol);
int jd_conv_plan_create(int H, int W, int kh, int kw, int mode, jd_conv_plan** plan_out);
int jd_conv_plan_destroy(jd_conv_plan* plan);
/* shape[0..5] = {H, W, Hp, Wp, oy, ox}; (Hp, Wp) = (H, W) for the direct method */
int jd_conv_plan_shape(const jd_conv_plan* plan, int* shape6);
/* 0 = FFT, 1 = DIRECT, 2 = SEPARABLE */
int jd_conv_plan_method(const jd_conv_plan* plan);
/* 1 when a plan of this geometry with the FFT method runs the hand-written transforms of csrc/fftnative.hip (and with them
 * the batched joint steps), 0 when it runs rocFFT (a size the native kernels do not cover, or option JD_FFT_NATIVE=0).
 * Host logic only, no device needed: a caller that is about to embed PSFs of different sizes in one array so that their
 * datasets share a plan (jolideco_amd/models/npred.py common_kernel_shape) asks first. */
int jd_conv_native_fft_supported(int H, int W, int kh, int kw);
/* 1 when a SEPARABLE plan runs a launch over `n_datasets` datasets (1 = the per-dataset calls) on the strip-walk kernels
 * of csrc/walkconv.hip instead of the tile kernel of csrc/sepconv.hip, given rank-1 operators (PSFs up to 17 x 17, image
 * width a multiple of 4, enough pixels x datasets to fill the chip; option JD_SEP_WALK = 0 / 1 forces either).  Both
 * compute the same function; a benchmark uses this to name the kernel it timed. */
int jd_conv_plan_takes_walk(const jd_conv_plan* plan, int n_datasets);
/* The frame -- 17 or 33 taps per direction -- in which the strip-walk kernels run the operator `khat` that
 * jd_conv_psf_spectrum built for this SEPARABLE plan, or 0 when they do not take it (rank > 1, non-zero taps wider than 33,
 * a buffer the library did not build).  The frame follows from the NON-ZERO taps of the operator, not from the plan's
 * (kh, kw): a 17 x 17 PSF embedded in a 33 x 33 array of zeros (datasets with different PSF sizes share one plan -- and one
 * batched joint step -- that way) walks in the 17-tap frame. */
int jd_conv_operator_walk_frame(const jd_conv_plan* plan, const float* khat);
/* The library remembers, by device address, what it knows about every SEPARABLE operator buffer it built (rank, support of
 * the taps).  A caller that frees such a buffer tells the library so: a later allocation at the same address is then an
 * unknown buffer again (and takes the general kernels) until jd_conv_psf_spectrum fills it. */
int jd_conv_operator_forget(const float* khat);
/* HALF the number of floats of one per-(dataset, component) kernel operator buffer `khat`:
 * FFT: complex64 elements of the kernel spectrum, Hp * (Wp/2 + 1); DIRECT: floats of one Toeplitz
 * fragment table (the buffer holds the forward and the adjoint table); SEPARABLE: the rank and the row /
 * column taps of both directions (a few hundred floats). */
size_t jd_conv_plan_spectrum_size(const jd_conv_plan* plan);

/* Kernel operator of one PSF, computed ONCE per (dataset, component) and cached by the caller in
 * `khat` (device, 2*spectrum_size floats).  FFT: K-hat = rfft2(psf, s=(Hp, Wp)) / (Hp*Wp); DIRECT: the
 * PSF in MFMA fragment order (forward and flipped).  Replaces the per-call
 * `torch.fft.rfft2(kernel, s=shape)` of utils/torch.py:368 (and the unused cache of
 * models/npred.py:117-127).  The 1/(Hp*Wp) of the unnormalised inverse transform is folded in. */
int jd_conv_psf_spectrum(jd_conv_plan* plan, const float* psf, float* khat, void* stream);

/* out[H,W] = crop( irfft2( rfft2(pad(image * scale_image)) * khat ) ); scale_image may be NULL.
 * Stand-alone `convolve_fft_torch` (utils/torch.py:347-370); used at setup for the exposure edge
 * correction of models/npred.py:108-113 and by NPredModel.forward (npred.py:175-179). */
int jd_conv_same(jd_conv_plan* plan, const float* image, const float* scale_image, const float* khat,
                 float* out, void* stream);

/* Fused forward model + Poisson NLL + gradient for ONE dataset and n_comp flux components.
 * Replaces, per optimizer step (jolideco/core.py:217-221,228):
 *   NPredModels.evaluate            models/npred.py:210-261  (sum_c clip(conv(flux_c*E_c, psf_c),0) + bkg)
 *   NPredModel.forward              models/npred.py:160-191  (upsampling_factor 1, no rmf)
 *   convolve_fft_torch              utils/torch.py:347-370
 *   nn.PoissonNLLLoss(log_input=False, reduction="mean", eps, full=True)   loss.py:35-37
 *   and the autograd backward of all of the above down to d loss / d flux_c.
 * loss_out      <- mean(n - c*log(n+eps)) + stirling_mean        (device scalar)
 * grad_flux[c]  <- (accumulate ? += : =) grad_scale * E_c * corr(psf_c, g * [conv_c >= 0]),
 *                  g = (1 - c/(n+eps)) / (H*W);  pass grad_flux == NULL for a forward-only
 *                  evaluation (PoissonLoss.evaluate, loss.py:56-71).
 * npred_out     optional output of the total predicted counts (counts grid).
 * upsampling    u >= 1: flux_c, exposure_c and grad_flux_c live on the plan's (H, W) grid, background,
 *               counts and npred_out on (H/u, W/u); the convolution is sum-pooled u x u before the clip
 *               (models/npred.py:181-184, `upsampling_factor`).  The PSF given to jd_conv_psf_spectrum and
 *               the exposure must already be up-sampled (models/npred.py:96-106 does that at setup).
 * `stirling_mean` = mean([c>1] * (c*log c - c + 0.5*log(2*pi*c))) is flux independent and is
 * supplied by the caller (computed once per dataset). */
int jd_npred_poisson_fwd_bwd(jd_conv_plan* plan, int n_comp, const float* const* flux,
                             const float* const* exposure, const float* const* khat,
                             const float* background, const float* counts, float stirling_mean,
                             float eps, float* loss_out, float* const* grad_flux, int accumulate,
                             float grad_scale, float* npred_out, int upsampling, void* stream);

/* The joint step over SEVERAL datasets in three launches instead of four per dataset (new: the reference has no joint
 * mode; this is the batched form of the loop `for dataset: jd_npred_poisson_fwd_bwd(..., accumulate = dataset > 0)` that
 * jolideco_amd's fit_mode="joint" runs, jolideco/core.py:214-229 being its per-dataset counterpart).  Restrictions: one
 * flux component, no up-sampling, no calibration, ONE plan shared by all datasets (same image and PSF array shape) with
 * the SEPARABLE method or with the FFT method on the native transforms (then every launch of the likelihood step covers
 * all datasets: csrc/fftnative.hip; a forward-only call and option JD_FFT_BATCH=0 run the per-dataset calls); at most 16
 * datasets per call.  Same results as the loop, bit for bit (the per-dataset gradient contributions are added in
 * dataset order).
 *   exposure, khat, background, counts, loss_out : host arrays of n_datasets device pointers
 *   stirling_mean                                : host array of n_datasets floats
 *   grad_flux                                    : nullable (forward only) */
int jd_npred_poisson_batch_fwd_bwd(jd_conv_plan* plan, int n_datasets, const float* flux, const float* const* exposure,
                                   const float* const* khat, const float* const* background,
                                   const float* const* counts, const float* stirling_mean, float eps,
                                   float* const* loss_out, float* grad_flux, int accumulate, float grad_scale,
                                   void* stream);

/* The batched joint step for SEVERAL flux components (NPredModels.evaluate sums the per-component models after their
 * clip, models/npred.py:210-261; per-component PSF / exposure as in npred.py:281-295): dataset d convolves flux[c] with
 * khat[d * n_components + c] after scaling by exposure[d * n_components + c]; one forward launch (grid.y = dataset, the
 * components walked inside the block), one loss launch and one adjoint launch per component.  Restrictions as above;
 * at most 4 components.  Same results as the per-dataset loop, bit for bit.
 *   flux, grad_flux     : host arrays of n_components device pointers (grad_flux nullable: forward only)
 *   exposure, khat      : host arrays of n_datasets * n_components device pointers, dataset major */
int jd_npred_poisson_batch_multi_fwd_bwd(jd_conv_plan* plan, int n_datasets, int n_components, const float* const* flux,
                                         const float* const* exposure, const float* const* khat,
                                         const float* const* background, const float* const* counts,
                                         const float* stirling_mean, float eps, float* const* loss_out,
                                         float* const* grad_flux, int accumulate, float grad_scale, void* stream);

/* The same with the per-dataset calibration of NPredCalibration (models/npred.py:298-402,225-237):
 *   shift_xy             device [2] = {shift_x, shift_y} in COUNTS pixels or NULL: every flux_c is shifted
 *                        (bilinear, zero padding = shift_image_torch, utils/torch.py:196-223) before the
 *                        exposure / PSF are applied; scale = upsampling
 *   log_background_norm  device [1] or NULL: background * exp(.)
 *   grad_shift_xy        device [2] or NULL  <- grad_scale * d loss / d shift_xy   (summed over components)
 *   grad_log_background_norm  device [1] or NULL  <- grad_scale * d loss / d log_background_norm
 * The parameters are read from device memory so that a fit never synchronises on them.  The caller
 * decides whether the shift is active (the reference skips it -- and its gradient -- while it is ~0). */
int jd_npred_poisson_calibrated_fwd_bwd(jd_conv_plan* plan, int n_comp, const float* const* flux,
                                        const float* const* exposure, const float* const* khat,
                                        const float* background, const float* counts, float stirling_mean,
                                        float eps, float* loss_out, float* const* grad_flux, int accumulate,
                                        float grad_scale, float* npred_out, int upsampling, const float* shift_xy,
                                        const float* log_background_norm, float* grad_shift_xy,
                                        float* grad_log_background_norm, void* stream);

/* The joint step over SEVERAL datasets of ONE flux component with the per-dataset calibration and / or up-sampling (the
 * fits of the reference's examples: examples/chandra-e0102-filament.py:178-203): the batched form of the loop
 * `for dataset: jd_npred_poisson_calibrated_fwd_bwd(..., accumulate = dataset > 0)` -- same results, bit for bit.  On a
 * plan of the native FFT convolution with up-sampling 2 or 4 every launch covers all datasets: rows (with the dataset's
 * shift), columns, the pooled Poisson launch, the adjoint's column pass, rows^-1 + adjoint epilogue into one image per
 * dataset, a transposed shift that adds the datasets up in dataset order, and the finalize of the shift gradients (at
 * every size since round 5; option JD_FFT_BATCH=0 runs the per-dataset calls, 3 / 4 the intermediate forms measured in
 * csrc/fftconv.hip).  Every other case runs the per-dataset calls.
 *   exposure, khat, background, counts, loss_out : host arrays of n_datasets device pointers
 *   shift_xy, log_background_norm, grad_shift_xy, grad_log_background_norm : NULL, or host arrays of n_datasets device
 *                                                  pointers with NULL entries where a dataset has none (see above) */
int jd_npred_poisson_calibrated_batch_fwd_bwd(jd_conv_plan* plan, int n_datasets, const float* flux,
                                              const float* const* exposure, const float* const* khat,
                                              const float* const* background, const float* const* counts,
                                              const float* stirling_mean, float eps, float* const* loss_out,
                                              float* grad_flux, int accumulate, float grad_scale, int upsampling,
                                              const float* const* shift_xy, const float* const* log_background_norm,
                                              float* const* grad_shift_xy, float* const* grad_log_background_norm,
                                              void* stream);

/* Same chain split at the reference's object seams, for callers that keep the reference's own
 * loop structure (autograd.Function wrappers in jolideco_amd/ops.py):
 * conv_padded[c] are plan-owned buffers exposed through jd_conv_plan_conv_buffer. */
int jd_poisson_nll(const float* npred, const float* counts, size_t n, float stirling_mean, float eps,
                   float* loss_out, float* grad_npred /* nullable; (1 - c/(n+eps))/n_total */,
                   void* stream);
/* grad_image (+)= scale_image * crop_adjoint( irfft2( rfft2(pad(grad_out)) * conj(khat) ) ):
 * the adjoint of jd_conv_same (autograd of utils/torch.py:367-370). */
int jd_conv_same_adjoint(jd_conv_plan* plan, const float* grad_out, const float* scale_image,
                         const float* khat, float* grad_image, int accumulate, void* stream);

/* GMM patch prior ------------------------------------------------------------------------
 * jd_gmm_create takes HOST pointers to the fp32 constants of
 * jolideco/priors/patches/gmm.py: precisions_cholesky (K, D, D) (:139-149, utils/numpy.py:16-34),
 * means_precisions_cholesky (K, D) (:217-228), const_k[k] = -0.5*D*log(2*pi) + log_det_cholesky[k]
 * + log_weights[k] (:235-240,114-117,276-281) and pixel_weights (D,) (:283-299).  D must be 64
 * (8x8 patches), K <= 4096.  The library re-lays them out in MFMA fragment order with
 * sqrt(pixel_weight) folded into the columns. */
int jd_gmm_create(int K, int D, const float* prec_chol, const float* mu_prec, const float* const_k,
                  const float* pixel_w, jd_gmm** gmm_out);
int jd_gmm_destroy(jd_gmm* gmm);
/* 1 when every precisions_cholesky[k] is upper triangular (what compute_precision_cholesky,
 * utils/numpy.py:16-34, produces): the kernels then skip the all-zero 16x16 blocks (40 instead of 64
 * MFMAs per component and 16 patches, bit-identical results); 0 = dense variant. */
int jd_gmm_is_triangular(const jd_gmm* gmm);

/* log-prior value and gradient of GMMPatchPrior.__call__ (priors/patches/core.py:189-246) with
 * IdentityImageNorm, SubtractMeanPatchNorm (utils/norms.py:97-103), cycle-spin roll by
 * (shift_y, shift_x) (utils/torch.py:108-119), patch size 8, stride `stride`
 * (utils/torch.py:226-275), GaussianMixtureModel.estimate_log_prob (patches/gmm.py:262-281),
 * max over components (marginalize = 0) or logsumexp (marginalize = 1).  With a gradient and upper triangular
 * factors both modes go through the fp16 screen: the arg-max result is that of the dense fp32 kernel bit for bit; the
 * logsumexp leaves out terms below exp(-25) of the largest one (< 2e-9 of the sum) and is held to the reference's
 * values at 5e-5 like the dense logsumexp kernel (options JD_GMM_SCREEN=0 / JD_GMM_LSE_SCREEN=0: dense kernels).
 * Reproducibility: the arg-max mode is bit-reproducible from run to run.  The logsumexp mode with a gradient is
 * reproducible with JD_GMM_LSE_SCREEN=0 (always the dense kernels) or =2 (always the screen); by DEFAULT the library
 * decides per pass, from host-mapped statistics of earlier passes that it reads without synchronising, whether the
 * screen is worth running -- on images where it keeps overflowing, WHICH pass first skips it depends on host / GPU
 * timing, and the two paths differ at the rounding level (both within 5e-5 of the reference).
 *   value_out       <- (accumulate_value ? += : =) value_scale * sum_{patches in shard} v_patch
 *   grad_flux_accum += grad_coef * d(sum_{patches in shard} v_patch)/d flux   (NULL: forward only)
 *   argmax_out      optional int32 per patch (global patch index order), max mode only
 * Only patch rows [patch_row_begin, patch_row_end) are evaluated (multi-GPU shard of the prior;
 * pass 0 and -1 for all rows).
 * DEVICE-RESIDENT STEP SCALARS (new: what lets a whole epoch be captured in a hipGraph and replayed): with
 * shift_dev != NULL the kernels read the roll from device memory -- shift_dev[0] = shift_y in [0, H), shift_dev[1] =
 * shift_x in [0, W), the residues the host would have passed -- and ignore the two by-value arguments; every launch
 * argument of the pass is then the same from step to step (the caller uploads the shifts of the coming steps with one
 * small copy, jolideco_amd/core.py StepScalars).  Same results as the by-value form, bit for bit.
 * TWO PHASES (new: the prior beside the likelihood): `phases` = 3 runs the whole pass; 1 runs everything up to the
 * per-patch gradient rows -- value, arg-max, rows: it reads the flux and writes value_out and the handle's work buffers
 * only --, 2 the gather of those rows into grad_flux_accum (for _step: + the optimizer step).  A caller enqueues phase 1
 * on a second stream next to the likelihood launches of the step, joins the streams and calls phase 2 with the same
 * arguments (JD_ERR_INVALID if they differ or another pass of the handle ran in between); a handle has one pass in
 * flight at a time. */
int jd_gmm_prior_fwd_bwd(jd_gmm* gmm, const float* flux, int H, int W, int stride, int shift_y,
                         int shift_x, int patch_row_begin, int patch_row_end, int marginalize,
                         float value_scale, float* value_out, int accumulate_value, float grad_coef,
                         float* grad_flux_accum, int32_t* argmax_out, const int* shift_dev, int phases, void* stream);

/* The same evaluation with the OPTIMIZER STEP of the component in the epilogue of its last kernel (new: one pass over
 * the gradient image and one launch less per step; jolideco/core.py:229 is the step it folds in).  Where
 * jd_gmm_prior_fwd_bwd adds grad_coef * d(sum v_patch)/d flux into the gradient image, this call forms
 *   g = step->grad_flux[pixel] + grad_coef * d(sum v_patch)/d flux[pixel]      (grad_flux: all other terms, read only)
 * and applies jd_adam_step's (or jd_sgd_step's) update to every pixel of the image: theta, exp_avg, exp_avg_sq in
 * place, flux_out = exp(theta_new) [* mask].  Same bits as the two separate calls.  Whole prior only (no shard);
 * stride >= 4; JD_ERR_INVALID otherwise (the caller then makes the two calls). */
typedef struct {
  float* theta;
  const float* flux_in;
  float* flux_out;
  const float* grad_flux; /* d loss / d flux of everything but this prior */
  float* exp_avg;         /* NULL with sgd */
  float* exp_avg_sq;
  const float* mask;      /* nullable */
  float step_size, beta1, beta2, one_minus_beta1, one_minus_beta2, bias2_sqrt, eps; /* as jd_adam_step */
  float lr;               /* sgd */
  int use_log_flux, sgd;
  const float* bias_dev;  /* nullable, device [2] = {step_size, bias2_sqrt}: read by the kernel instead of the two members
                             above (the step count of a captured graph's optimizer step lives in device memory) */
} jd_step;
int jd_gmm_prior_fwd_bwd_step(jd_gmm* gmm, const float* flux, int H, int W, int stride, int shift_y, int shift_x,
                              int marginalize, float value_scale, float* value_out, int accumulate_value,
                              float grad_coef, const jd_step* step, const int* shift_dev, int phases, void* stream);

/* Diagnostics of the screened arg-max path, read without synchronisation from host-mapped memory the last block of a
 * pass writes: out[0..4] = {generation of the last finished pass, it fell back to the dense fp32 kernel (0 / 1), bucket
 * slots its surviving records used, patches it covered, gradient rows per patch the record buffer has room for now}.
 * The library uses the same numbers to double that room (4 -> 32) after a pass that ran out of it. */
int jd_gmm_screen_stats(const jd_gmm* gmm, int* out);
/* The shader clock the board holds INSIDE the screen kernel (a measurement tool of bench.py: the matrix roof the kernel
 * is priced against scales with it).  The first call arms the handle: its default screen launch then runs an instantiation
 * whose blocks leave s_memtime / s_memrealtime tick counts between their first and last instruction.  Later calls
 * synchronise the device, return the mean of 100 MHz x (shader ticks / reference ticks) over the blocks stamped since the
 * previous call (samples_out of them, at most 4096 per launch) and clear the stamps. */
int jd_gmm_screen_clock(jd_gmm* gmm, double* mhz_out, int* samples_out);

/* The same evaluation for ONE RANK OF A SHARDED PRIOR (joint fit over several GPUs; the
 * reference has no distributed code): instead of accumulating into the gradient image, the gradient of the shard's
 * patch rows is written as a compact band of the ROLLED frame,
 *   band_out[(Y - y_begin) * W + X] = grad_coef * d(sum_{patches in shard} v_patch)/d flux_rolled[Y, X],
 *   Y in [y_begin, y_end) = [patch_row_begin * stride, (patch_row_end - 1) * stride + 8)   (0 where no patch covers),
 * which the ranks exchange with ONE all-gather while the all-reduce of the likelihood gradient is in flight. */
int jd_gmm_prior_band_fwd_bwd(jd_gmm* gmm, const float* flux, int H, int W, int stride, int shift_y,
                              int shift_x, int patch_row_begin, int patch_row_end, int marginalize,
                              float value_scale, float* value_out, int accumulate_value, float grad_coef,
                              float* band_out, void* stream);
/* grad[(Y - shift_y) mod H, (X - shift_x) mod W] += sum_b bands[b * chunk_floats + (Y - y_begin[b]) * W + X] over the
 * bands with y_begin[b] <= Y < y_end[b], added in band order (identical on every rank).  `bands` is the device buffer an
 * all-gather of the per-rank bands produced; y_begin / y_end are HOST arrays of n_bands (<= 64) entries. */
int jd_add_rolled_bands(float* grad, int H, int W, int shift_y, int shift_x, const float* bands, size_t chunk_floats,
                        int n_bands, const int* y_begin, const int* y_end, void* stream);
/* The same sum followed at once by the optimizer step of jd_adam_step (sharded fits, where the prior's bands are the
 * last term of the gradient): every pixel takes g = step->grad_flux[pixel] + (the band sum above, same additions in the
 * same order) -- the gradient image itself is only read -- and the update of jd_adam_step with g: the bits of
 * jd_add_rolled_bands followed by jd_adam_step, one launch and one pass over the gradient image less.  Needs W % 4 == 0
 * and 16-byte aligned images (JD_ERR_INVALID otherwise: use the two calls). */
int jd_add_rolled_bands_step(int H, int W, int shift_y, int shift_x, const float* bands, size_t chunk_floats, int n_bands,
                             const int* y_begin, const int* y_end, const jd_step* step, void* stream);

/* (Np, K) log-probabilities of explicit patches: GaussianMixtureModel.estimate_log_prob
 * (patches/gmm.py:262-281).  x: (n, 64) device, out: (n, K) device. */
int jd_gmm_estimate_log_prob(jd_gmm* gmm, const float* x, int n, float* out, void* stream);

/* Element-wise priors: InverseGammaPrior (priors/core.py:207-226; kind 1: alpha, beta) and
 * ExponentialPrior (priors/core.py:308-326; kind 2: alpha).  value_out <- sum(v)/n + log_const;
 * grad_flux_accum += grad_coef * d value / d flux. */
int jd_elementwise_prior_fwd_bwd(int kind, const float* flux, size_t n, float alpha, float beta,
                                 float log_const, float* value_out, float grad_coef,
                                 float* grad_flux_accum, void* stream);

/* Parameter update -------------------------------------------------------------------------
 * flux = exp(theta) [* mask]   (SpatialFluxComponent.flux_upsampled, models/core.py:583-594);
 * use_log_flux == 0: flux = theta [* mask] (the parameter is the flux itself, no positivity). */
int jd_flux_from_theta(const float* theta, const float* mask, float* flux, size_t n, int use_log_flux, void* stream);

/* Flux components that SHARE one forward operator.  NPredModels.from_dataset_numpy (models/npred.py:279-295) builds the
 * model of every component of a dataset from the same exposure and -- unless `psf` is a dict -- the same PSF, and
 * NPredModels.evaluate (models/npred.py:241-261) adds clip(PSF * (flux_c x exposure), 0) over the components: with a
 * non-negative PSF, exposure and fluxes no term is ever clipped, so by linearity ONE convolution of (sum_c flux_c) x exposure
 * gives npred, and d loss / d flux_c is the same image for every c.  The host side (jolideco_amd/loss.py) then runs the
 * single-component launches between these two helpers:
 *   jd_sum_images:    out[i] = ((srcs[0][i] + srcs[1][i]) + ...)    n_srcs in [1, 4], host array of device pointers
 *   jd_copy_image_to: dsts[d][i] = src[i]                           n_dsts in [1, 4] */
int jd_sum_images(float* out, const float* const* srcs, int n_srcs, size_t n, void* stream);
int jd_copy_image_to(const float* src, float* const* dsts, int n_dsts, size_t n, void* stream);

/* Device-resident step scalars (see jd_gmm_prior_fwd_bwd): dst[0 .. n) <- host_row[0 .. n), read by ONE small block
 * straight from PINNED host memory (hipHostMalloc'ed, e.g. a torch tensor after pin_memory(): device-accessible) -- the
 * epoch's shifts and bias terms reach the device without a copy-engine hand-over on the stream.  The caller keeps the row
 * untouched until the launch has run (jolideco_amd/core.py StepScalars: a ring of rows guarded by events). */
int jd_step_scalars_fetch(const int32_t* host_row, int32_t* dst, int n, void* stream);

/* One torch.optim.Adam step (jolideco/core.py:39-42,229) on theta with the chain rule of
 * models/core.py:588-592 fused in: g_theta = grad_flux * flux_in.  Writes theta, exp_avg,
 * exp_avg_sq in place, writes flux_out = exp(theta_new) [* mask] (may alias flux_in or be a second
 * buffer so the caller can keep the pre-step flux for the loss trace, core.py:247) and zeroes
 * grad_flux when zero_grad != 0.  step_size = lr / (1 - beta1^t), bias2_sqrt = sqrt(1 - beta2^t),
 * one_minus_beta1 = 1 - beta1 and one_minus_beta2 = 1 - beta2 are computed by the caller in double
 * precision and rounded once, as torch does (fp32(1 - 0.999) != 1 - fp32(0.999)).
 * bias_dev != NULL: device [2] = {step_size, bias2_sqrt}, read by the kernel instead of the two by-value arguments (see
 * jd_gmm_prior_fwd_bwd: device-resident step scalars). */
int jd_adam_step(float* theta, const float* flux_in, float* flux_out, float* grad_flux, float* exp_avg,
                 float* exp_avg_sq, const float* mask, size_t n, float step_size, float beta1,
                 float beta2, float one_minus_beta1, float one_minus_beta2, float bias2_sqrt, float eps,
                 int zero_grad, int use_log_flux, const float* bias_dev, void* stream);
/* jd_adam_step with use_log_flux = 0 for MANY small parameter vectors in ONE launch (the calibration parameters of the
 * datasets of a joint step, jolideco/core.py:197-204,229): tensor i (sizes[i] floats) takes its own step_size[i] /
 * bias2_sqrt[i] (its own step count, as torch.optim.Adam keeps one per parameter).  Host arrays of n_tensors <= 64
 * entries.  Same bits as n_tensors calls of jd_adam_step.  bias_dev != NULL: device [2 n_tensors], {step_size,
 * bias2_sqrt} of tensor i at [2 i], read by the kernel instead of the two host arrays (which may then be NULL). */
int jd_adam_step_multi(int n_tensors, float* const* theta, const float* const* grad, float* const* exp_avg,
                       float* const* exp_avg_sq, const int* sizes, const float* step_size, const float* bias2_sqrt,
                       float beta1, float beta2, float one_minus_beta1, float one_minus_beta2, float eps,
                       const float* bias_dev, void* stream);
/* plain SGD (core.py:41): theta -= lr * grad_flux * flux_in */
int jd_sgd_step(float* theta, const float* flux_in, float* flux_out, float* grad_flux,
                const float* mask, size_t n, float lr, int zero_grad, int use_log_flux, void* stream);

/* Kernel timers -----------------------------------------------------------------------------
 * New (the reference has no profiler hooks).  After jd_profile_enable(n) the
 * library brackets every launch of the kernels below with a hipEvent pair on the launch stream
 * (at most n pairs; further launches are not timed); jd_profile_read synchronises on the recorded
 * events and returns the summed duration and the number of timed launches of one kernel.
 * Calling jd_profile_enable again resets the counters. */
enum {
  JD_KERNEL_POISSON_FUSED = 0,   /* K3: clip + background + Poisson NLL + gradient */
  JD_KERNEL_GMM_FWD = 1,         /* K4: GMM patch log-likelihood, max / logsumexp over components (whole forward pass) */
  JD_KERNEL_GMM_BWD = 2,         /* K4b: per-patch gradient for the selected component(s) */
  JD_KERNEL_GMM_GATHER = 3,      /* K4c: deterministic overlap-add of the patch gradients */
  JD_KERNEL_PAD_MUL = 4,         /* K1 */
  JD_KERNEL_CMUL = 5,            /* K2 */
  JD_KERNEL_ADJOINT_EPILOGUE = 6,/* K5 */
  JD_KERNEL_ADAM = 7,            /* K6 */
  JD_KERNEL_FFT_R2C = 8,         /* rocFFT real forward transform (all its kernels) */
  JD_KERNEL_FFT_C2R = 9,         /* rocFFT real inverse transform (all its kernels) */
  JD_KERNEL_DIRECT_CONV = 10,    /* MFMA Toeplitz convolution / correlation (small PSFs) */
  JD_KERNEL_SEP_CONV = 11,       /* separable (low-rank PSF) convolution / correlation */
  JD_KERNEL_GMM_SCREEN = 12,     /* K4 screened arg-max, stage 1: fp16 MFMA screen with error bounds (inside GMM_FWD) */
  JD_KERNEL_GMM_SORT = 13,       /*   stage 2: counting sort of the surviving (patch, component) records (inside GMM_FWD) */
  JD_KERNEL_GMM_EXACT = 14,      /*   stage 3: exact fp32 MFMA evaluation of the survivors (inside GMM_FWD) */
  JD_KERNEL_GMM_STAGE = 15,      /*   stage 0: patches -> mean-subtracted fp16 fragments, norms, scales (inside GMM_FWD) */
  JD_KERNEL_SHIFT = 16,          /* calibration: bilinear sub-pixel shift and its transpose (+ shift gradient partial sums) */
  JD_KERNEL_COUNT = 17
};
int jd_profile_enable(int capacity);
int jd_profile_disable(void);
/* pause (1) / resume (0) the timers without resetting them: lets a caller time a sample of its steps */
int jd_profile_pause(int paused);
int jd_profile_read(int kernel, double* total_ms, long long* launches);
const char* jd_kernel_name(int kernel);
/* Sustained shader clock of the current device under a vector-ALU load on every CU, in MHz (new; a measurement aid:
 * MI355X boards hold different clocks under load, so a benchmark line should say which clock its times were taken at).
 * Runs a dependent-FMA kernel for about `milliseconds` on `stream`, stamped with s_memtime / s_memrealtime, and
 * SYNCHRONISES the stream. */
int jd_clock_probe(double milliseconds, double* mhz_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* JOLIDECO_HIP_H */
