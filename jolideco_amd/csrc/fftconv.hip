// FFT "same" convolution on rocFFT (R2C / C2R, single precision, out-of-place) and the fused
// forward-model + Poisson step built on it.  One plan per image/PSF geometry; kernel spectra are
// computed once per (dataset, component) and cached by the caller.
#include <rocfft/rocfft.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <mutex>

#include "jd_common.h"
#include "kernels.h"

#define JD_FFT(call)                                                                               \
  do {                                                                                             \
    rocfft_status s_ = (call);                                                                     \
    if (s_ != rocfft_status_success)                                                               \
      return jd::fail(JD_ERR_FFT, "%s failed with rocfft_status %d (%s:%d)", #call, (int)s_,       \
                      __FILE__, __LINE__);                                                         \
  } while (0)

struct jd_conv_plan {
  int H = 0, W = 0, kh = 0, kw = 0, Hp = 0, Wp = 0, oy = 0, ox = 0;
  // JD_CONV_FFT: rocFFT on the padded (Hp, Wp) grid | JD_CONV_DIRECT: MFMA Toeplitz | JD_CONV_SEPARABLE: row + column pass
  int method = jd::JD_CONV_FFT;
  int py = 0, px = 0;        // offset of the (H, W) image inside the conv / pad buffers (FFT: oy, ox; direct: 0)
  size_t nspec = 0;  // complex elements of one spectrum (direct: floats of one Toeplitz fragment table)
  int split = 0;     // direct: the split-fp16 kernel (default where it fits; JD_DIRECT_FP32=1: the fp32 MFMA kernel)
  // FFT method on sizes the hand-written transforms cover (fftnative.hip): no rocFFT plans, no padded buffers
  bool native = false;
  jd::FftNative fftn;
  rocfft_plan fwd = nullptr, inv = nullptr;
  rocfft_execution_info info = nullptr;
  void* work = nullptr;
  size_t work_bytes = 0;
  float* pad[JD_MAX_COMPONENTS] = {nullptr};   // R2C inputs (padded u = flux*E, later padded g)
  float* conv[JD_MAX_COMPONENTS] = {nullptr};  // C2R outputs (padded convolution / correlation)
  float2* spec = nullptr;
  double* partials = nullptr;
  int partials_cap = 0;
  // calibration work space (allocated on first use): shifted flux / gradient w.r.t. the shifted flux per
  // component on the (H, W) flux grid, partial sums of the background-norm and shift gradients
  float* shifted[JD_MAX_COMPONENTS] = {nullptr};
  float* gshift[JD_MAX_COMPONENTS] = {nullptr};
  double* partials_cal = nullptr;
  // batched joint step (jd_npred_poisson_batch_fwd_bwd): one g work image per dataset, partial sums per dataset
  float* gbatch[jd::SEP_MAX_BATCH * jd::SEP_BATCH_MAX_COMP] = {nullptr};
  double* partials_batch = nullptr;
  int partials_batch_cap = 0;
  // batched joint step on the native FFT path: the work arrays of datasets 1 .. (dataset 0 uses fftn's own)
  float2* fft_extra_spec[jd::FFT_MAX_BATCH - 1] = {nullptr};
  float2* fft_extra_work[jd::FFT_MAX_BATCH - 1] = {nullptr};
  // the tables of the batched steps in device memory: a few slots keyed by content (the plain and the calibrated step, the
  // chunks of a fit with more than FFT_MAX_BATCH datasets and several sessions on one plan alternate between tables; an
  // upload has to wait for the stream)
  static constexpr int N_FFT_TABLES = 6;
  jd::FftBatch fft_batch_host[N_FFT_TABLES] = {};
  jd::FftBatch* fft_batch_dev[N_FFT_TABLES] = {};
  unsigned long long fft_batch_used[N_FFT_TABLES] = {};  // last use (call counter), 0 = empty
  unsigned long long fft_batch_clock = 0;
  double* partials_shift_batch = nullptr;  // calibrated batched step: 2 x shift blocks doubles per dataset
  float* gshift_batch[jd::FFT_MAX_BATCH] = {nullptr};  // and one (H, W) image per dataset: exposure x corr
  int partials_shift_batch_cap = 0;
  // pointer tables of the batched joint step in device memory: a few slots keyed by content, so that sessions (or the
  // chunks of a fit with more than SEP_MAX_BATCH datasets) that alternate between tables never re-upload -- an upload
  // has to wait for the stream
  static constexpr int N_TABLES = 8;
  jd::SepBatchTable table_host[N_TABLES] = {};
  jd::SepBatchTable* table_dev[N_TABLES] = {};
  unsigned long long table_used[N_TABLES] = {};  // last use (call counter), 0 = empty
  unsigned long long table_clock = 0;
};

namespace jd {

// 2,3,5-smooth only: measured on MI355X / rocFFT (ROCm 7.2) for a 2048^2 image + 17x17 PSF, one
// convolution (R2C + multiply + C2R + pad / crop): 2064 (the exact grid) 423 us, 2100 (has a factor 7) 180 us,
// 2160 138 us, 2304 134 us, 2400 145 us; 1024^2 + 129x129: 1152 69 us, 1176 (7^2) 74 us, 1216 (19) 164 us.
static bool is_smooth(int n) {
  for (int p : {2, 3, 5})
    while (n % p == 0) n /= p;
  return n == 1;
}

// smallest 2,3,5-smooth integer >= n that is a multiple of `mult`
static int next_fast_len(int n, int mult) {
  int m = ((n + mult - 1) / mult) * mult;
  while (!is_smooth(m)) m += mult;
  return m;
}

static int ensure_component_buffers(jd_conv_plan* p, int n_comp) {
  const size_t bytes = (size_t)p->Hp * p->Wp * sizeof(float);
  for (int c = 0; c < n_comp; ++c) {
    if (!p->pad[c]) JD_HIP(hipMalloc(&p->pad[c], bytes));
    if (!p->conv[c]) JD_HIP(hipMalloc(&p->conv[c], bytes));
  }
  return JD_OK;
}

static int ensure_calibration_buffers(jd_conv_plan* p, int n_comp, bool shift) {
  if (!p->partials_cal) {
    const size_t n = (size_t)(p->partials_cap > 2 * shift_bwd_max_blocks(p->H, p->W) ? p->partials_cap
                                                                                    : 2 * shift_bwd_max_blocks(p->H, p->W));
    JD_HIP(hipMalloc(&p->partials_cal, n * sizeof(double)));
  }
  if (shift) {
    const size_t bytes = (size_t)p->H * p->W * sizeof(float);
    for (int c = 0; c < n_comp; ++c) {
      if (!p->shifted[c]) JD_HIP(hipMalloc(&p->shifted[c], bytes));
      if (!p->gshift[c]) JD_HIP(hipMalloc(&p->gshift[c], bytes));
    }
  }
  return JD_OK;
}

static int exec_fft(jd_conv_plan* p, rocfft_plan plan, void* in, void* out, hipStream_t stream) {
  JD_FFT(rocfft_execution_info_set_stream(p->info, stream));
  void* in_buf[1] = {in};
  void* out_buf[1] = {out};
  ProfScope prof(plan == p->fwd ? JD_KERNEL_FFT_R2C : JD_KERNEL_FFT_C2R, stream);
  JD_FFT(rocfft_execute(plan, in_buf, out_buf, p->info));
  return JD_OK;
}

// conv[c] <- irfft2( rfft2(pad(image*scale)) * khat )   (padded layout, crop on read)
static int conv_forward(jd_conv_plan* p, int c, const float* image, const float* scale, const float* khat,
                        hipStream_t stream) {
  if (p->method == JD_CONV_DIRECT)
    return launch_direct_conv(image, scale, khat, p->conv[c], nullptr, p->H, p->W, p->kh, p->kw, p->oy, p->ox, 0,
                              1.f, 0, p->split, stream);
  if (p->method == JD_CONV_SEPARABLE)
    return launch_sep_conv(image, scale, khat, p->conv[c], nullptr, p->H, p->W, p->kh, p->kw, p->oy, p->ox, 0, 1.f, 0,
                           stream);
  if (p->native)
    return fftn_conv(p->fftn, image, scale, reinterpret_cast<const float2*>(khat), p->conv[c], nullptr, 0, 1.f, 0, stream);
  int rc = launch_pad_mul(image, scale, p->pad[c], p->H, p->W, p->Hp, p->Wp, stream);
  if (rc) return rc;
  if ((rc = exec_fft(p, p->fwd, p->pad[c], p->spec, stream))) return rc;
  if ((rc = launch_cmul(p->spec, reinterpret_cast<const float2*>(khat), p->nspec, false, stream))) return rc;
  return exec_fft(p, p->inv, p->spec, p->conv[c], stream);
}

// grad (+)= coef * scale * crop_adjoint( corr(pad[c], psf) ); pad[c] already holds the (padded) gradient g_c.
// FFT: conv[c] <- irfft2( rfft2(pad[c]) * conj(khat) ), then the K5 epilogue; direct: one fused kernel.
static int corr_backward_into(jd_conv_plan* p, int c, const float* khat, const float* scale, float* grad, float coef,
                              int accumulate, hipStream_t stream, const SepLossFold* fold = nullptr, int* fold_done = nullptr);

static int corr_backward(jd_conv_plan* p, int c, const float* khat, hipStream_t stream) {
  int rc = exec_fft(p, p->fwd, p->pad[c], p->spec, stream);
  if (rc) return rc;
  if ((rc = launch_cmul(p->spec, reinterpret_cast<const float2*>(khat), p->nspec, true, stream))) return rc;
  return exec_fft(p, p->inv, p->spec, p->conv[c], stream);
}

static int corr_backward_into(jd_conv_plan* p, int c, const float* khat, const float* scale, float* grad, float coef,
                              int accumulate, hipStream_t stream, const SepLossFold* fold, int* fold_done) {
  if (fold_done) *fold_done = 0;
  if (p->method == JD_CONV_DIRECT)
    return launch_direct_conv(p->pad[c], nullptr, khat + p->nspec, grad, scale, p->H, p->W, p->kh, p->kw, p->oy, p->ox,
                              1, coef, accumulate, p->split, stream);
  if (p->method == JD_CONV_SEPARABLE)
    return launch_sep_conv(p->pad[c], nullptr, khat, grad, scale, p->H, p->W, p->kh, p->kw, p->oy, p->ox, 1, coef,
                           accumulate, stream, fold, fold_done);
  if (p->native)
    return fftn_conv(p->fftn, p->pad[c], nullptr, reinterpret_cast<const float2*>(khat), grad, scale, 1, coef, accumulate, stream);
  int rc = corr_backward(p, c, khat, stream);
  if (rc) return rc;
  return launch_adjoint_epilogue(p->conv[c], scale, grad, p->H, p->W, p->Hp, p->Wp, p->oy, p->ox, coef, accumulate,
                                 stream);
}

__global__ __launch_bounds__(256) void scale_copy_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                        size_t n, float s) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) out[i] = in[i] * s;
}

// The device copy of a batched step's table: looked up by content (`batch` must come from a value-initialised FftBatch{},
// so that its padding bytes are zero), uploaded into the least recently used slot when it is new -- after waiting for the
// launches of `s` that may still read that slot (a table is only ever read by launches of the stream that uploaded it:
// handles are not shared between streams, include/jolideco_hip.h).
static int fft_batch_table(jd_conv_plan* p, const FftBatch& batch, hipStream_t s, int* slot_out) {
  int slot = -1, victim = 0;
  for (int i = 0; i < jd_conv_plan::N_FFT_TABLES; ++i) {
    if (p->fft_batch_used[i] && memcmp(&batch, &p->fft_batch_host[i], sizeof(batch)) == 0) slot = i;
    if (p->fft_batch_used[i] < p->fft_batch_used[victim]) victim = i;
  }
  if (slot < 0) {
    slot = victim;
    if (!p->fft_batch_dev[slot]) JD_HIP(hipMalloc(&p->fft_batch_dev[slot], sizeof(FftBatch)));
    if (p->fft_batch_used[slot]) JD_HIP(hipStreamSynchronize(s));
    JD_HIP(hipMemcpy(p->fft_batch_dev[slot], &batch, sizeof(batch), hipMemcpyHostToDevice));
    p->fft_batch_host[slot] = batch;
  }
  p->fft_batch_used[slot] = ++p->fft_batch_clock;
  *slot_out = slot;
  return JD_OK;
}

}  // namespace jd

using namespace jd;

extern "C" int jd_conv_plan_create(int H, int W, int kh, int kw, int mode, jd_conv_plan** plan_out) {
  JD_REQUIRE(plan_out, "jd_conv_plan_create: plan_out is null");
  JD_REQUIRE(H > 0 && W > 0 && kh > 0 && kw > 0, "jd_conv_plan_create: non-positive shape (%d,%d,%d,%d)", H,
             W, kh, kw);
  JD_REQUIRE((long)H * W < (1L << 31), "jd_conv_plan_create: image too large");
  JD_REQUIRE(mode >= JD_CONV_MODE_AUTO && mode <= JD_CONV_MODE_SEPARABLE, "jd_conv_plan_create: unknown mode %d", mode);
  JD_REQUIRE(mode != JD_CONV_MODE_SEPARABLE || sep_conv_supported(kh, kw),
             "jd_conv_plan_create: the separable method supports PSFs up to %dx%d, got %dx%d", SEP_MAX_K, SEP_MAX_K, kh, kw);
  JD_REQUIRE(mode != JD_CONV_MODE_DIRECT || direct_conv_supported(kh, kw),
             "jd_conv_plan_create: the direct method supports PSFs up to 33x33, got %dx%d", kh, kw);
  const bool exact_shape = mode == JD_CONV_MODE_FFT_EXACT;
  // "auto" for a general PSF: the MFMA Toeplitz kernel up to 17 taps; beyond, its cost grows with the PSF area while the
  // native FFT convolution's does not (2048^2, forward convolution: 21 taps 77 against 49 us, 33 taps 136 against 48 us;
  // 1024^2: 24 against 23 and 38 against 23 us; 17 taps: 27 against 49 us -- tools/conv_bench.py), so images of a
  // megapixel and more take the native FFT path where its sizes allow
  const bool native_ok = !exact_shape && opt_value(OPT_FFT_NATIVE, 1) != 0 && fftn_supported(H, W, kh, kw);
  const bool auto_direct = direct_conv_supported(kh, kw) &&
                           (std::max(kh, kw) <= 17 || !native_ok || (long)H * W < (1L << 20) || opt_is_set(OPT_DIRECT_AUTO_ALL));
  if (mode == JD_CONV_MODE_SEPARABLE || mode == JD_CONV_MODE_DIRECT || (mode == JD_CONV_MODE_AUTO && auto_direct)) {
    // both work on the unpadded (H, W) grid: no FFT plans, only the image-sized work buffers
    jd_conv_plan* p = new (std::nothrow) jd_conv_plan();
    if (!p) return fail(JD_ERR_ALLOC, "jd_conv_plan_create: out of host memory");
    p->method = mode == JD_CONV_MODE_SEPARABLE ? JD_CONV_SEPARABLE : JD_CONV_DIRECT;
    p->H = H, p->W = W, p->kh = kh, p->kw = kw, p->Hp = H, p->Wp = W;
    p->oy = (kh - 1) / 2, p->ox = (kw - 1) / 2, p->py = 0, p->px = 0;
    p->split = p->method == JD_CONV_DIRECT && direct_conv_split_supported(kh, kw) && !opt_is_set(OPT_DIRECT_FP32) ? 1 : 0;
    p->nspec = p->method == JD_CONV_SEPARABLE ? (sep_conv_operator_floats() + 1) / 2
                                              : direct_conv_fragment_floats(kh, kw, p->split);
    p->partials_cap = std::max(std::max(poisson_fused_max_partials(H, W), sep_conv_tiles(H, W)), direct_conv_tiles(H, W));
    int rc = JD_OK;
    if (hipMalloc(&p->partials, (size_t)p->partials_cap * sizeof(double)) != hipSuccess)
      rc = fail(JD_ERR_ALLOC, "jd_conv_plan_create: hipMalloc of the partial sums failed");
    if (!rc) rc = ensure_component_buffers(p, 1);
    if (rc) {
      jd_conv_plan_destroy(p);
      return rc;
    }
    *plan_out = p;
    return JD_OK;
  }
  if (native_ok) {
    // FFT method on the hand-written transforms: works on the un-padded (H, W) grid like the direct kernels
    jd_conv_plan* p = new (std::nothrow) jd_conv_plan();
    if (!p) return fail(JD_ERR_ALLOC, "jd_conv_plan_create: out of host memory");
    p->method = JD_CONV_FFT, p->native = true;
    p->H = H, p->W = W, p->kh = kh, p->kw = kw, p->Hp = H, p->Wp = W;
    p->oy = (kh - 1) / 2, p->ox = (kw - 1) / 2, p->py = 0, p->px = 0;
    int rc = fftn_create(&p->fftn, H, W, kh, kw);
    p->nspec = fftn_spectrum_elements(p->fftn);
    p->partials_cap = poisson_fused_max_partials(H, W);
    if (!rc && hipMalloc(&p->partials, (size_t)p->partials_cap * sizeof(double)) != hipSuccess)
      rc = fail(JD_ERR_ALLOC, "jd_conv_plan_create: hipMalloc of the partial sums failed");
    if (!rc) rc = ensure_component_buffers(p, 1);
    if (rc) {
      jd_conv_plan_destroy(p);
      return rc;
    }
    *plan_out = p;
    return JD_OK;
  }
  static std::once_flag once;
  static rocfft_status setup_status = rocfft_status_success;
  std::call_once(once, [] { setup_status = rocfft_setup(); });
  if (setup_status != rocfft_status_success) return fail(JD_ERR_FFT, "rocfft_setup failed (%d)", (int)setup_status);

  jd_conv_plan* p = new (std::nothrow) jd_conv_plan();
  if (!p) return fail(JD_ERR_ALLOC, "jd_conv_plan_create: out of host memory");
  p->H = H, p->W = W, p->kh = kh, p->kw = kw;
  const int fh = H + kh - 1, fw = W + kw - 1;
  p->Hp = exact_shape ? fh : next_fast_len(fh, 2);
  p->Wp = exact_shape ? fw : next_fast_len(fw, 4);
  if (const char* env = getenv("JD_FFT_FORCE_PAD")) {  // tuning only: "Hp:Wp" (must be >= the full size, Wp % 4 == 0)
    int hp = 0, wp = 0;
    if (sscanf(env, "%d:%d", &hp, &wp) == 2 && hp >= fh && wp >= fw && wp % 4 == 0) p->Hp = hp, p->Wp = wp;
  }
  p->oy = (kh - 1) / 2;  // `_centered`: (full - new) // 2   (utils/torch.py:337-344)
  p->ox = (kw - 1) / 2;
  p->py = p->oy, p->px = p->ox;
  p->nspec = (size_t)p->Hp * (p->Wp / 2 + 1);

  auto cleanup = [&](int rc) {
    jd_conv_plan_destroy(p);
    return rc;
  };
  const size_t lengths[2] = {(size_t)p->Wp, (size_t)p->Hp};  // rocFFT: fastest dimension first
  rocfft_status s;
  s = rocfft_plan_create(&p->fwd, rocfft_placement_notinplace, rocfft_transform_type_real_forward,
                         rocfft_precision_single, 2, lengths, 1, nullptr);
  if (s != rocfft_status_success) return cleanup(fail(JD_ERR_FFT, "rocfft_plan_create(R2C %dx%d) failed (%d)", p->Hp, p->Wp, (int)s));
  s = rocfft_plan_create(&p->inv, rocfft_placement_notinplace, rocfft_transform_type_real_inverse,
                         rocfft_precision_single, 2, lengths, 1, nullptr);
  if (s != rocfft_status_success) return cleanup(fail(JD_ERR_FFT, "rocfft_plan_create(C2R %dx%d) failed (%d)", p->Hp, p->Wp, (int)s));
  s = rocfft_execution_info_create(&p->info);
  if (s != rocfft_status_success) return cleanup(fail(JD_ERR_FFT, "rocfft_execution_info_create failed (%d)", (int)s));
  size_t wf = 0, wi = 0;
  rocfft_plan_get_work_buffer_size(p->fwd, &wf);
  rocfft_plan_get_work_buffer_size(p->inv, &wi);
  p->work_bytes = wf > wi ? wf : wi;
  if (p->work_bytes) {
    if (hipMalloc(&p->work, p->work_bytes) != hipSuccess)
      return cleanup(fail(JD_ERR_ALLOC, "jd_conv_plan_create: hipMalloc(%zu) for the rocFFT work buffer failed", p->work_bytes));
    s = rocfft_execution_info_set_work_buffer(p->info, p->work, p->work_bytes);
    if (s != rocfft_status_success) return cleanup(fail(JD_ERR_FFT, "rocfft_execution_info_set_work_buffer failed (%d)", (int)s));
  }
  if (hipMalloc(&p->spec, p->nspec * sizeof(float2)) != hipSuccess)
    return cleanup(fail(JD_ERR_ALLOC, "jd_conv_plan_create: hipMalloc of the spectrum buffer failed"));
  p->partials_cap = poisson_fused_max_partials(p->Hp, p->Wp);
  if (hipMalloc(&p->partials, (size_t)p->partials_cap * sizeof(double)) != hipSuccess)
    return cleanup(fail(JD_ERR_ALLOC, "jd_conv_plan_create: hipMalloc of the partial sums failed"));
  int rc = ensure_component_buffers(p, 1);
  if (rc) return cleanup(rc);
  *plan_out = p;
  return JD_OK;
}

extern "C" int jd_conv_plan_destroy(jd_conv_plan* p) {
  if (!p) return JD_OK;
  (void)hipDeviceSynchronize();
  if (p->native) fftn_destroy(&p->fftn);
  for (int i = 0; i < jd::FFT_MAX_BATCH - 1; ++i) {
    if (p->fft_extra_spec[i]) (void)hipFree(p->fft_extra_spec[i]);
    if (p->fft_extra_work[i]) (void)hipFree(p->fft_extra_work[i]);
  }
  for (auto* t : p->fft_batch_dev)
    if (t) (void)hipFree(t);
  if (p->partials_shift_batch) (void)hipFree(p->partials_shift_batch);
  for (int i = 0; i < jd::FFT_MAX_BATCH; ++i)
    if (p->gshift_batch[i]) (void)hipFree(p->gshift_batch[i]);
  if (p->fwd) rocfft_plan_destroy(p->fwd);
  if (p->inv) rocfft_plan_destroy(p->inv);
  if (p->info) rocfft_execution_info_destroy(p->info);
  if (p->work) (void)hipFree(p->work);
  if (p->spec) (void)hipFree(p->spec);
  if (p->partials) (void)hipFree(p->partials);
  if (p->partials_cal) (void)hipFree(p->partials_cal);
  if (p->partials_batch) (void)hipFree(p->partials_batch);
  for (auto* t : p->table_dev)
    if (t) (void)hipFree(t);
  for (float* g : p->gbatch)
    if (g) (void)hipFree(g);
  for (int c = 0; c < JD_MAX_COMPONENTS; ++c) {
    if (p->pad[c]) (void)hipFree(p->pad[c]);
    if (p->conv[c]) (void)hipFree(p->conv[c]);
    if (p->shifted[c]) (void)hipFree(p->shifted[c]);
    if (p->gshift[c]) (void)hipFree(p->gshift[c]);
  }
  delete p;
  return JD_OK;
}

extern "C" int jd_conv_plan_shape(const jd_conv_plan* p, int* shape6) {
  JD_REQUIRE(p && shape6, "jd_conv_plan_shape: null argument");
  shape6[0] = p->H, shape6[1] = p->W, shape6[2] = p->Hp, shape6[3] = p->Wp, shape6[4] = p->oy, shape6[5] = p->ox;
  return JD_OK;
}

extern "C" size_t jd_conv_plan_spectrum_size(const jd_conv_plan* p) { return p ? p->nspec : 0; }

extern "C" int jd_conv_plan_method(const jd_conv_plan* p) { return p ? p->method : -1; }

extern "C" int jd_conv_native_fft_supported(int H, int W, int kh, int kw) {
  return H > 0 && W > 0 && kh > 0 && kw > 0 && opt_value(OPT_FFT_NATIVE, 1) != 0 && fftn_supported(H, W, kh, kw) ? 1 : 0;
}

extern "C" int jd_conv_plan_takes_walk(const jd_conv_plan* p, int n_datasets) {
  if (!p || p->method != JD_CONV_SEPARABLE || n_datasets < 1) return 0;
  return walk_takes_launch(p->H, p->W, n_datasets, p->kh, p->kw, p->oy, p->ox) ? 1 : 0;
}

extern "C" int jd_conv_operator_walk_frame(const jd_conv_plan* p, const float* khat) {
  if (!p || !khat || p->method != JD_CONV_SEPARABLE) return 0;
  return walk_operator_frame(khat, p->kh, p->kw, p->oy, p->ox);
}

extern "C" int jd_conv_operator_forget(const float* khat) {
  if (khat) sep_forget_operator(khat);
  return JD_OK;
}

extern "C" int jd_psf_separable_rank(const float* psf_host, int kh, int kw, float tol) {
  if (!psf_host || !sep_conv_supported(kh, kw)) return 0;
  return sep_factorize(psf_host, kh, kw, tol > 0.f ? (double)tol : SEP_DEFAULT_TOL, nullptr, nullptr);
}

extern "C" int jd_conv_psf_spectrum(jd_conv_plan* p, const float* psf, float* khat, void* stream) {
  JD_REQUIRE(p && psf && khat, "jd_conv_psf_spectrum: null argument");
  hipStream_t s = as_stream(stream);
  if (p->method == JD_CONV_DIRECT)
    return launch_toeplitz_fragments(psf, khat, khat + p->nspec, p->kh, p->kw, p->split, s);
  if (p->method == JD_CONV_SEPARABLE) {
    // setup-time only: the factorisation runs on the host (a PSF is a few KB), so this call synchronises
    std::vector<float> host((size_t)p->kh * p->kw), op;
    JD_HIP(hipMemcpyAsync(host.data(), psf, host.size() * sizeof(float), hipMemcpyDeviceToHost, s));
    JD_HIP(hipStreamSynchronize(s));
    SepOpInfo info;
    const int rank = sep_build_operator(host.data(), p->kh, p->kw, p->oy, p->ox, SEP_DEFAULT_TOL, &op, &info);
    JD_REQUIRE(rank > 0, "jd_conv_psf_spectrum: the %dx%d PSF is not a sum of <= %d outer products to %.0e of its sum "
               "(ask jd_psf_separable_rank() first); use JD_CONV_MODE_AUTO", p->kh, p->kw, SEP_MAX_RANK, SEP_DEFAULT_TOL);
    JD_HIP(hipMemcpyAsync(khat, op.data(), op.size() * sizeof(float), hipMemcpyHostToDevice, s));
    JD_HIP(hipStreamSynchronize(s));
    sep_register_operator(khat, info);
    return JD_OK;
  }
  if (p->native) return fftn_spectrum(p->fftn, psf, reinterpret_cast<float2*>(khat), s);
  int rc = launch_pad_mul(psf, nullptr, p->pad[0], p->kh, p->kw, p->Hp, p->Wp, s);
  if (rc) return rc;
  if ((rc = exec_fft(p, p->fwd, p->pad[0], p->spec, s))) return rc;
  const size_t n = p->nspec * 2;
  size_t blocks = (n + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  const float scale = (float)(1.0 / ((double)p->Hp * (double)p->Wp));
  scale_copy_kernel<<<(unsigned)blocks, 256, 0, s>>>(reinterpret_cast<const float*>(p->spec), khat, n, scale);
  JD_LAUNCH_CHECK();
  return JD_OK;
}

extern "C" int jd_conv_same(jd_conv_plan* p, const float* image, const float* scale_image, const float* khat,
                            float* out, void* stream) {
  JD_REQUIRE(p && image && khat && out, "jd_conv_same: null argument");
  hipStream_t s = as_stream(stream);
  if (p->method == JD_CONV_DIRECT)
    return launch_direct_conv(image, scale_image, khat, out, nullptr, p->H, p->W, p->kh, p->kw, p->oy, p->ox, 0, 1.f, 0,
                              p->split, s);
  if (p->method == JD_CONV_SEPARABLE)
    return launch_sep_conv(image, scale_image, khat, out, nullptr, p->H, p->W, p->kh, p->kw, p->oy, p->ox, 0, 1.f, 0, s);
  if (p->native)
    return fftn_conv(p->fftn, image, scale_image, reinterpret_cast<const float2*>(khat), out, nullptr, 0, 1.f, 0, s);
  int rc = conv_forward(p, 0, image, scale_image, khat, s);
  if (rc) return rc;
  return launch_crop(p->conv[0], out, p->H, p->W, p->Wp, p->oy, p->ox, s);
}

extern "C" int jd_conv_same_adjoint(jd_conv_plan* p, const float* grad_out, const float* scale_image,
                                    const float* khat, float* grad_image, int accumulate, void* stream) {
  JD_REQUIRE(p && grad_out && khat && grad_image, "jd_conv_same_adjoint: null argument");
  hipStream_t s = as_stream(stream);
  if (p->method == JD_CONV_DIRECT)
    return launch_direct_conv(grad_out, nullptr, khat + p->nspec, grad_image, scale_image, p->H, p->W, p->kh, p->kw,
                              p->oy, p->ox, 1, 1.f, accumulate, p->split, s);
  if (p->method == JD_CONV_SEPARABLE)
    return launch_sep_conv(grad_out, nullptr, khat, grad_image, scale_image, p->H, p->W, p->kh, p->kw, p->oy, p->ox, 1,
                           1.f, accumulate, s);
  if (p->native)
    return fftn_conv(p->fftn, grad_out, nullptr, reinterpret_cast<const float2*>(khat), grad_image, scale_image, 1, 1.f, accumulate, s);
  int rc = launch_pad_mul(grad_out, nullptr, p->pad[0], p->H, p->W, p->Hp, p->Wp, s);
  if (rc) return rc;
  return corr_backward_into(p, 0, khat, scale_image, grad_image, 1.f, accumulate, s);
}

namespace {
struct Calibration {
  const float* shift_xy = nullptr;      // device [2] = {x, y}; null: no shift
  float shift_scale = 1.f;              // up-sampling factor (shift is given in counts pixels)
  const float* log_bkg_norm = nullptr;  // device [1]; null: background unscaled
  float* grad_shift_xy = nullptr;       // device [2], overwritten; null: not wanted
  float* grad_log_bkg_norm = nullptr;   // device [1], overwritten; null: not wanted
};
}  // namespace

static int npred_poisson_impl(const char* who, jd_conv_plan* p, int n_comp, const float* const* flux,
                              const float* const* exposure, const float* const* khat, const float* background,
                              const float* counts, float stirling_mean, float eps, float* loss_out,
                              float* const* grad_flux, int accumulate, float grad_scale, float* npred_out,
                              int upsampling, const Calibration& cal, void* stream) {
  JD_REQUIRE(p && flux && exposure && khat && background && counts && loss_out, "%s: null argument", who);
  JD_REQUIRE(upsampling >= 1 && upsampling <= 8 && p->H % upsampling == 0 && p->W % upsampling == 0,
             "%s: upsampling = %d must be in [1, 8] and divide the flux grid (%d, %d)", who, upsampling, p->H, p->W);
  JD_REQUIRE(n_comp >= 1 && n_comp <= JD_MAX_COMPONENTS, "%s: n_comp = %d not in [1, %d]", who, n_comp,
             JD_MAX_COMPONENTS);
  for (int c = 0; c < n_comp; ++c) {
    JD_REQUIRE(flux[c] && khat[c], "%s: flux[%d] or khat[%d] is null", who, c, c);
    if (grad_flux) JD_REQUIRE(grad_flux[c], "%s: grad_flux[%d] is null", who, c);
  }
  hipStream_t s = as_stream(stream);
  int rc = ensure_component_buffers(p, n_comp);
  if (rc) return rc;
  const bool calibrated = cal.shift_xy || cal.log_bkg_norm;
  if (calibrated && (rc = ensure_calibration_buffers(p, n_comp, cal.shift_xy != nullptr))) return rc;

  const double n_pix = (double)(p->H / upsampling) * (double)(p->W / upsampling);
  int n_partials = 0;
  bool fold_loss = false;  // the loss of the fused single-component path is finalised by the adjoint launch (see below)
  // one component convolved on the unpadded grid (separable or MFMA direct), no up-sampling, no background norm: the
  // Poisson pass is the epilogue of the convolution
  if (p->native && n_comp == 1 && upsampling == 1 && !calibrated && grad_flux && !npred_out && !opt_is_set(OPT_SEP_NO_FUSION) &&
      p->partials_cap >= p->fftn.Hh) {
    // native FFT path, one component: rows, columns, rows^-1 + Poisson pass + rows of g, columns, rows^-1 + adjoint epilogue
    return fftn_poisson_step(p->fftn, flux[0], exposure[0], reinterpret_cast<const float2*>(khat[0]), background, counts,
                             p->partials, &n_partials, eps, (float)(1.0 / n_pix), grad_flux[0], grad_scale, accumulate, s,
                             1.0 / n_pix, (double)stirling_mean, loss_out);
  }
  // native FFT path with up-sampling (and optionally the calibration): the sum-pool, the Poisson pass and the row transform
  // of the up-sampled g are one launch between the two column passes; the loss and the background-norm gradient are
  // finalised by the adjoint's last launch
  if (p->native && n_comp == 1 && upsampling > 1 && fftn_pooled_supported(p->fftn, upsampling) && grad_flux && !npred_out &&
      !opt_is_set(OPT_SEP_NO_FUSION) && p->partials_cap >= p->fftn.Hh) {
    const bool want_norm = cal.log_bkg_norm && cal.grad_log_bkg_norm;
    if ((rc = fftn_poisson_step_pooled(p->fftn, upsampling, flux[0], exposure[0], reinterpret_cast<const float2*>(khat[0]), background,
                                       counts, cal.log_bkg_norm, p->partials, want_norm ? p->partials_cal : nullptr, eps,
                                       (float)(1.0 / n_pix), cal.shift_xy ? p->gshift[0] : grad_flux[0], grad_scale,
                                       cal.shift_xy ? 0 : accumulate, s, 1.0 / n_pix, (double)stirling_mean, loss_out,
                                       (double)grad_scale, want_norm ? cal.grad_log_bkg_norm : nullptr, cal.shift_xy,
                                       cal.shift_scale)))
      return rc;
    if (!cal.shift_xy) return JD_OK;
    int n_blocks = 0;
    if ((rc = launch_shift_bwd(flux[0], p->gshift[0], grad_flux[0], accumulate, p->H, p->W, cal.shift_xy, cal.shift_scale,
                               p->partials_cal, &n_blocks, s)))
      return rc;
    return cal.grad_shift_xy ? launch_finalize_multi(p->partials_cal, n_blocks, 2, 1.0, cal.grad_shift_xy, 0, s) : JD_OK;
  }
  const bool fused = (p->method == JD_CONV_SEPARABLE || p->method == JD_CONV_DIRECT) && n_comp == 1 &&
                     upsampling == 1 && !cal.log_bkg_norm && !opt_is_set(OPT_SEP_NO_FUSION);
  if (fused) {
    const float* in = flux[0];
    if (cal.shift_xy) {
      if ((rc = launch_shift_fwd(flux[0], p->shifted[0], p->H, p->W, cal.shift_xy, cal.shift_scale, s))) return rc;
      in = p->shifted[0];
    }
    if (p->method == JD_CONV_SEPARABLE && grad_flux && !npred_out) {
      // forward model, Poisson pass and adjoint in one launch where the strip-walk kernels apply
      SepBatchTable one{};
      one.scale[0] = exposure[0], one.op[0] = khat[0], one.bkg[0] = background, one.cnt[0] = counts;
      float* target = cal.shift_xy ? p->gshift[0] : grad_flux[0];
      rc = walk_joint_step(1, in, one, nullptr, target, p->H, p->W, p->kh, p->kw, p->oy, p->ox, p->partials, eps,
                           (float)(1.0 / n_pix), grad_scale, cal.shift_xy ? 0 : accumulate, &n_partials, s);
      if (rc != JD_WALK_NOT_TAKEN) {
        if (rc) return rc;
        if ((rc = launch_finalize_sum(p->partials, n_partials, 1.0 / n_pix, (double)stirling_mean, loss_out, 0, s))) return rc;
        if (!cal.shift_xy) return JD_OK;
        int n_blocks = 0;
        if ((rc = launch_shift_bwd(flux[0], p->gshift[0], grad_flux[0], accumulate, p->H, p->W, cal.shift_xy, cal.shift_scale,
                                   p->partials_cal, &n_blocks, s)))
          return rc;
        return cal.grad_shift_xy ? launch_finalize_multi(p->partials_cal, n_blocks, 2, 1.0, cal.grad_shift_xy, 0, s) : JD_OK;
      }
    }
    rc = p->method == JD_CONV_SEPARABLE
             ? launch_sep_conv_poisson(in, exposure[0], khat[0], p->pad[0], p->H, p->W, p->kh, p->kw, p->oy, p->ox,
                                       background, counts, npred_out, p->partials, eps, (float)(1.0 / n_pix),
                                       grad_flux ? 1 : 0, &n_partials, s)
             : launch_direct_conv_poisson(in, exposure[0], khat[0], p->pad[0], p->H, p->W, p->kh, p->kw, p->oy, p->ox,
                                          background, counts, npred_out, p->partials, eps, (float)(1.0 / n_pix),
                                          grad_flux ? 1 : 0, &n_partials, p->split, s);
    if (rc) return rc;
    // the loss: by block 0 of the adjoint launch where that is the separable tile kernel (one dependent launch less),
    // by a launch of its own otherwise
    fold_loss = p->method == JD_CONV_SEPARABLE && grad_flux != nullptr;
    if (!fold_loss &&
        (rc = launch_finalize_sum(p->partials, n_partials, 1.0 / n_pix, (double)stirling_mean, loss_out, 0, s)))
      return rc;
    if (!grad_flux) return JD_OK;
  } else {
  // forward model per component (models/npred.py:175-179), after the calibration shift (:225-232)
  for (int c = 0; c < n_comp; ++c) {
    const float* in = flux[c];
    if (cal.shift_xy) {
      if ((rc = launch_shift_fwd(flux[c], p->shifted[c], p->H, p->W, cal.shift_xy, cal.shift_scale, s))) return rc;
      in = p->shifted[c];
    }
    if ((rc = conv_forward(p, c, in, exposure[c], khat[c], s))) return rc;
  }

  // fused clip + background + NLL + gradient (models/npred.py:191,254-261; loss.py:35-37)
  PoissonArgs a{};
  for (int c = 0; c < n_comp; ++c) {
    a.conv[c] = p->conv[c];
    a.g[c] = p->pad[c];
  }
  a.background = background, a.counts = counts, a.npred_out = npred_out, a.partials = p->partials;
  a.n_comp = n_comp, a.H = p->H, a.W = p->W, a.Hp = p->Hp, a.Wp = p->Wp, a.oy = p->py, a.ox = p->px;
  a.eps = eps;
  a.up = upsampling;
  a.log_bkg_norm = cal.log_bkg_norm;
  a.partials_b = (cal.log_bkg_norm && cal.grad_log_bkg_norm && grad_flux) ? p->partials_cal : nullptr;
  // the loss is the mean over the COUNTS pixels (loss.py:35-37)
  a.inv_n = (float)(1.0 / n_pix);
  a.write_grad = grad_flux ? 1 : 0;
  if ((rc = upsampling > 1 ? launch_poisson_pooled(a, &n_partials, s) : launch_poisson_fused(a, &n_partials, s)))
    return rc;
  if ((rc = launch_finalize_sum(p->partials, n_partials, 1.0 / n_pix, (double)stirling_mean, loss_out, 0, s)))
    return rc;
  if (!grad_flux) return JD_OK;
  if (a.partials_b &&
      (rc = launch_finalize_sum(p->partials_cal, n_partials, (double)grad_scale, 0.0, cal.grad_log_bkg_norm, 0, s)))
    return rc;
  }

  // adjoint: d loss / d flux_c = [shift^T] ( E_c * corr(psf_c, g_c) )
  const SepLossFold fold{p->partials, n_partials, 1.0 / n_pix, (double)stirling_mean, loss_out};
  for (int c = 0; c < n_comp; ++c) {
    int folded = 0;
    const bool ask = fold_loss && c == 0;
    if (!cal.shift_xy) {
      rc = corr_backward_into(p, c, khat[c], exposure[c], grad_flux[c], grad_scale, accumulate, s, ask ? &fold : nullptr, &folded);
    } else {
      rc = corr_backward_into(p, c, khat[c], exposure[c], p->gshift[c], grad_scale, 0, s, ask ? &fold : nullptr, &folded);
    }
    if (rc) return rc;
    if (ask && !folded &&
        (rc = launch_finalize_sum(p->partials, n_partials, 1.0 / n_pix, (double)stirling_mean, loss_out, 0, s)))
      return rc;
    if (!cal.shift_xy) continue;
    int n_blocks = 0;
    if ((rc = launch_shift_bwd(flux[c], p->gshift[c], grad_flux[c], accumulate, p->H, p->W, cal.shift_xy,
                               cal.shift_scale, p->partials_cal, &n_blocks, s)))
      return rc;
    if (cal.grad_shift_xy &&
        (rc = launch_finalize_multi(p->partials_cal, n_blocks, 2, 1.0, cal.grad_shift_xy, c > 0, s)))
      return rc;
  }
  return JD_OK;
}

extern "C" int jd_npred_poisson_fwd_bwd(jd_conv_plan* p, int n_comp, const float* const* flux,
                                        const float* const* exposure, const float* const* khat,
                                        const float* background, const float* counts, float stirling_mean,
                                        float eps, float* loss_out, float* const* grad_flux, int accumulate,
                                        float grad_scale, float* npred_out, int upsampling, void* stream) {
  return npred_poisson_impl("jd_npred_poisson_fwd_bwd", p, n_comp, flux, exposure, khat, background, counts,
                            stirling_mean, eps, loss_out, grad_flux, accumulate, grad_scale, npred_out, upsampling,
                            Calibration{}, stream);
}

extern "C" int jd_npred_poisson_batch_multi_fwd_bwd(jd_conv_plan* p, int n_datasets, int n_comp, const float* const* flux,
                                                    const float* const* exposure, const float* const* khat,
                                                    const float* const* background, const float* const* counts,
                                                    const float* stirling_mean, float eps, float* const* loss_out,
                                                    float* const* grad_flux, int accumulate, float grad_scale,
                                                    void* stream) {
  const char* who = "jd_npred_poisson_batch_multi_fwd_bwd";
  JD_REQUIRE(p && flux && exposure && khat && background && counts && stirling_mean && loss_out, "%s: null argument", who);
  JD_REQUIRE(p->method == JD_CONV_SEPARABLE || (p->native && n_comp == 1),
             "%s: the plan must use the separable method, or the native FFT path with one flux component (one "
             "jd_npred_poisson_fwd_bwd per dataset otherwise)", who);
  if (p->native) {
    // native FFT path: every launch of the likelihood step covers all datasets (fftn_poisson_step_batch); forward-only
    // evaluations (the trace) and switched-off fusion run the per-dataset calls this stands for
    JD_REQUIRE(flux[0], "%s: flux[0] is null", who);
    for (int d = 0; d < n_datasets; ++d)
      JD_REQUIRE(exposure[d] && khat[d] && background[d] && counts[d] && loss_out[d], "%s: null pointer for dataset %d", who, d);
    hipStream_t s = as_stream(stream);
    // (block d of the last launch's Hh blocks finalises the loss of dataset d: a tiny image with more datasets than row
    // pairs runs the per-dataset calls)
    if (!grad_flux || !grad_flux[0] || n_datasets > FFT_MAX_BATCH || n_datasets < 2 || n_datasets > p->fftn.Hh ||
        opt_is_set(OPT_SEP_NO_FUSION) || opt_value(OPT_FFT_BATCH, 1) == 0) {
      for (int d = 0; d < n_datasets; ++d) {
        int rc = npred_poisson_impl(who, p, 1, flux, exposure + d, khat + d, background[d], counts[d], stirling_mean[d], eps,
                                    loss_out[d], grad_flux, accumulate || d > 0, grad_scale, nullptr, 1, Calibration{}, stream);
        if (rc) return rc;
      }
      return JD_OK;
    }
    const FftNative& fn = p->fftn;
    for (int d = 0; d + 1 < n_datasets; ++d) {
      if (!p->fft_extra_spec[d]) JD_HIP(hipMalloc(&p->fft_extra_spec[d], (size_t)fn.Hh * fn.Nx * sizeof(float2)));
      if (!p->fft_extra_work[d]) JD_HIP(hipMalloc(&p->fft_extra_work[d], (size_t)fn.Ny * fn.Nx * sizeof(float2)));
    }
    if (p->partials_batch_cap < n_datasets * fn.Hh) {
      if (p->partials_batch) (void)hipFree(p->partials_batch);
      p->partials_batch = nullptr, p->partials_batch_cap = 0;
      JD_HIP(hipMalloc(&p->partials_batch, (size_t)n_datasets * fn.Hh * sizeof(double)));
      p->partials_batch_cap = n_datasets * fn.Hh;
    }
    FftBatch batch{};
    batch.n = n_datasets;
    for (int d = 0; d < n_datasets; ++d) {
      batch.exposure[d] = exposure[d], batch.khat[d] = reinterpret_cast<const float2*>(khat[d]);
      batch.background[d] = background[d], batch.counts[d] = counts[d];
      batch.loss_out[d] = loss_out[d], batch.loss_offset[d] = stirling_mean[d];
      batch.spec[d] = d ? p->fft_extra_spec[d - 1] : fn.spec, batch.work[d] = d ? p->fft_extra_work[d - 1] : fn.work;
    }
    int slot = 0;
    if (int rc = fft_batch_table(p, batch, s, &slot)) return rc;
    const double n_pix = (double)p->H * (double)p->W;
    return fftn_poisson_step_batch(fn, n_datasets, p->fft_batch_dev[slot], flux[0], p->partials_batch, eps, (float)(1.0 / n_pix),
                                   grad_flux[0], grad_scale, accumulate, s, 1.0 / n_pix);
  }
  JD_REQUIRE(n_datasets >= 1 && n_datasets <= SEP_MAX_BATCH, "%s: n_datasets = %d not in [1, %d]", who, n_datasets,
             SEP_MAX_BATCH);
  JD_REQUIRE(n_comp >= 1 && n_comp <= SEP_BATCH_MAX_COMP, "%s: n_comp = %d not in [1, %d]", who, n_comp, SEP_BATCH_MAX_COMP);
  for (int c = 0; c < n_comp; ++c) {
    JD_REQUIRE(flux[c], "%s: flux[%d] is null", who, c);
    if (grad_flux) JD_REQUIRE(grad_flux[c], "%s: grad_flux[%d] is null", who, c);
  }
  for (int d = 0; d < n_datasets; ++d) {
    JD_REQUIRE(background[d] && counts[d] && loss_out[d], "%s: null pointer for dataset %d", who, d);
    for (int c = 0; c < n_comp; ++c)
      JD_REQUIRE(exposure[d * n_comp + c] && khat[d * n_comp + c], "%s: null exposure or operator for dataset %d, component %d",
                 who, d, c);
  }
  hipStream_t s = as_stream(stream);
  {
    // a batch whose datasets would not all take the same kernel family (some PSFs rank 1 and small: strip-walk kernels,
    // others not) runs the per-dataset calls it stands for -- same results by definition
    SepBatchTable probe{};
    for (int d = 0; d < n_datasets; ++d) probe.bkg[d] = background[d], probe.cnt[d] = counts[d];
    for (int i = 0; i < n_datasets * n_comp; ++i) probe.scale[i] = exposure[i], probe.op[i] = khat[i], probe.g[i] = nullptr;
    if (sep_batch_is_mixed(n_datasets, n_comp, probe, p->H, p->W, p->kh, p->kw, p->oy, p->ox)) {
      for (int d = 0; d < n_datasets; ++d) {
        int rc = npred_poisson_impl(who, p, n_comp, flux, exposure + d * n_comp, khat + d * n_comp, background[d], counts[d],
                                    stirling_mean[d], eps, loss_out[d], grad_flux, accumulate || d > 0, grad_scale, nullptr,
                                    1, Calibration{}, stream);
        if (rc) return rc;
      }
      return JD_OK;
    }
  }
  const size_t bytes = (size_t)p->H * p->W * sizeof(float);
  for (int i = 0; i < n_datasets * n_comp; ++i)
    if (!p->gbatch[i]) JD_HIP(hipMalloc(&p->gbatch[i], bytes));
  const int tiles = sep_conv_tiles(p->H, p->W);
  if (p->partials_batch_cap < n_datasets * tiles) {
    // (launches of an earlier call may still read the old buffer: hipFree waits for the device)
    if (p->partials_batch) (void)hipFree(p->partials_batch);
    p->partials_batch = nullptr, p->partials_batch_cap = 0;
    JD_HIP(hipMalloc(&p->partials_batch, (size_t)n_datasets * tiles * sizeof(double)));
    p->partials_batch_cap = n_datasets * tiles;
  }
  SepBatchTable table{};
  for (int d = 0; d < n_datasets; ++d) table.bkg[d] = background[d], table.cnt[d] = counts[d];
  for (int i = 0; i < n_datasets * n_comp; ++i) table.scale[i] = exposure[i], table.op[i] = khat[i], table.g[i] = p->gbatch[i];
  for (int d = 0; d < n_datasets; ++d) table.loss_out[d] = loss_out[d], table.loss_offset[d] = stirling_mean[d];
  walk_batch_order(table, n_datasets, n_comp, p->kh, p->kw, p->oy, p->ox);
  // a session passes the same pointers every step: look the table up by content, upload only a new one
  int slot = -1, victim = 0;
  for (int i = 0; i < jd_conv_plan::N_TABLES; ++i) {
    if (p->table_used[i] && memcmp(&table, &p->table_host[i], sizeof(table)) == 0) slot = i;
    if (p->table_used[i] < p->table_used[victim]) victim = i;
  }
  if (slot < 0) {
    // least recently used slot: wait for launches that may still read it, then copy synchronously (stack source)
    slot = victim;
    if (!p->table_dev[slot]) JD_HIP(hipMalloc(&p->table_dev[slot], sizeof(SepBatchTable)));
    if (p->table_used[slot]) JD_HIP(hipStreamSynchronize(s));
    JD_HIP(hipMemcpy(p->table_dev[slot], &table, sizeof(table), hipMemcpyHostToDevice));
    p->table_host[slot] = table;
  }
  p->table_used[slot] = ++p->table_clock;
  SepBatchTable* const table_dev = p->table_dev[slot];
  const double n_pix = (double)p->H * (double)p->W;
  int n_part = 0;  // partial sums per dataset the forward launch wrote (<= tiles)
  if (n_comp == 1 && grad_flux) {
    // one launch for the whole likelihood step where the strip-walk kernels apply (the g images are never written)
    int rc = walk_joint_step(n_datasets, flux[0], table, table_dev, grad_flux[0], p->H, p->W, p->kh, p->kw, p->oy, p->ox,
                             p->partials_batch, eps, (float)(1.0 / n_pix), grad_scale, accumulate, &n_part, s);
    if (rc != JD_WALK_NOT_TAKEN)
      return rc ? rc : launch_finalize_rows(p->partials_batch, n_part, n_datasets, 1.0 / n_pix, stirling_mean, loss_out, s);
  }
  int rc = launch_sep_conv_poisson_batch(n_datasets, n_comp, flux, table, table_dev, p->H, p->W, p->kh, p->kw, p->oy, p->ox,
                                         p->partials_batch, eps, (float)(1.0 / n_pix), grad_flux ? 1 : 0, &n_part, s);
  if (rc) return rc;
  // with a gradient the losses are finalised by the first blocks of the first adjoint launch where it can (one dependent
  // launch less per step)
  if (!grad_flux) return launch_finalize_rows(p->partials_batch, n_part, n_datasets, 1.0 / n_pix, stirling_mean, loss_out, s);
  int folded = 0;
  // all components in one adjoint launch where the strip-walk kernels apply (several components, or 9-16 datasets)
  rc = walk_conv_adjoint_batch_all(n_datasets, n_comp, table, table_dev, grad_flux, p->H, p->W, p->kh, p->kw, p->oy, p->ox,
                                   grad_scale, accumulate, s, p->partials_batch, 1.0 / n_pix, n_part, &folded);
  if (rc != JD_WALK_NOT_TAKEN) {
    if (rc || folded) return rc;
    return launch_finalize_rows(p->partials_batch, n_part, n_datasets, 1.0 / n_pix, stirling_mean, loss_out, s);
  }
  for (int c = 0; c < n_comp; ++c) {
    int done = 0;
    if ((rc = launch_sep_conv_adjoint_batch(n_datasets, n_comp, c, table, table_dev, grad_flux[c], p->H, p->W, p->kh,
                                            p->kw, p->oy, p->ox, grad_scale, accumulate, s,
                                            c == 0 && n_datasets <= 8 ? p->partials_batch : nullptr, 1.0 / n_pix, n_part,
                                            &done)))
      return rc;
    folded |= done;
  }
  if (!folded) return launch_finalize_rows(p->partials_batch, n_part, n_datasets, 1.0 / n_pix, stirling_mean, loss_out, s);
  return JD_OK;
}

extern "C" int jd_npred_poisson_batch_fwd_bwd(jd_conv_plan* p, int n_datasets, const float* flux,
                                              const float* const* exposure, const float* const* khat,
                                              const float* const* background, const float* const* counts,
                                              const float* stirling_mean, float eps, float* const* loss_out,
                                              float* grad_flux, int accumulate, float grad_scale, void* stream) {
  JD_REQUIRE(flux, "jd_npred_poisson_batch_fwd_bwd: null argument");
  const float* fluxes[1] = {flux};
  float* grads[1] = {grad_flux};
  return jd_npred_poisson_batch_multi_fwd_bwd(p, n_datasets, 1, fluxes, exposure, khat, background, counts, stirling_mean,
                                              eps, loss_out, grad_flux ? grads : nullptr, accumulate, grad_scale, stream);
}

// The joint step over several CALIBRATED and / or UP-SAMPLED datasets of one flux component on the native FFT path
// (include/jolideco_hip.h).  Anything else -- another method, an up-sampling factor the fused launches do not take, a
// forward-only evaluation, switched-off fusion -- runs the per-dataset calls this stands for.
extern "C" int jd_npred_poisson_calibrated_batch_fwd_bwd(jd_conv_plan* p, int n_datasets, const float* flux,
                                                         const float* const* exposure, const float* const* khat,
                                                         const float* const* background, const float* const* counts,
                                                         const float* stirling_mean, float eps, float* const* loss_out,
                                                         float* grad_flux, int accumulate, float grad_scale, int upsampling,
                                                         const float* const* shift_xy, const float* const* log_background_norm,
                                                         float* const* grad_shift_xy, float* const* grad_log_background_norm,
                                                         void* stream) {
  const char* who = "jd_npred_poisson_calibrated_batch_fwd_bwd";
  JD_REQUIRE(p && flux && exposure && khat && background && counts && stirling_mean && loss_out, "%s: null argument", who);
  JD_REQUIRE(n_datasets >= 1, "%s: n_datasets = %d", who, n_datasets);
  JD_REQUIRE(upsampling >= 1 && upsampling <= 8 && p->H % upsampling == 0 && p->W % upsampling == 0,
             "%s: upsampling = %d must be in [1, 8] and divide the flux grid (%d, %d)", who, upsampling, p->H, p->W);
  for (int d = 0; d < n_datasets; ++d)
    JD_REQUIRE(exposure[d] && khat[d] && background[d] && counts[d] && loss_out[d], "%s: null pointer for dataset %d", who, d);
  hipStream_t s = as_stream(stream);
  const bool any_cal = shift_xy || log_background_norm;
  // Measured, 8 calibrated observations, up-sampling x2, launches over all datasets against per-dataset calls
  // (tools/gpu/cb1024.py): flux grid 512^2 204 against 499 us per step, 1024^2 330 against 590, 2048^2 855 against 1071.  At
  // 4096^2 (bench config c6) the round-4 kernels tied; with the round-5 kernels (tools/ab.py c6, one process): launches over
  // all datasets 4.33 ms per step, per-dataset calls 4.63, the FFT launches dataset by dataset on one set of work arrays with
  // only the tail over all datasets 4.63 (the tail alone: 383 us against 8 x 54).  So: every launch over all datasets, at
  // every size (JD_FFT_BATCH = 0: per-dataset calls; 3: per-dataset calls beyond 2048 rows, the round-4 rule; 4: beyond
  // 2048 rows the FFT launches dataset by dataset, the tail over all datasets).
  const int mode = opt_value(OPT_FFT_BATCH, 1);
  const bool batched = p->native && grad_flux && n_datasets >= 2 && n_datasets <= FFT_MAX_BATCH &&
                       fftn_pooled_supported(p->fftn, upsampling) && !opt_is_set(OPT_SEP_NO_FUSION) && mode != 0 &&
                       (p->fftn.Hh <= 1024 || mode != 3);
  const bool sequential = batched && p->fftn.Hh > 1024 && mode == 4;
  if (!batched) {
    const float* fluxes[1] = {flux};
    float* grads[1] = {grad_flux};
    for (int d = 0; d < n_datasets; ++d) {
      Calibration cal;
      cal.shift_xy = shift_xy ? shift_xy[d] : nullptr, cal.shift_scale = (float)upsampling;
      cal.log_bkg_norm = log_background_norm ? log_background_norm[d] : nullptr;
      cal.grad_shift_xy = grad_shift_xy ? grad_shift_xy[d] : nullptr;
      cal.grad_log_bkg_norm = grad_log_background_norm ? grad_log_background_norm[d] : nullptr;
      int rc = npred_poisson_impl(who, p, 1, fluxes, exposure + d, khat + d, background[d], counts[d], stirling_mean[d], eps,
                                  loss_out[d], grad_flux ? grads : nullptr, accumulate || d > 0, grad_scale, nullptr, upsampling,
                                  cal, stream);
      if (rc) return rc;
    }
    return JD_OK;
  }
  int rc = ensure_component_buffers(p, 1);
  if (rc) return rc;
  bool any_shift = false;
  for (int d = 0; d < n_datasets; ++d) any_shift = any_shift || (shift_xy && shift_xy[d]);
  if (any_cal && (rc = ensure_calibration_buffers(p, 1, any_shift))) return rc;
  const FftNative& fn = p->fftn;
  for (int d = 0; d + 1 < n_datasets && !sequential; ++d) {
    if (!p->fft_extra_spec[d]) JD_HIP(hipMalloc(&p->fft_extra_spec[d], (size_t)fn.Hh * fn.Nx * sizeof(float2)));
    if (!p->fft_extra_work[d]) JD_HIP(hipMalloc(&p->fft_extra_work[d], (size_t)fn.Ny * fn.Nx * sizeof(float2)));
  }
  const int per = fn.Hh / upsampling;
  if (p->partials_batch_cap < 2 * n_datasets * per) {  // [losses | background-norm gradients], n_datasets * per each
    if (p->partials_batch) (void)hipFree(p->partials_batch);
    p->partials_batch = nullptr, p->partials_batch_cap = 0;
    JD_HIP(hipMalloc(&p->partials_batch, (size_t)2 * n_datasets * per * sizeof(double)));
    p->partials_batch_cap = 2 * n_datasets * per;
  }
  for (int d = 0; d < n_datasets; ++d)
    if (!p->gshift_batch[d]) JD_HIP(hipMalloc(&p->gshift_batch[d], (size_t)p->H * p->W * sizeof(float)));
  const int shift_need = 2 * shift_bwd_max_blocks(p->H, p->W) * n_datasets;
  if (p->partials_shift_batch_cap < shift_need) {
    if (p->partials_shift_batch) (void)hipFree(p->partials_shift_batch);
    p->partials_shift_batch = nullptr, p->partials_shift_batch_cap = 0;
    JD_HIP(hipMalloc(&p->partials_shift_batch, (size_t)shift_need * sizeof(double)));
    p->partials_shift_batch_cap = shift_need;
  }
  FftBatch batch{};
  batch.n = n_datasets;
  for (int d = 0; d < n_datasets; ++d) {
    batch.exposure[d] = exposure[d], batch.khat[d] = reinterpret_cast<const float2*>(khat[d]);
    batch.background[d] = background[d], batch.counts[d] = counts[d];
    batch.loss_out[d] = loss_out[d], batch.loss_offset[d] = stirling_mean[d];
    batch.spec[d] = d && !sequential ? p->fft_extra_spec[d - 1] : fn.spec;
    batch.work[d] = d && !sequential ? p->fft_extra_work[d - 1] : fn.work;
    batch.shift_xy[d] = shift_xy ? shift_xy[d] : nullptr;
    batch.log_bkg_norm[d] = log_background_norm ? log_background_norm[d] : nullptr;
    batch.grad_shift_xy[d] = grad_shift_xy ? grad_shift_xy[d] : nullptr;
    batch.grad_log_bkg_norm[d] = (batch.log_bkg_norm[d] && grad_log_background_norm) ? grad_log_background_norm[d] : nullptr;
    batch.gshift[d] = p->gshift_batch[d];
  }
  int slot = 0;
  if ((rc = fft_batch_table(p, batch, s, &slot))) return rc;
  const double n_pix = (double)(p->H / upsampling) * (double)(p->W / upsampling);
  return fftn_poisson_step_pooled_batch(fn, upsampling, n_datasets, p->fft_batch_dev[slot], p->fft_batch_host[slot], flux, p->partials_batch,
                                        p->partials_batch + (size_t)n_datasets * per, eps, (float)(1.0 / n_pix), grad_flux,
                                        p->partials_shift_batch, grad_scale, accumulate, s, 1.0 / n_pix, (double)grad_scale,
                                        sequential ? 1 : 0);
}

extern "C" int jd_npred_poisson_calibrated_fwd_bwd(jd_conv_plan* p, int n_comp, const float* const* flux,
                                                   const float* const* exposure, const float* const* khat,
                                                   const float* background, const float* counts,
                                                   float stirling_mean, float eps, float* loss_out,
                                                   float* const* grad_flux, int accumulate, float grad_scale,
                                                   float* npred_out, int upsampling, const float* shift_xy,
                                                   const float* log_background_norm, float* grad_shift_xy,
                                                   float* grad_log_background_norm, void* stream) {
  Calibration cal;
  cal.shift_xy = shift_xy, cal.shift_scale = (float)upsampling, cal.log_bkg_norm = log_background_norm;
  cal.grad_shift_xy = grad_shift_xy, cal.grad_log_bkg_norm = grad_log_background_norm;
  return npred_poisson_impl("jd_npred_poisson_calibrated_fwd_bwd", p, n_comp, flux, exposure, khat, background, counts,
                            stirling_mean, eps, loss_out, grad_flux, accumulate, grad_scale, npred_out, upsampling, cal,
                            stream);
}
