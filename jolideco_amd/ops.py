"""Python handles over the C-ABI of libjolideco_hip.so and the autograd seams built on them.

`ConvPlan` / `GmmHandle` own the native handles; the `*Function` classes are the
`torch.autograd.Function` wrappers that sit where the reference calls ATen ops
(SURVEY.md section 8(b)).  Nothing in this module computes on the CPU: every method requires HIP
tensors and raises RuntimeError otherwise.
"""
import ctypes
import math
import weakref
from ctypes import c_float, c_int, c_void_p

import numpy as np
import torch

from . import _hip
from ._hip import check, ptr, ptr_array, stream_ptr

__all__ = [
    "ConvPlan",
    "GmmHandle",
    "ConvSameFunction",
    "PoissonNLLFunction",
    "GMMPatchPriorFunction",
    "ElementwisePriorFunction",
    "stirling_mean",
    "require_hip_tensor",
    "band_rows",
    "add_rolled_bands",
]

POISSON_EPS = 1e-25  # jolideco/loss.py:36


def require_hip_tensor(t, name="tensor"):
    if not (isinstance(t, torch.Tensor) and t.is_cuda):
        raise RuntimeError(f"{name} must live on a HIP device: jolideco_amd has no CPU path")
    if t.dtype != torch.float32:
        raise RuntimeError(f"{name} must be float32, got {t.dtype}")
    return t.contiguous()


def stirling_mean(counts):
    """mean([c>1] * (c log c - c + 0.5 log(2 pi c))): the flux-independent Stirling term of
    nn.PoissonNLLLoss(full=True) (jolideco/loss.py:35-37), computed once per dataset in float64."""
    c = np.asarray(counts, dtype=np.float64)
    term = np.zeros_like(c)
    m = c > 1
    term[m] = c[m] * np.log(c[m]) - c[m] + 0.5 * np.log(2 * np.pi * c[m])
    return float(term.mean())


CONV_MODES = {"auto": 0, "fft-exact": 1, "fft": 2, "direct": 3, "separable": 4}  # JD_CONV_MODE_* (jolideco_hip.h)
CONV_METHOD_NAMES = {0: "fft", 1: "direct", 2: "separable"}  # jd_conv_plan_method


def psf_separable_rank(psf, tol=0.0):
    """Number of outer products (1..3) that reproduce a host PSF array to a few fp32 ulps, 0 if it is not
    that low-rank (jd_psf_separable_rank).  A sampled Gaussian is 1."""
    import numpy as np

    psf = np.ascontiguousarray(psf, dtype=np.float32)
    if psf.ndim != 2:
        raise ValueError(f"psf must be two dimensional, got shape {psf.shape}")
    return int(_hip.lib().jd_psf_separable_rank(psf.ctypes.data, psf.shape[0], psf.shape[1], float(tol)))


def default_conv_method():
    """Convolution method used when none is requested: "auto" -- the separable kernels when the PSF is a sum of at most
    three outer products; else the MFMA direct kernel up to 17 taps (up to 33 on images below a megapixel or where the
    native FFT sizes do not fit); else the FFT path (native transforms where the sizes allow, rocFFT otherwise) -- unless
    JOLIDECO_CONV_METHOD overrides it.  "general" = "auto" without the low-rank test (every PSF as a general kernel: what
    an instrument PSF that is no short sum of outer products gets; `bench.py`'s `general_psf` run)."""
    import os

    method = os.environ.get("JOLIDECO_CONV_METHOD", "auto")
    if method not in CONV_MODES and method != "general":
        raise ValueError(f"JOLIDECO_CONV_METHOD={method!r}, must be one of {sorted(CONV_MODES) + ['general']}")
    return method


def _forget_operator(address):
    try:
        _hip.lib().jd_conv_operator_forget(address)
    except Exception:  # interpreter shutdown: the library may be gone
        pass


class ConvPlan:
    """'same'-convolution plan for one (H, W, kh, kw) geometry (jd_conv_plan).

    ``method``: "auto" | "fft" (rocFFT, fast padded grid) | "fft-exact" (rocFFT on the reference's
    (H+kh-1, W+kw-1) grid) | "direct" (MFMA Toeplitz kernel, PSFs up to 33x33) | "separable" (row + column
    pass for low-rank PSFs, `psf_separable_rank`; `psf_spectrum` raises for a PSF that is not)."""

    _cache = {}

    def __init__(self, H, W, kh, kw, device, exact_shape=False, method=None):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("ConvPlan needs a HIP device (no CPU fallback)")
        if method is None:
            method = "fft-exact" if exact_shape else default_conv_method()
        if method == "general":  # (the low-rank test is the caller's: a plan asked for by size alone is "auto")
            method = "auto"
        if method not in CONV_MODES:
            raise ValueError(f"unknown convolution method {method!r}, must be one of {sorted(CONV_MODES)}")
        handle = c_void_p()
        with torch.cuda.device(self.device):
            check(_hip.lib().jd_conv_plan_create(H, W, kh, kw, CONV_MODES[method], ctypes.byref(handle)))
        self._handle = handle
        shape = (c_int * 6)()
        check(_hip.lib().jd_conv_plan_shape(self._handle, shape))
        self.H, self.W, self.Hp, self.Wp, self.oy, self.ox = (int(v) for v in shape)
        self.kh, self.kw = kh, kw
        self.spectrum_size = int(_hip.lib().jd_conv_plan_spectrum_size(self._handle))
        self.method = CONV_METHOD_NAMES[_hip.lib().jd_conv_plan_method(self._handle)]

    @classmethod
    def get(cls, H, W, kh, kw, device, exact_shape=False, method=None):
        device = torch.device(device)
        if device.type == "cuda" and device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        if method is None:
            method = "fft-exact" if exact_shape else default_conv_method()
        if method == "general":
            method = "auto"
        key = (str(device), H, W, kh, kw, method)
        if key not in cls._cache:
            cls._cache[key] = cls(H, W, kh, kw, device, method=method)
        return cls._cache[key]

    @classmethod
    def clear_cache(cls):
        for plan in cls._cache.values():
            plan.close()
        cls._cache.clear()

    def close(self):
        if self._handle is not None:
            _hip.lib().jd_conv_plan_destroy(self._handle)
            self._handle = None

    @property
    def native_fft(self):
        """True when an "fft" plan runs on the hand-written transforms of csrc/fftnative.hip (no padded grid) instead of
        rocFFT (padded sizes 2^a * {1, 3, 9} up to 4608 columns x 2304 row pairs; any H, W since round 5; option JD_FFT_NATIVE=0 switches it off)."""
        return self.method == "fft" and (self.Hp, self.Wp) == (self.H, self.W)

    def takes_walk(self, n_datasets=1):
        """True when a launch over ``n_datasets`` datasets runs on the strip-walk kernels (jd_conv_plan_takes_walk)."""
        return bool(_hip.lib().jd_conv_plan_takes_walk(self._handle, int(n_datasets)))

    # --- kernel spectrum (once per dataset and component) -----------------------------------
    def psf_spectrum(self, psf, out=None):
        """The kernel operator of `psf` for this plan (jd_conv_psf_spectrum).  Operator buffers are immutable: to change
        the PSF of an existing buffer build the new operator INTO it (``out=``), which re-registers the address; an
        in-place copy of another operator's bytes is reported by the next library call."""
        psf = require_hip_tensor(psf, "psf")
        if tuple(psf.shape[-2:]) != (self.kh, self.kw):
            raise ValueError(f"psf shape {tuple(psf.shape)} does not match the plan ({self.kh}, {self.kw})")
        if out is not None:
            out = require_hip_tensor(out, "out")
            if out.numel() != 2 * self.spectrum_size or not out.is_contiguous():
                raise ValueError("out must be a contiguous operator buffer of this plan")
            check(_hip.lib().jd_conv_psf_spectrum(self._handle, ptr(psf), ptr(out), stream_ptr(psf.device)))
            return out
        khat = torch.empty(2 * self.spectrum_size, dtype=torch.float32, device=psf.device)
        check(_hip.lib().jd_conv_psf_spectrum(self._handle, ptr(psf), ptr(khat), stream_ptr(psf.device)))
        if self.method == "separable":
            # the library keeps what it knows about the operator (rank, support of its taps) by device address: the entry
            # goes with this tensor, so that a later allocation at the same address is an unknown buffer again (a view
            # that outlives the tensor is then unknown too: it takes the general kernels, never a wrong one)
            weakref.finalize(khat, _forget_operator, khat.data_ptr())
        return khat

    def walk_frame(self, khat):
        """17 / 33: the frame the strip-walk kernels run this operator in, 0: not theirs (jd_conv_operator_walk_frame)."""
        return int(_hip.lib().jd_conv_operator_walk_frame(self._handle, ptr(khat)))

    def _check_image(self, image, name):
        image = require_hip_tensor(image, name)
        if tuple(image.shape[-2:]) != (self.H, self.W) or image.numel() != self.H * self.W:
            raise ValueError(f"{name} shape {tuple(image.shape)} does not match the plan ({self.H}, {self.W})")
        return image

    def conv_same(self, image, scale, khat):
        image = self._check_image(image, "image")
        if scale is not None:
            scale = self._check_image(scale, "scale")
        out = torch.empty_like(image)
        check(_hip.lib().jd_conv_same(self._handle, ptr(image), ptr(scale), ptr(khat), ptr(out), stream_ptr(image.device)))
        return out

    def conv_same_adjoint(self, grad_out, scale, khat, grad_image=None, accumulate=False):
        grad_out = self._check_image(grad_out, "grad_out")
        if grad_image is None:
            grad_image = torch.empty_like(grad_out)
            accumulate = False
        check(
            _hip.lib().jd_conv_same_adjoint(
                self._handle, ptr(grad_out), ptr(scale), ptr(khat), ptr(grad_image), int(accumulate),
                stream_ptr(grad_out.device),
            )
        )
        return grad_image

    def npred_poisson_fwd_bwd(
        self, fluxes, exposures, khats, background, counts, stirling, loss_out, grads=None, accumulate=False,
        grad_scale=1.0, npred_out=None, eps=POISSON_EPS, upsampling=1, calibration=None,
    ):
        """Fused forward model + Poisson NLL (+ gradient) of one dataset; see include/jolideco_hip.h."""
        n = len(fluxes)
        if not (len(exposures) == len(khats) == n):
            raise ValueError("fluxes, exposures and khats must have the same length")
        for f in fluxes:
            self._check_image(f, "flux")
        if calibration is not None:
            # (shift_xy | None, log_background_norm, grad_shift_xy | None, grad_log_background_norm | None)
            shift, log_norm, grad_shift, grad_log_norm = calibration
            check(
                _hip.lib().jd_npred_poisson_calibrated_fwd_bwd(
                    self._handle, n, ptr_array(fluxes), ptr_array(exposures), ptr_array(khats), ptr(background),
                    ptr(counts), c_float(stirling), c_float(eps), ptr(loss_out),
                    ptr_array(grads) if grads is not None else None, int(accumulate), c_float(grad_scale),
                    ptr(npred_out), int(upsampling), ptr(shift), ptr(log_norm), ptr(grad_shift), ptr(grad_log_norm),
                    stream_ptr(background.device),
                )
            )
            return
        check(
            _hip.lib().jd_npred_poisson_fwd_bwd(
                self._handle, n, ptr_array(fluxes), ptr_array(exposures), ptr_array(khats), ptr(background),
                ptr(counts), c_float(stirling), c_float(eps), ptr(loss_out),
                ptr_array(grads) if grads is not None else None, int(accumulate), c_float(grad_scale),
                ptr(npred_out), int(upsampling), stream_ptr(background.device),
            )
        )

    MAX_BATCH = 16  # SEP_MAX_BATCH of csrc/kernels.h
    MAX_BATCH_COMPONENTS = 4  # SEP_BATCH_MAX_COMP

    def npred_poisson_batch_fwd_bwd(self, flux, exposures, khats, backgrounds, counts, stirlings, loss_outs, grad=None,
                                    accumulate=False, grad_scale=1.0, eps=POISSON_EPS):
        """All datasets of a joint step at once (separable plan shared by every dataset and component): one launch for
        the forward models + Poisson passes, one for the losses, one adjoint launch per flux component
        (jd_npred_poisson_batch_multi_fwd_bwd).  Same numbers as the per-dataset calls with ``accumulate`` from the
        second dataset on.

        One component: ``flux`` / ``grad`` are tensors and ``exposures`` / ``khats`` lists of per-dataset tensors.
        Several components: ``flux`` / ``grad`` are lists of tensors and ``exposures[d]`` / ``khats[d]`` lists with one
        tensor per component."""
        n = len(exposures)
        if not (len(khats) == len(backgrounds) == len(counts) == len(stirlings) == len(loss_outs) == n):
            raise ValueError("all per-dataset lists must have the same length")
        single = torch.is_tensor(flux)
        fluxes = [flux] if single else list(flux)
        grads = None if grad is None else ([grad] if single else list(grad))
        if single:
            exposures, khats = [[e] for e in exposures], [[k] for k in khats]
        n_comp = len(fluxes)
        if not 1 <= n_comp <= self.MAX_BATCH_COMPONENTS:
            raise ValueError(f"{n_comp} flux components: the batched joint step takes 1 to {self.MAX_BATCH_COMPONENTS}")
        if any(len(e) != n_comp for e in exposures) or any(len(k) != n_comp for k in khats):
            raise ValueError("every dataset needs one exposure and one operator per flux component")
        if grads is not None and len(grads) != n_comp:
            raise ValueError("one gradient image per flux component")
        for f in fluxes:
            self._check_image(f, "flux")
        for start in range(0, n, self.MAX_BATCH):
            sl = slice(start, min(n, start + self.MAX_BATCH))
            m = sl.stop - sl.start
            stirling_arr = (c_float * m)(*[float(v) for v in stirlings[sl]])
            check(
                _hip.lib().jd_npred_poisson_batch_multi_fwd_bwd(
                    self._handle, m, n_comp, ptr_array(fluxes), ptr_array([e for es in exposures[sl] for e in es]),
                    ptr_array([k for ks in khats[sl] for k in ks]), ptr_array(backgrounds[sl]), ptr_array(counts[sl]),
                    stirling_arr, c_float(eps), ptr_array(loss_outs[sl]), None if grads is None else ptr_array(grads),
                    int(accumulate or start > 0), c_float(grad_scale), stream_ptr(fluxes[0].device),
                )
            )

    def npred_poisson_calibrated_batch_fwd_bwd(self, flux, exposures, khats, backgrounds, counts, stirlings, loss_outs,
                                               calibrations, upsampling=1, grad=None, accumulate=False, grad_scale=1.0,
                                               eps=POISSON_EPS):
        """All datasets of a joint step of ONE flux component with per-dataset calibrations and / or up-sampling
        (jd_npred_poisson_calibrated_batch_fwd_bwd): the batched form of `npred_poisson_fwd_bwd(..., calibration=...)`
        per dataset with ``accumulate`` from the second dataset on -- same numbers.
        ``calibrations``: per dataset None or (shift_xy | None, log_background_norm | None, grad_shift_xy | None,
        grad_log_background_norm | None)."""
        n = len(exposures)
        if not (len(khats) == len(backgrounds) == len(counts) == len(stirlings) == len(loss_outs) == len(calibrations) == n):
            raise ValueError("all per-dataset lists must have the same length")
        self._check_image(flux, "flux")
        cals = [c if c is not None else (None, None, None, None) for c in calibrations]
        for start in range(0, n, self.MAX_BATCH):
            sl = slice(start, min(n, start + self.MAX_BATCH))
            m = sl.stop - sl.start
            stirling_arr = (c_float * m)(*[float(v) for v in stirlings[sl]])
            check(
                _hip.lib().jd_npred_poisson_calibrated_batch_fwd_bwd(
                    self._handle, m, ptr(flux), ptr_array(exposures[sl]), ptr_array(khats[sl]), ptr_array(backgrounds[sl]),
                    ptr_array(counts[sl]), stirling_arr, c_float(eps), ptr_array(loss_outs[sl]), ptr(grad),
                    int(accumulate or start > 0), c_float(grad_scale), int(upsampling),
                    ptr_array([c[0] for c in cals[sl]]), ptr_array([c[1] for c in cals[sl]]),
                    ptr_array([c[2] for c in cals[sl]]), ptr_array([c[3] for c in cals[sl]]), stream_ptr(flux.device),
                )
            )


class DeviceShifts:
    """Cycle-spin shifts of one prior evaluation held in DEVICE memory: ``dev`` = int32 tensor [shift_y mod H, shift_x mod W]
    the kernels read (jd_gmm_prior_fwd_bwd: device-resident step scalars), ``host`` = the same pair as the host drew it."""

    __slots__ = ("dev", "host")

    def __init__(self, dev, host):
        self.dev, self.host = dev, host


class GmmHandle:
    """GMM constants in MFMA fragment order on the device (jd_gmm)."""

    def __init__(self, precisions_cholesky, means_precisions_cholesky, log_det_cholesky, log_weights,
                 pixel_weights, device):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("GmmHandle needs a HIP device (no CPU fallback)")
        pc = np.ascontiguousarray(precisions_cholesky, dtype=np.float32)
        mp = np.ascontiguousarray(means_precisions_cholesky, dtype=np.float32)
        K, D, _ = pc.shape
        # const_k = -0.5 * D * log(2 pi) + log|P_k| + log pi_k   (patches/gmm.py:276-281)
        two_pi_log = np.float32(np.log(np.float32(2 * np.pi)))
        const_k = (
            np.float32(-0.5) * (np.float32(D) * two_pi_log)
            + np.asarray(log_det_cholesky, dtype=np.float32)
            + np.asarray(log_weights, dtype=np.float32)
        ).astype(np.float32)
        pw = np.ascontiguousarray(np.asarray(pixel_weights, dtype=np.float32).reshape(-1))
        self.K, self.D = K, D
        handle = c_void_p()
        as_fp = lambda a: a.ctypes.data_as(ctypes.POINTER(c_float))  # noqa: E731
        with torch.cuda.device(self.device):
            check(_hip.lib().jd_gmm_create(K, D, as_fp(pc), as_fp(mp), as_fp(const_k), as_fp(pw), ctypes.byref(handle)))
        self._handle = handle

    def close(self):
        if self._handle is not None:
            _hip.lib().jd_gmm_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def prior_fwd_bwd(self, flux, stride, shifts, value_out, value_scale, grad=None, grad_coef=0.0,
                      marginalize=False, patch_rows=(0, -1), accumulate_value=False, argmax_out=None, band_out=None, phases=3):
        """``band_out``: instead of accumulating into ``grad``, write the gradient of the patch rows ``patch_rows`` as the
        band of the rolled frame they cover (jd_gmm_prior_band_fwd_bwd; `band_rows` gives its extent)."""
        flux = require_hip_tensor(flux, "flux")
        H, W = flux.shape[-2:]
        if flux.numel() != H * W:
            raise ValueError("flux must be a single (H, W) image")
        shift_dev = None
        if isinstance(shifts, DeviceShifts):  # the kernels read the roll from device memory (captured graphs)
            if band_out is not None:
                raise ValueError("device-resident shifts are not available for the band form")
            shift_dev, shifts = ptr(shifts.dev), shifts.host
        sy, sx = (0, 0) if shifts is None else shifts
        if band_out is not None:
            if grad is not None or argmax_out is not None:
                raise ValueError("band_out excludes grad and argmax_out")
            y0, y1 = band_rows(patch_rows, stride, H)
            if band_out.numel() < (y1 - y0) * W:
                raise ValueError(f"band_out holds {band_out.numel()} values, rows [{y0}, {y1}) x {W} need {(y1 - y0) * W}")
            check(
                _hip.lib().jd_gmm_prior_band_fwd_bwd(
                    self._handle, ptr(flux), H, W, int(stride), int(sy), int(sx), int(patch_rows[0]), int(patch_rows[1]),
                    int(bool(marginalize)), c_float(value_scale), ptr(value_out), int(accumulate_value),
                    c_float(grad_coef), ptr(band_out), stream_ptr(flux.device),
                )
            )
            return
        check(
            _hip.lib().jd_gmm_prior_fwd_bwd(
                self._handle, ptr(flux), H, W, int(stride), int(sy), int(sx), int(patch_rows[0]), int(patch_rows[1]),
                int(bool(marginalize)), c_float(value_scale), ptr(value_out), int(accumulate_value),
                c_float(grad_coef), ptr(grad), ptr(argmax_out), shift_dev, int(phases), stream_ptr(flux.device),
            )
        )

    def prior_fwd_bwd_step(self, flux, stride, shifts, value_out, value_scale, grad_coef, step, marginalize=False,
                           accumulate_value=False, phases=3):
        """The whole prior with the component's optimizer step in the epilogue of its gather kernel
        (jd_gmm_prior_fwd_bwd_step; ``step``: a filled `_hip.Step`).  Raises RuntimeError where the library does not
        support it (stride < 4): the caller then evaluates the prior and steps separately."""
        flux = require_hip_tensor(flux, "flux")
        H, W = flux.shape[-2:]
        shift_dev = None
        if isinstance(shifts, DeviceShifts):
            shift_dev, shifts = ptr(shifts.dev), shifts.host
        sy, sx = (0, 0) if shifts is None else shifts
        check(
            _hip.lib().jd_gmm_prior_fwd_bwd_step(
                self._handle, ptr(flux), H, W, int(stride), int(sy), int(sx), int(bool(marginalize)), c_float(value_scale),
                ptr(value_out), int(accumulate_value), c_float(grad_coef), ctypes.byref(step), shift_dev, int(phases),
                stream_ptr(flux.device),
            )
        )

    def screen_clock(self):
        """(MHz, samples): the shader clock inside the screen kernel since the last call (jd_gmm_screen_clock; the first call
        arms the handle and returns (0.0, 0); synchronises the device)."""
        mhz, samples = ctypes.c_double(0.0), c_int(0)
        check(_hip.lib().jd_gmm_screen_clock(self._handle, ctypes.byref(mhz), ctypes.byref(samples)))
        return float(mhz.value), int(samples.value)

    def screen_stats(self):
        """(generation, fell_back, bucket_slots, patches, rows_per_patch) of the last screened pass that has finished
        (jd_gmm_screen_stats; no synchronisation)."""
        out = (c_int * 5)()
        check(_hip.lib().jd_gmm_screen_stats(self._handle, out))
        return tuple(int(v) for v in out)

    def estimate_log_prob(self, x):
        x = require_hip_tensor(x, "x")
        if x.ndim != 2 or x.shape[1] != self.D:
            raise ValueError(f"x must have shape (n, {self.D}), got {tuple(x.shape)}")
        out = torch.empty((x.shape[0], self.K), dtype=torch.float32, device=x.device)
        if x.shape[0]:
            check(_hip.lib().jd_gmm_estimate_log_prob(self._handle, ptr(x), x.shape[0], ptr(out), stream_ptr(x.device)))
        return out


# ------------------------------------------------------------------------------------------
# autograd seams
# ------------------------------------------------------------------------------------------
class ConvSameFunction(torch.autograd.Function):
    """out = conv_same(image * scale, psf) through rocFFT; backward = correlation with the PSF.
    Stands where the reference calls `convolve_fft_torch` (jolideco/utils/torch.py:347-370)."""

    @staticmethod
    def forward(ctx, image, scale, khat, plan):
        ctx.plan, ctx.scale, ctx.khat = plan, scale, khat
        ctx.shape = image.shape
        return plan.conv_same(image, scale, khat).reshape(image.shape)

    @staticmethod
    def backward(ctx, grad_out):
        grad = ctx.plan.conv_same_adjoint(grad_out.contiguous(), ctx.scale, ctx.khat)
        return grad.reshape(ctx.shape), None, None, None


class PoissonNLLFunction(torch.autograd.Function):
    """nn.PoissonNLLLoss(log_input=False, reduction='mean', eps=1e-25, full=True) (loss.py:35-37)."""

    @staticmethod
    def forward(ctx, npred, counts, stirling):
        npred = require_hip_tensor(npred, "npred")
        counts = require_hip_tensor(counts, "counts")
        loss = torch.empty(1, dtype=torch.float32, device=npred.device)
        grad = torch.empty_like(npred)
        check(
            _hip.lib().jd_poisson_nll(
                ptr(npred), ptr(counts), npred.numel(), c_float(stirling), c_float(POISSON_EPS), ptr(loss), ptr(grad),
                stream_ptr(npred.device),
            )
        )
        ctx.save_for_backward(grad)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, grad_loss):
        (grad,) = ctx.saved_tensors
        return grad * grad_loss, None, None


class GMMPatchPriorFunction(torch.autograd.Function):
    """Scalar GMM patch log-prior with its HIP gradient (priors/patches/core.py:227-246)."""

    @staticmethod
    def forward(ctx, flux, handle, stride, shifts, marginalize, value_scale):
        image = require_hip_tensor(flux, "flux")
        value = torch.empty(1, dtype=torch.float32, device=image.device)
        grad = None
        if flux.requires_grad:
            grad = torch.zeros(image.shape[-2:], dtype=torch.float32, device=image.device)
        handle.prior_fwd_bwd(
            image.reshape(image.shape[-2:]), stride, shifts, value, value_scale, grad=grad, grad_coef=value_scale,
            marginalize=marginalize,
        )
        ctx.grad, ctx.shape = grad, flux.shape
        return value.reshape(())

    @staticmethod
    def backward(ctx, grad_value):
        return (ctx.grad * grad_value).reshape(ctx.shape), None, None, None, None, None


class ElementwisePriorFunction(torch.autograd.Function):
    """InverseGammaPrior / ExponentialPrior value and gradient (priors/core.py:207-226,308-326)."""

    @staticmethod
    def forward(ctx, flux, kind, alpha, beta, log_const):
        image = require_hip_tensor(flux, "flux")
        value = torch.empty(1, dtype=torch.float32, device=image.device)
        grad = torch.zeros_like(image) if flux.requires_grad else None
        check(
            _hip.lib().jd_elementwise_prior_fwd_bwd(
                int(kind), ptr(image), image.numel(), c_float(alpha), c_float(beta), c_float(log_const), ptr(value),
                c_float(1.0 / image.numel()), ptr(grad), stream_ptr(image.device),
            )
        )
        ctx.grad, ctx.shape = grad, flux.shape
        return value.reshape(())

    @staticmethod
    def backward(ctx, grad_value):
        return (ctx.grad * grad_value).reshape(ctx.shape), None, None, None, None


def band_rows(patch_rows, stride, H, patch=8):
    """Pixel rows [y_begin, y_end) of the rolled frame that the patch rows ``patch_rows = (begin, end)`` cover
    (``end < 0``: up to the last patch row); an empty shard covers no row."""
    n_py = (H - patch) // stride + 1
    begin, end = patch_rows
    end = n_py if end < 0 else end
    if end <= begin:
        return begin * stride, begin * stride
    return begin * stride, (end - 1) * stride + patch


def add_rolled_bands(grad, shifts, bands, chunk, y_ranges):
    """grad (un-rolled image) += the bands of the rolled frame (jd_add_rolled_bands): ``bands`` is the flat device buffer
    of an all-gather, band b starts at ``b * chunk`` and holds the rows ``y_ranges[b] = (y_begin, y_end)``."""
    grad = require_hip_tensor(grad, "grad")
    bands = require_hip_tensor(bands, "bands")
    H, W = grad.shape[-2:]
    n = len(y_ranges)
    if bands.numel() < (n - 1) * chunk + max((y1 - y0) * W for y0, y1 in y_ranges):
        raise ValueError("bands buffer too small for n_bands pieces of `chunk` values")
    sy, sx = (0, 0) if shifts is None else shifts
    y0 = (c_int * n)(*[int(r[0]) for r in y_ranges])
    y1 = (c_int * n)(*[int(r[1]) for r in y_ranges])
    check(_hip.lib().jd_add_rolled_bands(ptr(grad), H, W, int(sy), int(sx), ptr(bands), int(chunk), n, y0, y1, stream_ptr(grad.device)))


def add_rolled_bands_step(shape, shifts, bands, chunk, y_ranges, step):
    """The sum of `add_rolled_bands` followed at once by the optimizer step of the image (jd_add_rolled_bands_step):
    ``step`` is the `_hip.Step` of the component, whose gradient image is only read.  Needs W % 4 == 0."""
    bands = require_hip_tensor(bands, "bands")
    H, W = (int(v) for v in shape[-2:])
    n = len(y_ranges)
    if bands.numel() < (n - 1) * chunk + max((y1 - y0) * W for y0, y1 in y_ranges):
        raise ValueError("bands buffer too small for n_bands pieces of `chunk` values")
    sy, sx = (0, 0) if shifts is None else shifts
    y0 = (c_int * n)(*[int(r[0]) for r in y_ranges])
    y1 = (c_int * n)(*[int(r[1]) for r in y_ranges])
    check(_hip.lib().jd_add_rolled_bands_step(H, W, int(sy), int(sx), ptr(bands), int(chunk), n, y0, y1, ctypes.byref(step),
                                              stream_ptr(bands.device)))


def sum_images(out, srcs):
    """out <- ((srcs[0] + srcs[1]) + ...) (jd_sum_images: the summed flux of components that share one forward operator)."""
    out = require_hip_tensor(out, "out")
    if not 1 <= len(srcs) <= 4:
        raise ValueError("1 to 4 source images")
    for t in srcs:
        require_hip_tensor(t, "source")
        if t.numel() != out.numel() or t.dtype != torch.float32 or not t.is_contiguous():
            raise ValueError("sources must be contiguous float32 images of the size of `out`")
    check(_hip.lib().jd_sum_images(ptr(out), ptr_array(list(srcs)), len(srcs), out.numel(), stream_ptr(out.device)))
    return out


def copy_image_to(src, dsts):
    """dsts[d] <- src for every d (jd_copy_image_to: one gradient image for all components that share an operator)."""
    src = require_hip_tensor(src, "src")
    if not 1 <= len(dsts) <= 4:
        raise ValueError("1 to 4 destination images")
    for t in dsts:
        require_hip_tensor(t, "destination")
        if t.numel() != src.numel() or t.dtype != torch.float32 or not t.is_contiguous():
            raise ValueError("destinations must be contiguous float32 images of the size of `src`")
    check(_hip.lib().jd_copy_image_to(ptr(src), ptr_array(list(dsts)), len(dsts), src.numel(), stream_ptr(src.device)))


def adam_bias_terms(step, lr, beta1, beta2):
    """step_size and sqrt(bias_correction2) exactly as torch.optim.Adam computes them
    (python floats = float64), torch/optim/adam.py `_single_tensor_adam`."""
    bias1 = 1 - beta1**step
    bias2 = 1 - beta2**step
    return lr / bias1, math.sqrt(bias2)
