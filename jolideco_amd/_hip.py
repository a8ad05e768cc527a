"""ctypes binding of libjolideco_hip.so (the C-ABI declared in include/jolideco_hip.h).

There is NO fallback: if the library is missing or a call fails, a RuntimeError is raised.
PyTorch is imported first so that the HIP runtime / rocFFT already loaded by torch are the ones
this library binds to (same sonames) and device pointers/streams are interchangeable.
"""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int32, c_longlong, c_size_t, c_void_p
from pathlib import Path

import torch  # noqa: F401  (must precede the CDLL so one HIP runtime is shared)

__all__ = ["lib", "library_path", "check", "ptr", "stream_ptr", "EXPORTS"]

LIB_NAME = "libjolideco_hip.so"


def library_path():
    """Path of the in-tree shared library (built by `make -C jolideco_amd/csrc`)."""
    override = os.environ.get("JOLIDECO_HIP_LIBRARY")
    return Path(override) if override else Path(__file__).resolve().parent / LIB_NAME


fp = POINTER(c_float)
fpp = POINTER(c_void_p)

# name -> (restype, argtypes); one entry per symbol declared in include/jolideco_hip.h
EXPORTS = {
    "jd_version": (c_int, []),
    "jd_last_error": (c_char_p, []),
    "jd_target_arch": (c_char_p, []),
    "jd_set_option": (c_int, [c_char_p, c_char_p]),
    "jd_get_option": (c_int, [c_char_p, POINTER(c_int), POINTER(c_int)]),
    "jd_conv_plan_create": (c_int, [c_int, c_int, c_int, c_int, c_int, POINTER(c_void_p)]),
    "jd_conv_plan_destroy": (c_int, [c_void_p]),
    "jd_conv_plan_shape": (c_int, [c_void_p, POINTER(c_int)]),
    "jd_conv_plan_spectrum_size": (c_size_t, [c_void_p]),
    "jd_conv_plan_method": (c_int, [c_void_p]),
    "jd_conv_native_fft_supported": (c_int, [c_int, c_int, c_int, c_int]),
    "jd_conv_plan_takes_walk": (c_int, [c_void_p, c_int]),
    "jd_conv_operator_walk_frame": (c_int, [c_void_p, c_void_p]),
    "jd_conv_operator_forget": (c_int, [c_void_p]),
    "jd_psf_separable_rank": (c_int, [c_void_p, c_int, c_int, c_float]),
    "jd_conv_psf_spectrum": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "jd_conv_same": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "jd_conv_same_adjoint": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "jd_npred_poisson_fwd_bwd": (
        c_int,
        [c_void_p, c_int, fpp, fpp, fpp, c_void_p, c_void_p, c_float, c_float, c_void_p, fpp, c_int, c_float,
         c_void_p, c_int, c_void_p],
    ),
    "jd_npred_poisson_calibrated_fwd_bwd": (
        c_int,
        [c_void_p, c_int, fpp, fpp, fpp, c_void_p, c_void_p, c_float, c_float, c_void_p, fpp, c_int, c_float,
         c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    ),
    "jd_npred_poisson_calibrated_batch_fwd_bwd": (
        c_int,
        [c_void_p, c_int, c_void_p, fpp, fpp, fpp, fpp, fp, c_float, fpp, c_void_p, c_int, c_float, c_int, fpp, fpp, fpp, fpp,
         c_void_p],
    ),
    "jd_npred_poisson_batch_fwd_bwd": (
        c_int,
        [c_void_p, c_int, c_void_p, fpp, fpp, fpp, fpp, fp, c_float, fpp, c_void_p, c_int, c_float, c_void_p],
    ),
    "jd_npred_poisson_batch_multi_fwd_bwd": (
        c_int,
        [c_void_p, c_int, c_int, fpp, fpp, fpp, fpp, fpp, fp, c_float, fpp, fpp, c_int, c_float, c_void_p],
    ),
    "jd_poisson_nll": (c_int, [c_void_p, c_void_p, c_size_t, c_float, c_float, c_void_p, c_void_p, c_void_p]),
    "jd_gmm_create": (c_int, [c_int, c_int, fp, fp, fp, fp, POINTER(c_void_p)]),
    "jd_gmm_destroy": (c_int, [c_void_p]),
    "jd_gmm_is_triangular": (c_int, [c_void_p]),
    "jd_gmm_prior_fwd_bwd": (
        c_int,
        [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p, c_int,
         c_float, c_void_p, c_void_p, c_void_p, c_int, c_void_p],
    ),
    "jd_gmm_prior_fwd_bwd_step": (
        c_int,
        [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p, c_int, c_float, c_void_p, c_void_p,
         c_int, c_void_p],
    ),
    "jd_gmm_screen_stats": (c_int, [c_void_p, POINTER(c_int)]),
    "jd_gmm_screen_clock": (c_int, [c_void_p, POINTER(c_double), POINTER(c_int)]),
    "jd_gmm_prior_band_fwd_bwd": (
        c_int,
        [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p, c_int,
         c_float, c_void_p, c_void_p],
    ),
    "jd_add_rolled_bands": (
        c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_size_t, c_int, POINTER(c_int), POINTER(c_int), c_void_p],
    ),
    "jd_add_rolled_bands_step": (
        c_int, [c_int, c_int, c_int, c_int, c_void_p, c_size_t, c_int, POINTER(c_int), POINTER(c_int), c_void_p, c_void_p],
    ),
    "jd_gmm_estimate_log_prob": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "jd_elementwise_prior_fwd_bwd": (
        c_int,
        [c_int, c_void_p, c_size_t, c_float, c_float, c_float, c_void_p, c_float, c_void_p, c_void_p],
    ),
    "jd_flux_from_theta": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_void_p]),
    "jd_sum_images": (c_int, [c_void_p, c_void_p, c_int, c_size_t, c_void_p]),
    "jd_copy_image_to": (c_int, [c_void_p, c_void_p, c_int, c_size_t, c_void_p]),
    "jd_step_scalars_fetch": (c_int, [c_void_p, c_void_p, c_int, c_void_p]),
    "jd_adam_step": (
        c_int,
        [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_float, c_float, c_float,
         c_float, c_float, c_float, c_float, c_int, c_int, c_void_p, c_void_p],
    ),
    "jd_adam_step_multi": (
        c_int,
        [c_int, fpp, fpp, fpp, fpp, POINTER(c_int), fp, fp, c_float, c_float, c_float, c_float, c_float, c_void_p, c_void_p],
    ),
    "jd_sgd_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_float, c_int, c_int, c_void_p]),
    "jd_profile_enable": (c_int, [c_int]),
    "jd_profile_disable": (c_int, []),
    "jd_profile_pause": (c_int, [c_int]),
    "jd_profile_read": (c_int, [c_int, POINTER(c_double), POINTER(c_longlong)]),
    "jd_kernel_name": (c_char_p, [c_int]),
    "jd_clock_probe": (c_int, [c_double, POINTER(c_double), c_void_p]),
}

class Step(ctypes.Structure):
    """`jd_step` of include/jolideco_hip.h: the optimizer step a prior evaluation applies in its epilogue."""

    _fields_ = [
        ("theta", c_void_p), ("flux_in", c_void_p), ("flux_out", c_void_p), ("grad_flux", c_void_p), ("exp_avg", c_void_p),
        ("exp_avg_sq", c_void_p), ("mask", c_void_p),
        ("step_size", c_float), ("beta1", c_float), ("beta2", c_float), ("one_minus_beta1", c_float),
        ("one_minus_beta2", c_float), ("bias2_sqrt", c_float), ("eps", c_float), ("lr", c_float),
        ("use_log_flux", c_int), ("sgd", c_int), ("bias_dev", c_void_p),
    ]


KERNEL_IDS = {
    "poisson_fused": 0, "gmm_fwd": 1, "gmm_bwd": 2, "gmm_gather": 3, "pad_mul": 4, "cmul": 5,
    "adjoint_epilogue": 6, "adam": 7, "fft_r2c": 8, "fft_c2r": 9, "direct_conv": 10, "sep_conv": 11,
    "gmm_screen": 12, "gmm_sort": 13, "gmm_exact": 14, "gmm_stage": 15, "shift": 16,
}

_lib = None


def lib():
    """Load (once) and return the shared library; raises RuntimeError when it is not built."""
    global _lib
    if _lib is None:
        path = library_path()
        if not path.exists():
            raise RuntimeError(
                f"{path} not found: the HIP extension is required (no CPU fallback). "
                "Build it with `make -C jolideco_amd/csrc` or `python -c 'import __graft_entry__ as g; g.build()'`."
            )
        handle = ctypes.CDLL(str(path))
        for name, (restype, argtypes) in EXPORTS.items():
            fn = getattr(handle, name)  # AttributeError -> missing export
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = handle
    return _lib


OPTION_GENERATION = 0  # bumped by every set_option: captured epochs (core.FitSession) are valid for one generation


def set_option(key, value=None):
    """Set a tuning / test switch of the library (`JD_*`, csrc/options.hip); None returns it to its default.  The library
    reads the environment only once, when it is loaded."""
    global OPTION_GENERATION
    check(lib().jd_set_option(key.encode(), None if value is None else str(value).encode()))
    OPTION_GENERATION += 1


def get_option(key):
    """Value of a switch (int) or None when it is unset."""
    is_set, value = c_int(0), c_int(0)
    check(lib().jd_get_option(key.encode(), ctypes.byref(is_set), ctypes.byref(value)))
    return value.value if is_set.value else None


class options:
    """Context manager: `with _hip.options(JD_GMM_SCREEN=0): ...` sets switches and restores the previous values."""

    def __init__(self, **switches):
        self.switches = switches
        self.previous = {}

    def __enter__(self):
        for key, value in self.switches.items():
            self.previous[key] = get_option(key)
            set_option(key, value)
        return self

    def __exit__(self, *exc):
        for key, value in self.previous.items():
            set_option(key, value)
        return False


def clock_probe(milliseconds=2.0, device=None):
    """Shader clock (MHz) the device holds under a vector-ALU load on every CU (jd_clock_probe; synchronises)."""
    mhz = c_double(0.0)
    check(lib().jd_clock_probe(float(milliseconds), ctypes.byref(mhz), stream_ptr(device)))
    return mhz.value


_PROFILE_ACTIVE = False


def profile_active():
    """True between `profile_enable` and `profile_read`: launches are bracketed by event pairs (no graph capture / replay)."""
    return _PROFILE_ACTIVE


def profile_enable(capacity=8192):
    """Start timing the library's kernels with hipEvent pairs on their launch stream."""
    global _PROFILE_ACTIVE
    check(lib().jd_profile_enable(int(capacity)))
    _PROFILE_ACTIVE = True


def profile_pause(paused=True):
    """Pause / resume the kernel timers (recorded pairs are kept)."""
    check(lib().jd_profile_pause(int(bool(paused))))


def profile_read():
    """{kernel: (total_ms, launches)} of the launches timed since `profile_enable`; synchronises."""
    out = {}
    for name, kid in KERNEL_IDS.items():
        total, count = c_double(0.0), c_longlong(0)
        status = lib().jd_profile_read(kid, ctypes.byref(total), ctypes.byref(count))
        if status != 0 and "JOLIDECO_HIP_LIBRARY" in os.environ:
            continue  # (an older library variant of an A/B run that does not know this timer)
        check(status)
        out[name] = (total.value, count.value)
    lib().jd_profile_disable()
    global _PROFILE_ACTIVE
    _PROFILE_ACTIVE = False
    return out


def check(status):
    """Raise RuntimeError with the library's message for a non-zero status."""
    if status != 0:
        msg = lib().jd_last_error()
        raise RuntimeError(f"libjolideco_hip error {status}: {msg.decode() if msg else '?'}")


def ptr(tensor):
    """Device pointer of a contiguous fp32/int32 CUDA(HIP) tensor, or None."""
    if tensor is None:
        return None
    if not tensor.is_cuda:
        raise RuntimeError("libjolideco_hip needs tensors on a HIP device (no CPU fallback)")
    if not tensor.is_contiguous():
        raise RuntimeError("libjolideco_hip needs contiguous tensors")
    return c_void_p(tensor.data_ptr())


def ptr_array(tensors):
    """Host array of device pointers (for the `const float* const*` arguments)."""
    arr = (c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        p = ptr(t)
        arr[i] = p.value if p is not None else None
    return arr


def stream_ptr(device=None):
    """Current torch HIP stream as a void*."""
    return c_void_p(torch.cuda.current_stream(device).cuda_stream)
