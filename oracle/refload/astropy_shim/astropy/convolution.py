"""Gaussian2DKernel / Tophat2DKernel / convolve(_fft) with astropy>=5.2 semantics
(discretised at pixel centres or oversampled by `factor`, normalised to sum 1)."""
import numpy as np
from scipy import signal


def _round_up_to_odd(x):
    i = int(np.ceil(x))
    return i + 1 if i % 2 == 0 else i


def _axis(n):
    if n % 2 == 1:
        return np.arange(-(n - 1) // 2, (n - 1) // 2 + 1, dtype=float), -(n - 1) // 2, (n - 1) // 2 + 1
    return np.arange(-n // 2 + 0.5, n // 2 + 0.5, dtype=float), -n // 2 + 0.5, n // 2 + 0.5


def _oversample_axis(n, factor):
    _, lo, hi = _axis(n)
    return np.linspace(lo - 0.5 * (1 - 1 / factor), hi - 0.5 * (1 + 1 / factor), int((hi - lo) * factor))


class _Kernel2D:
    def __init__(self, func, default_size, x_size=None, y_size=None, mode="center", factor=10):
        x_size = default_size if x_size is None else int(x_size)
        y_size = x_size if y_size is None else int(y_size)
        if mode == "center":
            x, _, _ = _axis(x_size)
            y, _, _ = _axis(y_size)
            xx, yy = np.meshgrid(x, y)
            array = func(xx, yy)
        elif mode == "oversample":
            x = _oversample_axis(x_size, factor)
            y = _oversample_axis(y_size, factor)
            xx, yy = np.meshgrid(x, y)
            values = func(xx, yy)
            array = values.reshape(y_size, factor, x_size, factor).mean(axis=(1, 3))
        else:
            raise NotImplementedError(mode)
        self._array = array / array.sum()

    @property
    def array(self):
        return self._array

    @property
    def shape(self):
        return self._array.shape

    def __array__(self, dtype=None, copy=None):
        return self._array if dtype is None else self._array.astype(dtype)


class Gaussian2DKernel(_Kernel2D):
    def __init__(self, x_stddev, y_stddev=None, theta=0.0, **kwargs):
        sigma = float(x_stddev)
        amp = 1.0 / (2 * np.pi * sigma**2)
        super().__init__(
            lambda x, y: amp * np.exp(-0.5 * (x**2 + y**2) / sigma**2),
            _round_up_to_odd(8 * sigma),
            **kwargs,
        )


class Tophat2DKernel(_Kernel2D):
    def __init__(self, radius, **kwargs):
        r = float(radius)
        amp = 1.0 / (np.pi * r**2)
        super().__init__(
            lambda x, y: np.where(x**2 + y**2 <= r**2, amp, 0.0),
            _round_up_to_odd(2 * r),
            **kwargs,
        )


def _arr(k):
    return k.array if isinstance(k, _Kernel2D) else np.asarray(k)


def convolve(array, kernel, **kwargs):
    return signal.convolve2d(np.asarray(array, dtype=float), _arr(kernel), mode="same", boundary="fill", fillvalue=0)


def convolve_fft(array, kernel, **kwargs):
    k = _arr(kernel)
    return signal.fftconvolve(np.asarray(array, dtype=float), k / k.sum(), mode="same")
