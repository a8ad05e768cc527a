"""Synthetic scenes (numpy, host side).

Own generators for the benchmark configurations of BASELINE.json / SURVEY.md section 8(d) plus the
three toy datasets of the reference's jolideco/data/core.py (same names, arguments and return
dict), written without astropy: Gaussian / top-hat kernels are discretised at pixel centres (or
oversampled by 10 per axis) and normalised to unit sum.
"""
import numpy as np

__all__ = [
    "gaussian_kernel",
    "tophat_kernel",
    "convolve_same",
    "point_source_gauss_psf",
    "disk_source_gauss_psf",
    "gauss_and_point_sources_gauss_psf",
    "synthetic_gmm",
    "synthetic_observations",
    "instrument_like_psf",
    "instrument_observations",
]

BACKGROUND_LEVEL_DEFAULT = 2


def _grid(size, oversample=1):
    lo = -(size - 1) / 2 if size % 2 else -size / 2 + 0.5
    if oversample == 1:
        return lo + np.arange(size, dtype=float)
    sub = (np.arange(oversample) + 0.5) / oversample - 0.5
    return (lo + np.arange(size, dtype=float))[:, None] + sub[None, :]


def _kernel(func, shape, oversample):
    ny, nx = shape
    if oversample == 1:
        y, x = np.meshgrid(_grid(ny), _grid(nx), indexing="ij")
        values = func(x, y)
    else:
        ys, xs = _grid(ny, oversample).ravel(), _grid(nx, oversample).ravel()
        y, x = np.meshgrid(ys, xs, indexing="ij")
        values = func(x, y).reshape(ny, oversample, nx, oversample).mean(axis=(1, 3))
    return values / values.sum()


def gaussian_kernel(sigma, shape, oversample=1):
    """Unit-sum circular Gaussian on a (ny, nx) grid centred on the array."""
    return _kernel(lambda x, y: np.exp(-0.5 * (x * x + y * y) / sigma**2), shape, oversample)


def tophat_kernel(radius, shape, oversample=1):
    """Unit-sum disk of the given radius."""
    return _kernel(lambda x, y: (x * x + y * y <= radius**2).astype(float), shape, oversample)


def convolve_same(image, kernel):
    """Zero padded 'same' convolution (host, float64) used to draw the synthetic counts."""
    from scipy.signal import fftconvolve

    return fftconvolve(image, kernel, mode="same")


def _pack(counts, psf, exposure, background, flux, dtype):
    return {
        "counts": counts.astype(dtype),
        "psf": psf.astype(dtype),
        "exposure": exposure.astype(dtype),
        "background": background.astype(dtype),
        "flux": flux.astype(dtype),
    }


def point_source_gauss_psf(
    shape=(32, 32), shape_psf=(17, 17), sigma_psf=3, source_level=1000,
    background_level=BACKGROUND_LEVEL_DEFAULT, random_state=None, dtype=np.float32,
):
    """One point source in the centre, Gaussian PSF, flat exposure (jolideco/data/core.py:14-68)."""
    rs = random_state if random_state is not None else np.random.RandomState(None)
    background = background_level * np.ones(shape)
    exposure = np.ones(shape)
    flux = np.zeros(shape)
    flux[shape[0] // 2, shape[1] // 2] = source_level
    psf = gaussian_kernel(sigma_psf, (shape_psf[1], shape_psf[1]))
    counts = rs.poisson(background + convolve_same(flux * exposure, psf))
    return _pack(counts, psf, exposure, background, flux, dtype)


def disk_source_gauss_psf(
    shape=(32, 32), shape_psf=(17, 17), sigma_psf=3, source_level=1000, source_radius=3,
    background_level=BACKGROUND_LEVEL_DEFAULT, random_state=None, dtype=np.float32,
):
    """Disk source, exposure gradient of 50 % left to right (jolideco/data/core.py:71-131)."""
    rs = random_state if random_state is not None else np.random.RandomState(None)
    background = background_level * np.ones(shape)
    exposure = np.ones(shape) + 0.5 * np.linspace(-1, 1, shape[0])
    flux = source_level * tophat_kernel(source_radius, (shape[1], shape[1]), oversample=10)
    psf = gaussian_kernel(sigma_psf, (shape_psf[1], shape_psf[1]))
    counts = rs.poisson(background + convolve_same(flux * exposure, psf))
    return _pack(counts, psf, exposure, background, flux, dtype)


def gauss_and_point_sources_gauss_psf(
    shape=(32, 32), shape_psf=(17, 17), sigma_psf=2, source_level=1000, source_radius=2,
    background_level=BACKGROUND_LEVEL_DEFAULT, random_state=None, dtype=np.float32,
):
    """Gaussian blob plus four point sources, exposure gradient top to bottom
    (jolideco/data/core.py:134-201)."""
    rs = random_state if random_state is not None else np.random.RandomState(None)
    background = background_level * np.ones(shape)
    exposure = np.ones(shape) + 0.5 * np.linspace(-1, 1, shape[0]).reshape((-1, 1))
    flux = source_level * gaussian_kernel(source_radius, (shape[1], shape[1]), oversample=10)
    for fraction, idx_x, idx_y in zip([1, 0.3, 0.1, 0.03], [16, 16, 26, 6], [26, 6, 16, 16]):
        flux[idx_y, idx_x] = fraction * source_level
    psf = gaussian_kernel(sigma_psf, (shape_psf[1], shape_psf[1]))
    counts = rs.poisson(background + convolve_same(flux * exposure, psf))
    return _pack(counts, psf, exposure, background, flux, dtype)


def synthetic_gmm(n_components=128, n_features=64, seed=0):
    """Seeded synthetic SPD mixture: A ~ N(0, 1/D),
    cov_k = A A^T * U(0.01, 1) + 1e-3 I, zero means, weights ~ Dirichlet(1).
    Returns float64 (means, covariances, weights)."""
    rs = np.random.RandomState(seed)
    covs = np.empty((n_components, n_features, n_features))
    for k in range(n_components):
        a = rs.normal(size=(n_features, n_features)) / np.sqrt(n_features)
        covs[k] = a @ a.T * rs.uniform(0.01, 1.0) + 1e-3 * np.eye(n_features)
    weights = rs.dirichlet(np.ones(n_components))
    return np.zeros((n_components, n_features)), covs, weights


def image_like_gmm(n_components=128, patch=8, seed=0, ridge=1e-4):
    """Seeded mixture with the structure of a patch prior trained on images (the reference's trained mixtures are not
    in its tree): component k is a stationary random field seen through a `patch` x `patch`
    window, cov_k[i, j] = s_k * (rho(d_k(i, j)) + ridge * [i == j]) with rho(d) = exp(-d) (even k; spectrum ~ f^-3) or
    (1 + d) exp(-d) (odd k; ~ f^-5), an anisotropic distance d_k (correlation lengths 1-8 pixels along a random axis,
    0.5-1 x that across it), amplitudes s_k log-uniform over four decades, zero means, Dirichlet weights.  Power-law
    spectra: condition numbers 1e2-1e4, and smooth patches are nearly orthogonal to the high-precision directions --
    the hard case for a low-precision screen of the arg-max.
    Returns float64 (means, covariances, weights)."""
    rs = np.random.RandomState(seed)
    d = patch * patch
    yy, xx = np.mgrid[0:patch, 0:patch]
    dy = (yy.reshape(-1, 1) - yy.reshape(1, -1)).astype(np.float64)
    dx = (xx.reshape(-1, 1) - xx.reshape(1, -1)).astype(np.float64)
    covs = np.empty((n_components, d, d))
    for k in range(n_components):
        angle = rs.uniform(0, np.pi)
        length = rs.uniform(1.0, 8.0)
        across = length * rs.uniform(0.5, 1.0)
        u = (np.cos(angle) * dx + np.sin(angle) * dy) / length
        v = (-np.sin(angle) * dx + np.cos(angle) * dy) / across
        scale = 10.0 ** rs.uniform(-2, 2)
        dist = np.sqrt(u * u + v * v)
        rho = np.exp(-dist) if k % 2 == 0 else (1.0 + dist) * np.exp(-dist)
        covs[k] = scale * (rho + ridge * np.eye(d))
    weights = rs.dirichlet(np.ones(n_components))
    return np.zeros((n_components, d)), covs, weights


def psf_shape(sigma):
    """PSF array size of the synthetic observations: 17x17, 33x33 for sigma >= 3."""
    return (33, 33) if sigma >= 3 else (17, 17)


def instrument_like_psf(index, shape=(65, 65), dtype=np.float32):
    """A PSF that is NOT a short sum of outer products (what a simulated instrument PSF looks like; the reference's
    Chandra example draws 128x128 MARX simulations, examples/chandra-e0102-filament.py:91-93): an elliptical core
    rotated by an angle that changes with ``index``, an off-axis coma lobe and broad wings.  Unit sum."""
    ny, nx = shape
    y, x = np.meshgrid(_grid(ny), _grid(nx), indexing="ij")
    angle = 0.35 + 0.4 * index
    c, s_ = np.cos(angle), np.sin(angle)
    u, v = c * x + s_ * y, -s_ * x + c * y
    core = np.exp(-0.5 * ((u / (1.6 + 0.15 * index)) ** 2 + (v / (0.9 + 0.05 * index)) ** 2))
    lobe = 0.25 * np.exp(-0.5 * (((u - 3.5) / 2.5) ** 2 + ((v + 1.0) / 1.4) ** 2))
    wings = 0.05 / (1.0 + (x * x + y * y) / 36.0) ** 1.5
    psf = core + lobe + wings
    return (psf / psf.sum()).astype(dtype)


def instrument_observations(shape=(2048, 2048), n_obs=8, seed=0, psf_shape=(65, 65), dtype=np.float32):
    """`n_obs` observations shaped like the reference's Chandra example (examples/chandra-e0102-filament.py:91-93,
    178-203): general (not low-rank) `psf_shape` PSFs on the counts grid, meant to be fitted with
    ``upsampling_factor=2`` and one `NPredCalibration` per observation.  Same sky, exposures and backgrounds as
    `synthetic_observations`.  Returns (datasets, truth, flux_init, calibration values {name: (shift_x, shift_y, norm)})."""
    datasets, truth, flux_init = synthetic_observations(shape=shape, n_obs=n_obs, seed=seed, dtype=dtype)
    rs = np.random.RandomState(seed + 1)
    cal = {}
    for i, (name, d) in enumerate(datasets.items()):
        d["psf"] = instrument_like_psf(i, psf_shape, dtype)
        npred = d["background"] + np.clip(convolve_same(truth * d["exposure"], d["psf"]), 0, None)
        d["counts"] = rs.poisson(npred).astype(dtype)
        cal[name] = (0.15 * ((i % 3) - 1) + 0.05, -0.1 * ((i % 4) - 1.5), 1.0 + 0.02 * (i - n_obs / 2))
    return datasets, truth, flux_init, cal


def synthetic_observations(shape=(2048, 2048), n_obs=8, seed=0, n_points=64, dtype=np.float32):
    """`n_obs` observations of one sky (smooth blobs + point sources) with varying PSF width,
    exposure and background (BASELINE config 3):
    PSF sigma_i = 1.5 + 0.25 i on a 17x17 grid -- 33x33 for sigma >= 3 (observations 6 and 7 of 8) --,
    E_i = (1 + 0.1 i) * (1 +- 0.5 row gradient), bkg_i = 0.5 + 0.1 i.
    Returns (datasets dict, truth image, flux_init)."""
    rs = np.random.RandomState(seed)
    h, w = shape
    y, x = np.mgrid[0:h, 0:w].astype(np.float32)
    truth = np.full(shape, 1.0, dtype=np.float64)
    for _ in range(6):
        cy, cx = rs.uniform(0.2, 0.8) * h, rs.uniform(0.2, 0.8) * w
        sy, sx = rs.uniform(0.03, 0.12) * h, rs.uniform(0.03, 0.12) * w
        truth += rs.uniform(5, 40) * np.exp(-0.5 * (((y - cy) / sy) ** 2 + ((x - cx) / sx) ** 2))
    for _ in range(n_points):
        truth[rs.randint(0, h), rs.randint(0, w)] += rs.uniform(100, 1000)
    datasets = {}
    gradient = np.linspace(-1, 1, h).reshape(-1, 1)
    for i in range(n_obs):
        sigma = 1.5 + 0.25 * i
        psf = gaussian_kernel(sigma, psf_shape(sigma))
        sign = 1.0 if i % 2 == 0 else -1.0
        exposure = (1 + 0.1 * i) * (1 + sign * 0.5 * gradient) * np.ones(shape)
        background = (0.5 + 0.1 * i) * np.ones(shape)
        npred = background + np.clip(convolve_same(truth * exposure, psf), 0, None)
        counts = rs.poisson(npred)
        datasets[f"obs-{i}"] = {
            "counts": counts.astype(dtype),
            "psf": psf.astype(dtype),
            "exposure": exposure.astype(dtype),
            "background": background.astype(dtype),
        }
    flux_init = rs.gamma(30, size=shape)
    return datasets, truth.astype(dtype), flux_init
