"""Host-side (setup time, float64) helpers of the GMM patch prior.

Mirrors the public names of the reference's jolideco/utils/numpy.py so existing user code keeps
working; everything here runs once at construction, never in the optimisation loop.
"""
import numpy as np

__all__ = [
    "compute_precision_cholesky",
    "evaluate_trapez",
    "get_pixel_weights",
    "view_as_overlapping_patches",
    "reconstruct_from_overlapping_patches",
    "split_datasets_validation",
    "next_fast_len_2357",
]


def compute_precision_cholesky(covariances):
    """Per component ``P_k = (L_k^-1)^T`` with ``cov_k = L_k L_k^T`` (float64).

    Same contract as jolideco/utils/numpy.py:16-34; a failed factorisation raises ValueError.
    """
    covariances = np.asarray(covariances, dtype=np.float64)
    n_features = covariances.shape[1]
    out = np.empty_like(covariances)
    identity = np.eye(n_features)
    for idx, cov in enumerate(covariances):
        try:
            lower = np.linalg.cholesky(cov)
        except np.linalg.LinAlgError as exc:
            raise ValueError(f"Cholesky decomposition failed for {cov}") from exc
        from scipy.linalg import solve_triangular

        out[idx] = solve_triangular(lower, identity, lower=True).T
    return out


def evaluate_trapez(x, width, slope):
    """Unit-height trapezoid with a flat top of ``width`` and flanks of the given slope
    (jolideco/utils/numpy.py:37-51)."""
    x = np.asarray(x, dtype=float)
    top_lo, top_hi = min(-0.5 * width, 0.0), max(0.5 * width, 0.0)
    foot_lo, foot_hi = top_lo - 1.0 / slope, top_hi + 1.0 / slope
    out = np.zeros_like(x)
    rising = (x >= foot_lo) & (x < top_lo)
    flat = (x >= top_lo) & (x < top_hi)
    falling = (x >= top_hi) & (x < foot_hi)
    out[rising] = slope * (x[rising] - foot_lo)
    out[flat] = 1.0
    out[falling] = slope * (foot_hi - x[falling])
    return out


def get_pixel_weights(patch_shape, stride):
    """Separable trapezoid weights of an overlapping patch, normalised to sum ``stride**2``
    (jolideco/utils/numpy.py:54-79)."""
    width = int(np.max(patch_shape))
    overlap = width - stride
    centre = 0.5 * (width - 1.0)
    profile = evaluate_trapez(np.linspace(-centre, centre, width), width=stride - overlap, slope=1.0 / overlap)
    weights = np.outer(profile, profile)
    return weights * (stride**2 / weights.sum())


def view_as_overlapping_patches(image, shape, stride=None):
    """(n_patches, p*p) array of the overlapping windows of a 2-D numpy image, patch-row major
    (jolideco/utils/numpy.py:82-106, without the scikit-image dependency)."""
    if stride is None:
        stride = shape[0] // 2
    windows = np.lib.stride_tricks.sliding_window_view(image, shape)[::stride, ::stride]
    return windows.reshape(-1, shape[0] * shape[1])


def reconstruct_from_overlapping_patches(patches, image_shape, stride=None):
    """Weighted overlap-add of (n, p, p) patches into an image (jolideco/utils/numpy.py:109-148)."""
    p_h, p_w = patches.shape[1:]
    if stride is None:
        stride = p_w // 2
    weights = get_pixel_weights(patch_shape=(p_h, p_w), stride=stride)
    image = np.zeros(image_shape)
    idx = 0
    for top in range(0, image_shape[0] - p_h + 1, stride):
        for left in range(0, image_shape[1] - p_w + 1, stride):
            image[top : top + p_h, left : left + p_w] += weights * patches[idx]
            idx += 1
    return image


def split_datasets_validation(datasets, n_validation, random_state=None):
    """Random train/validation split of a datasets dict (jolideco/utils/numpy.py:151-181)."""
    if random_state is None:
        random_state = np.random.RandomState()
    names = list(datasets.keys())
    random_state.shuffle(names)
    return {
        "datasets": {name: datasets[name] for name in names[n_validation:]},
        "datasets_validation": {name: datasets[name] for name in names[:n_validation]},
    }


def next_fast_len_2357(n, multiple=1):
    """Smallest 2,3,5,7-smooth integer >= n that is a multiple of ``multiple`` (the padded FFT
    grid rule of csrc/fftconv.hip, mirrored on the host for tests and sizing)."""
    m = -(-n // multiple) * multiple
    while True:
        r = m
        for p in (2, 3, 5, 7):
            while r % p == 0:
                r //= p
        if r == 1:
            return m
        m += multiple
