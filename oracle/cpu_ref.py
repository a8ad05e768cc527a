"""CPU oracle: a PyTorch-CPU restatement of Jolideco's MAP inner loop.

TEST INFRASTRUCTURE -- NOT PRODUCT CODE.  Only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` may import this module; `jolideco_amd` never does.

Every function cites the reference file:line (relative to /root/reference/) that it restates.
It follows the reference op-for-op (same ATen ops, same per-component Python loop, autograd for
all gradients) so that it (a) reproduces the reference bit-for-bit on the same torch build and
(b) is a fair stand-in when timed as the "reference CPU path".

Parity pinning: checked in the build container against the imported reference
(oracle/refload/make_golden.py asserts np.array_equal(oracle, reference) on every case it writes) and,
on every run of the CPU suite (tests/test_oracle_golden.py), against the golden
fixtures in tests/golden/ (generated from the reference, incl. the reference's own known-answer
numbers from jolideco/tests/test_core.py:72-79,144-153,181-188).
"""
import contextlib
from dataclasses import dataclass, field
from math import log, pi

import numpy as np
import torch
import torch.nn.functional as F

TORCH_DEFAULT_GENERATOR_SEED = 67280421310721  # torch.Generator("cpu").initial_seed()

# Working precision.  float32 is the reference's (and the default: every cast below is then the one the reference
# makes).  `precision(np.float64)` re-runs the SAME restatement in double precision: the tests use it to measure how far
# the two fp32 paths (this oracle, the HIP kernels) are from exact arithmetic where their difference exceeds 1e-5.
_NP_DTYPE = np.float32


@contextlib.contextmanager
def precision(np_dtype):
    """Context manager: run the oracle in `np_dtype` (np.float32 | np.float64)."""
    global _NP_DTYPE
    previous_np, previous_torch = _NP_DTYPE, torch.get_default_dtype()
    _NP_DTYPE = np_dtype
    torch.set_default_dtype(torch.float64 if np_dtype == np.float64 else torch.float32)
    try:
        yield
    finally:
        _NP_DTYPE = previous_np
        torch.set_default_dtype(previous_torch)


def _tensor(array):
    """numpy -> tensor in the working precision (a no-op cast for float32 input at the default precision)."""
    return torch.from_numpy(np.ascontiguousarray(array, dtype=_NP_DTYPE))


# --------------------------------------------------------------------------------------
# FFT convolution + forward model
# --------------------------------------------------------------------------------------
def convolve_fft(image, kernel):
    """'same' linear convolution through rfft2/irfft2 on a (H+kh-1, W+kw-1) grid.

    Restates jolideco/utils/torch.py:347-370 (`convolve_fft_torch`) and the centre crop
    `_centered` at :337-344 (start = (full - new) // 2, i.e. ((kh-1)//2, (kw-1)//2)).
    image, kernel: (1, 1, H, W) / (1, 1, kh, kw) tensors.
    """
    h, w = image.shape[-2:]
    kh, kw = kernel.shape[-2:]
    full = (h + kh - 1, w + kw - 1)
    spec = torch.fft.rfft2(image, s=full) * torch.fft.rfft2(kernel, s=full)
    out = torch.fft.irfft2(spec, s=full)
    y0, x0 = (full[0] - h) // 2, (full[1] - w) // 2
    return out[..., y0 : y0 + h, x0 : x0 + w]


def edge_corrected_exposure(exposure, psf):
    """exposure / conv(ones, psf): jolideco/models/npred.py:108-113 (setup, once per dataset)."""
    return exposure / convolve_fft(torch.ones_like(exposure), psf)


def npred_component(flux, exposure, psf, upsampling_factor=None):
    """clip(sum_pool_u(conv_same(flux * exposure, psf)), 0, inf): jolideco/models/npred.py:160-191
    (rmf=None).  `exposure` / `psf` are the (already up-sampled) buffers of `upsample_setup`."""
    npred = convolve_fft(flux * exposure, psf)
    if upsampling_factor:
        npred = F.avg_pool2d(npred, kernel_size=upsampling_factor, divisor_override=1)  # npred.py:181-184
    return torch.clip(npred, 0, torch.inf)


def upsample_setup(tensor, upsampling_factor, is_psf):
    """Bilinear up-sampling of exposure / PSF at setup, PSF divided by u^2:
    jolideco/models/npred.py:96-106."""
    if upsampling_factor:
        tensor = F.interpolate(tensor, scale_factor=upsampling_factor, mode="bilinear")
        if is_psf:
            tensor = tensor / upsampling_factor**2
    return tensor


def shift_image(image, shift_xy, scale=1):
    """Sub-pixel shift through affine_grid / grid_sample (bilinear, zero padding, align_corners=False);
    the identity -- and no gradient to `shift_xy` -- when the shift is close to zero:
    jolideco/utils/torch.py:196-223 (`shift_image_torch`)."""
    if shift_xy is None or torch.all(torch.isclose(shift_xy, torch.zeros_like(shift_xy))):
        return image
    size = image.size()
    scale = 2 * scale / torch.tensor([[size[-1]], [size[-2]]])
    theta = torch.cat([torch.eye(2), scale * shift_xy.T], dim=1)[None]
    grid = F.affine_grid(theta=theta, size=size)
    return F.grid_sample(image, grid=grid)


def rescale_image(image, factor):
    """jolideco/utils/torch.py:172-193 (`rescale_image_torch`): identity for factor None / ~1."""
    if factor is None or torch.isclose(factor, torch.tensor(1.0)):
        return image
    theta = torch.cat([torch.eye(2) / factor, torch.tensor([[0], [0]])], dim=1)[None]
    grid = F.affine_grid(theta=theta, size=image.size())
    return F.grid_sample(image, grid=grid)


@dataclass
class CalibrationRef:
    """Parameters of jolideco/models/npred.py:298-402 (`NPredCalibration`)."""

    shift_xy: torch.Tensor  # (1, 2) [x, y], trainable
    log_background_norm: torch.Tensor  # (1,), trainable
    psf_scale: torch.Tensor  # (1,), requires_grad=False
    frozen: bool = False

    @classmethod
    def create(cls, shift_x=0.0, shift_y=0.0, background_norm=1.0, psf_scale=1.0, frozen=False):
        return cls(
            shift_xy=torch.tensor([[float(shift_x), float(shift_y)]], requires_grad=True),
            log_background_norm=torch.log(torch.tensor([float(background_norm)])).requires_grad_(True),
            psf_scale=torch.tensor([float(psf_scale)]),
            frozen=frozen,
        )

    def parameters(self):
        return [] if self.frozen else [self.shift_xy, self.log_background_norm]

    def to_dict(self):
        return {
            "shift_x": float(self.shift_xy[0, 0]), "shift_y": float(self.shift_xy[0, 1]),
            "background_norm": float(torch.exp(self.log_background_norm)), "psf_scale": float(self.psf_scale),
        }


def npred_total(fluxes, exposures, psfs, background, upsampling_factors=None):
    """sum_c npred_c + background, accumulated into zeros in component order, background last:
    jolideco/models/npred.py:210-261 (no calibration)."""
    total = torch.zeros(background.shape)
    ups = upsampling_factors or [None] * len(fluxes)
    for flux, exposure, psf, u in zip(fluxes, exposures, psfs, ups):
        total += npred_component(flux, exposure, psf, u)
    total += background
    return total


def poisson_nll(npred, counts):
    """nn.PoissonNLLLoss(log_input=False, reduction="mean", eps=1e-25, full=True):
    jolideco/loss.py:35-37.  mean(n - c*log(n+eps) + [c>1](c*log c - c + 0.5*log(2*pi*c)))."""
    return F.poisson_nll_loss(npred, counts, log_input=False, full=True, eps=1e-25, reduction="mean")


def poisson_nll_numpy(npred, counts):
    """Independent float64 numpy statement of the same formula (cross-check of the ATen op)."""
    n = np.asarray(npred, dtype=np.float64)
    c = np.asarray(counts, dtype=np.float64)
    value = n - c * np.log(n + 1e-25)
    stirling = np.zeros_like(c)
    m = c > 1
    stirling[m] = c[m] * np.log(c[m]) - c[m] + 0.5 * np.log(2 * np.pi * c[m])
    return float(np.mean(value + stirling))


# --------------------------------------------------------------------------------------
# GMM constants (setup)
# --------------------------------------------------------------------------------------
def trapezoid(x, width, slope):
    """jolideco/utils/numpy.py:37-51 (`evaluate_trapez`)."""
    x2, x3 = min(-width / 2.0, 0), max(width / 2.0, 0)
    x1, x4 = x2 - 1.0 / slope, x3 + 1.0 / slope
    conds = [(x >= x1) & (x < x2), (x >= x2) & (x < x3), (x >= x3) & (x < x4)]
    return np.select(conds, [slope * (x - x1), 1, slope * (x4 - x)])


def pixel_weights(patch_shape, stride):
    """jolideco/utils/numpy.py:54-79 (`get_pixel_weights`): outer product of a trapezoid,
    rescaled to sum stride**2."""
    width = int(np.max(patch_shape))
    overlap = width - stride
    half = (width - 1.0) / 2
    x = np.linspace(-half, half, width)
    values = trapezoid(x, width=(stride - overlap), slope=1.0 / overlap)
    weights = values * values[:, np.newaxis]
    return weights / weights.sum() * stride**2


def precision_cholesky(covariances):
    """jolideco/utils/numpy.py:16-34: P_k = (L_k^-1)^T with L_k the lower Cholesky factor
    (scipy, float64)."""
    from scipy import linalg

    out = np.empty(covariances.shape)
    eye = np.eye(covariances.shape[1])
    for k, cov in enumerate(covariances):
        chol = linalg.cholesky(cov, lower=True)
        out[k] = linalg.solve_triangular(chol, eye, lower=True).T
    return out


@dataclass
class GMM:
    """Constants of jolideco/priors/patches/gmm.py:64-299 (`GaussianMixtureModel`), fp32."""

    means: torch.Tensor  # (K, D)
    precisions_cholesky: torch.Tensor  # (K, D, D)
    weights: torch.Tensor  # (K,)
    stride: int = None  # meta.stride -> pixel weights (gmm.py:290-299)
    means_precisions_cholesky: torch.Tensor = field(init=False)
    log_det_cholesky: torch.Tensor = field(init=False)
    log_weights: torch.Tensor = field(init=False)
    pixel_weights: torch.Tensor = field(init=False)

    def __post_init__(self):
        k, d = self.means.shape
        # gmm.py:217-228
        self.means_precisions_cholesky = torch.stack(
            [torch.matmul(mu, pc) for mu, pc in zip(self.means, self.precisions_cholesky)]
        )
        # gmm.py:235-240
        diag = self.precisions_cholesky.reshape(k, -1)[:, :: d + 1]
        self.log_det_cholesky = torch.sum(torch.log(diag), axis=1)
        # gmm.py:114-117
        self.log_weights = torch.log(self.weights)
        # gmm.py:283-299
        p = int(d**0.5)
        if self.stride is None:
            w = np.ones((p, p))
        else:
            w = pixel_weights((p, p), self.stride)
        self.pixel_weights = _tensor(w.reshape((1, -1)))

    @classmethod
    def from_numpy(cls, means, covariances, weights, stride=None):
        """gmm.py:119-149: float64 scipy Cholesky, then cast to fp32."""
        pc = precision_cholesky(covariances)
        return cls(
            means=_tensor(means),
            precisions_cholesky=_tensor(pc),
            weights=_tensor(weights),
            stride=stride,
        )

    @property
    def patch_shape(self):
        p = int(self.means.shape[-1] ** 0.5)
        return p, p


def gmm_log_prob(x, gmm):
    """(Np, K) weighted log-probabilities with the reference's per-component Python loop:
    jolideco/priors/patches/gmm.py:262-281."""
    n, d = x.shape
    k = gmm.means.shape[0]
    q = torch.empty((n, k))
    for idx, (mu_prec, prec) in enumerate(zip(gmm.means_precisions_cholesky, gmm.precisions_cholesky)):
        y = torch.matmul(x, prec) - mu_prec
        q[:, idx] = torch.sum(torch.square(y) * gmm.pixel_weights, axis=1)
    two_pi = torch.tensor(2 * np.pi)
    return -0.5 * (d * torch.log(two_pi) + q) + gmm.log_det_cholesky + gmm.log_weights


# --------------------------------------------------------------------------------------
# Priors
# --------------------------------------------------------------------------------------
def draw_cycle_spin_shifts(generator, patch_shape):
    """Two host `randint` draws in [-p//4, p//4]; first -> rows (dim -2), second -> columns
    (dim -1): jolideco/utils/torch.py:108-119."""
    wy, wx = patch_shape[0] // 4, patch_shape[1] // 4
    first = torch.randint(-wy, wy + 1, (1,), generator=generator)
    second = torch.randint(-wx, wx + 1, (1,), generator=generator)
    return int(first), int(second)


def overlapping_patches(image, p, stride):
    """unfold rows, unfold columns, flatten to (Np, p*p): jolideco/utils/torch.py:226-275."""
    win = image.unfold(image.ndim - 2, p, stride).unfold(image.ndim - 1, p, stride)
    return torch.reshape(win, (-1, p * p))


def gmm_patch_log_like(flux, gmm, stride, shifts):
    """(Np, K) log-likelihood of all overlapping patches:
    jolideco/priors/patches/core.py:189-220 with IdentityImageNorm, cycle_spin shift `shifts`
    (None = cycle_spin off), no jitter, SubtractMeanPatchNorm (jolideco/utils/norms.py:97-103)."""
    image = flux
    if shifts is not None:
        image = torch.roll(image, shifts=shifts, dims=(image.ndim - 2, image.ndim - 1))
    p = gmm.patch_shape[0]
    patches = overlapping_patches(image, p, stride)
    keep = torch.all(patches > -1e5, dim=1, keepdims=False)
    patches = patches[keep, :]
    patches = patches - torch.nanmean(patches, dim=1, keepdims=True)
    return gmm_log_prob(patches, gmm)


def gmm_patch_log_prior(flux, gmm, stride, shifts, marginalize=False, return_argmax=False):
    """Scalar log-prior: jolideco/priors/patches/core.py:222-246."""
    loglike = gmm_patch_log_like(flux, gmm, stride, shifts)
    if marginalize:
        values = torch.logsumexp(loglike, dim=1)
        arg = None
    else:
        best = torch.max(loglike, dim=1)
        values, arg = best.values, best.indices
    p = gmm.patch_shape[0]
    scale = stride**2 / (p * p)
    out = torch.sum(values) * scale / flux.numel()
    return (out, arg) if return_argmax else out


class GMMPatchPriorRef:
    """Callable with the RNG behaviour of jolideco/priors/patches/core.py:30-246: one pair of
    draws from a CPU generator with torch's default seed per evaluation."""

    def __init__(self, gmm, stride=None, cycle_spin=True, marginalize=False, generator=None):
        self.gmm = gmm
        self.stride = gmm.stride if stride is None else stride
        self.cycle_spin = cycle_spin
        self.marginalize = marginalize
        self.generator = generator if generator is not None else torch.Generator(device="cpu")
        self.last_shifts = None

    def __call__(self, flux):
        shifts = None
        if self.cycle_spin:
            shifts = draw_cycle_spin_shifts(self.generator, self.gmm.patch_shape)
        self.last_shifts = shifts
        return gmm_patch_log_prior(flux, self.gmm, self.stride, shifts, self.marginalize)


class UniformPriorRef:
    """jolideco/priors/core.py:110-129."""

    def __call__(self, flux):
        return torch.tensor(0)


class InverseGammaPriorRef:
    """jolideco/priors/core.py:178-226 (cycle_spin_subpix=False)."""

    def __init__(self, alpha=10, beta=3 / 2):
        self.alpha = torch.Tensor([alpha])
        self.beta = torch.Tensor([beta])
        value = self.alpha * torch.log(self.beta)
        value -= torch.lgamma(self.alpha)
        self.log_constant_term = float(value)

    def __call__(self, flux):
        value = -self.beta / flux
        value += (-self.alpha - 1) * torch.log(flux)
        return torch.sum(value) / flux.numel() + self.log_constant_term


class ExponentialPriorRef:
    """jolideco/priors/core.py:282-326 (cycle_spin_subpix=False)."""

    def __init__(self, alpha=10):
        self.alpha = torch.Tensor([alpha])
        self.log_constant_term = torch.log(self.alpha)

    def __call__(self, flux):
        value = -self.alpha * flux
        return torch.sum(value) / flux.numel() + self.log_constant_term


# --------------------------------------------------------------------------------------
# Datasets / components / fit loops
# --------------------------------------------------------------------------------------
@dataclass
class DatasetRef:
    """Per-dataset tensors of jolideco/loss.py:79-124 + jolideco/models/npred.py:263-295."""

    counts: torch.Tensor  # (1,1,H,W)
    background: torch.Tensor
    exposures: list  # per component, edge-corrected (up-sampled when the component is)
    psfs: list  # per component (1,1,kh,kw) (up-sampled, / u^2)
    upsampling_factors: list = None  # per component: None or int
    calibration: CalibrationRef = None

    @classmethod
    def from_numpy(cls, dataset, component_names, upsampling_factors=None, calibration=None):
        exposures, psfs = [], []
        ups = upsampling_factors or [None] * len(component_names)
        for name, u in zip(component_names, ups):
            psf = dataset["psf"]
            if isinstance(psf, dict):
                psf = psf[name]
            psf_t = upsample_setup(_tensor(psf[np.newaxis, np.newaxis]), u, is_psf=True)
            exp_t = upsample_setup(_tensor(dataset["exposure"][np.newaxis, np.newaxis]), u, is_psf=False)
            exposures.append(edge_corrected_exposure(exp_t, psf_t))
            psfs.append(psf_t)
        return cls(
            counts=_tensor(dataset["counts"][np.newaxis, np.newaxis]),
            background=_tensor(dataset["background"][np.newaxis, np.newaxis]),
            exposures=exposures,
            psfs=psfs,
            upsampling_factors=list(ups),
            calibration=calibration,
        )

    def npred(self, fluxes):
        cal = self.calibration
        if cal is None:
            return npred_total(fluxes, self.exposures, self.psfs, self.background, self.upsampling_factors)
        # jolideco/models/npred.py:210-239: shift every flux, rescale the PSF, scale the background
        fluxes = [shift_image(f, cal.shift_xy, scale=u) for f, u in zip(fluxes, self.upsampling_factors)]
        psfs = [rescale_image(p, cal.psf_scale) for p in self.psfs]
        background = self.background * torch.exp(cal.log_background_norm)
        return npred_total(fluxes, self.exposures, psfs, background, self.upsampling_factors)

    def loss(self, fluxes):
        return poisson_nll(self.npred(fluxes), self.counts)


def log_flux_parameter(flux_init, upsampling_factor=None, use_log_flux=True):
    """theta = log(float32(flux)) (or the flux itself for use_log_flux=False) as a (1,1,H,W) leaf,
    bilinearly up-sampled first when the component is: jolideco/models/core.py:399-402,505-540."""
    flux = _tensor(flux_init[np.newaxis, np.newaxis])
    if upsampling_factor:
        flux = F.interpolate(flux, scale_factor=upsampling_factor, mode="bilinear")
    if use_log_flux:
        flux = torch.log(flux)
    return flux.requires_grad_(True)


def downsampled_flux(flux_upsampled, upsampling_factor=None):
    """SpatialFluxComponent.flux: sum-pool of the up-sampled flux, jolideco/models/core.py:596-607."""
    if upsampling_factor:
        return F.avg_pool2d(flux_upsampled, kernel_size=upsampling_factor, divisor_override=1)
    return flux_upsampled


def to_flux(theta, mask=None, use_log_flux=True):
    """exp(theta) [* mask]: jolideco/models/core.py:583-594.  With use_log_flux=False and no mask the
    PARAMETER ITSELF is returned, like the reference: the optimizer then updates the "flux" of the last
    step in place and the per-epoch trace sees the post-step values."""
    flux = torch.exp(theta) if use_log_flux else theta
    if mask is not None:
        flux = flux * mask
    return flux


def _trace_row(names_d, names_p, loss_datasets, loss_priors, beta, loss_validation=None):
    """Row layout and signs of jolideco/loss.py:212-250."""
    d_total = sum(loss_datasets)
    p_total = beta * sum(loss_priors)
    row = {"total": d_total - p_total, "datasets-total": d_total, "priors-total": -p_total}
    for name, value in zip(names_p, loss_priors):
        row[f"prior-{name}"] = -beta * value
    for name, value in zip(names_d, loss_datasets):
        row[f"dataset-{name}"] = value
    if loss_validation is not None:
        row["datasets-validation-total"] = sum(loss_validation)
    return row


def map_fit_sequential(
    datasets,
    flux_inits,
    priors,
    n_epochs,
    beta=1.0,
    learning_rate=0.1,
    datasets_validation=None,
    masks=None,
    record_steps=False,
    upsampling_factors=None,
    calibrations=None,
    use_log_flux=True,
):
    """The reference optimisation loop: one Adam step per dataset on
    L_d - beta * logprior / n_datasets, then a no-grad trace row evaluated on the STALE fluxes
    of the last step (jolideco/core.py:209-247, jolideco/loss.py:212-255).

    datasets: dict name -> dict(counts, psf, exposure, background) of numpy arrays
    flux_inits: dict component name -> (H, W) numpy array;  priors: dict name -> callable
    Returns (dict name -> final flux numpy, list of trace rows[, per-step records]).
    """
    names_c = list(flux_inits)
    names_d = list(datasets)
    ups = [(upsampling_factors or {}).get(n) for n in names_c]
    thetas = [log_flux_parameter(flux_inits[n], u, use_log_flux) for n, u in zip(names_c, ups)]
    masks = masks or {}
    mask_t = [
        None if masks.get(n) is None else torch.from_numpy(masks[n][np.newaxis, np.newaxis].astype(bool))
        for n in names_c
    ]
    calibrations = calibrations or {}
    data = [DatasetRef.from_numpy(datasets[n], names_c, ups, calibrations.get(n)) for n in names_d]
    data_val = None
    if datasets_validation:
        data_val = [DatasetRef.from_numpy(d, names_c, ups, calibrations.get(n)) for n, d in datasets_validation.items()]
    parameters = list(thetas)
    for cal in calibrations.values():  # jolideco/core.py:197-200
        parameters.extend(cal.parameters())
    optimizer = torch.optim.Adam(parameters, lr=learning_rate)
    prior_list = [priors[n] for n in names_c]
    n_datasets = len(data)
    trace, steps = [], []

    for _ in range(n_epochs):
        for d in data:
            optimizer.zero_grad()
            fluxes = tuple(to_flux(t, m, use_log_flux) for t, m in zip(thetas, mask_t))
            loss = d.loss(fluxes)
            loss_prior = sum(p(f) for f, p in zip(fluxes, prior_list))
            total = loss - beta * loss_prior / n_datasets
            total.backward()
            if record_steps:
                steps.append(
                    {
                        "loss": float(loss),
                        "loss_prior": float(loss_prior),
                        "grads": [t.grad.detach().clone().numpy()[0, 0] for t in thetas],
                    }
                )
            optimizer.step()
        with torch.no_grad():
            loss_datasets = [d.loss(fluxes).item() for d in data]
            loss_priors = [torch.as_tensor(p(f)).item() for f, p in zip(fluxes, prior_list)]
            loss_val = None
            if data_val is not None:
                loss_val = [d.loss(fluxes).item() for d in data_val]
        trace.append(_trace_row(names_d, names_c, loss_datasets, loss_priors, beta, loss_val))

    # `final` holds the UP-SAMPLED fluxes (== the fluxes when upsampling_factor is None)
    final = {n: to_flux(t, m, use_log_flux).detach().numpy()[0, 0] for n, t, m in zip(names_c, thetas, mask_t)}
    if record_steps:
        return final, trace, steps
    return final, trace


def joint_loss(data, fluxes, prior_list, beta):
    """Joint objective assembled from the reference's pieces:
    sum_d loss_function(npred_d, counts_d) - beta * prior_loss(fluxes)
    (= TotalLoss.__call__ of jolideco/loss.py:257-261 with the data term kept attached)."""
    loss_datasets = [d.loss(fluxes) for d in data]
    loss_priors = [p(f) for f, p in zip(fluxes, prior_list)]
    return sum(loss_datasets) - beta * sum(loss_priors), loss_datasets, loss_priors


def map_fit_joint(datasets, flux_inits, priors, n_epochs, beta=1.0, learning_rate=0.1, record_steps=False,
                  calibrations=None, upsampling_factors=None):
    """Joint mode harness: ONE Adam step per epoch on the summed objective.  The trace row of an
    epoch is made of the values of that step's forward pass (pre-step fluxes, no extra prior
    evaluation and therefore no extra RNG draw).  `calibrations` (name -> CalibrationRef): their parameters join
    the one optimizer, as in jolideco/core.py:197-204."""
    names_c = list(flux_inits)
    names_d = list(datasets)
    ups = [(upsampling_factors or {}).get(n) for n in names_c]
    thetas = [log_flux_parameter(flux_inits[n], u) for n, u in zip(names_c, ups)]
    calibrations = calibrations or {}
    data = [DatasetRef.from_numpy(datasets[n], names_c, ups if any(ups) or calibrations else None, calibrations.get(n))
            for n in names_d]
    parameters = list(thetas)
    for cal in calibrations.values():
        parameters.extend(cal.parameters())
    optimizer = torch.optim.Adam(parameters, lr=learning_rate)
    prior_list = [priors[n] for n in names_c]
    trace, steps = [], []
    for _ in range(n_epochs):
        optimizer.zero_grad()
        fluxes = tuple(to_flux(t) for t in thetas)
        total, loss_datasets, loss_priors = joint_loss(data, fluxes, prior_list, beta)
        total.backward()
        if record_steps:
            steps.append({"grads": [t.grad.detach().clone().numpy()[0, 0] for t in thetas]})
        optimizer.step()
        trace.append(
            _trace_row(
                names_d,
                names_c,
                [v.item() for v in loss_datasets],
                [torch.as_tensor(v).item() for v in loss_priors],
                beta,
            )
        )
    final = {n: to_flux(t).detach().numpy()[0, 0] for n, t in zip(names_c, thetas)}
    if record_steps:
        return final, trace, steps
    return final, trace


# --------------------------------------------------------------------------------------
# Single-evaluation helpers used by the parity tests
# --------------------------------------------------------------------------------------
def poisson_loss_and_grad(theta_np, dataset, component="flux"):
    """loss, npred and dL/dtheta for one dataset / one component at theta (autograd)."""
    theta = _tensor(theta_np[np.newaxis, np.newaxis]).requires_grad_(True)
    d = DatasetRef.from_numpy(dataset, [component])
    flux = to_flux(theta)
    npred = d.npred((flux,))
    loss = poisson_nll(npred, d.counts)
    loss.backward()
    return float(loss), npred.detach().numpy()[0, 0], theta.grad.numpy()[0, 0]


def gmm_prior_value_and_grad(flux_np, gmm, stride, shifts, marginalize=False):
    """log-prior, d logprior / d flux and the arg-max component per patch (autograd)."""
    flux = _tensor(flux_np[np.newaxis, np.newaxis]).requires_grad_(True)
    value, arg = gmm_patch_log_prior(flux, gmm, stride, shifts, marginalize, return_argmax=True)
    value.backward()
    arg_np = None if arg is None else arg.numpy().astype(np.int32)
    return float(value), flux.grad.numpy()[0, 0], arg_np


def synthetic_gmm(n_components, n_features=64, seed=0, zero_means=True, stride=4):
    """Seeded synthetic SPD mixture: A ~ N(0, 1/D),
    cov_k = A A^T * U(0.01, 1) + 1e-3 I, weights ~ Dirichlet(1)."""
    rs = np.random.RandomState(seed)
    covs = np.empty((n_components, n_features, n_features))
    for k in range(n_components):
        a = rs.normal(size=(n_features, n_features)) / np.sqrt(n_features)
        covs[k] = a @ a.T * rs.uniform(0.01, 1.0) + 1e-3 * np.eye(n_features)
    weights = rs.dirichlet(np.ones(n_components))
    if zero_means:
        means = np.zeros((n_components, n_features))
    else:
        means = 0.1 * rs.normal(size=(n_components, n_features))
    return means, covs, weights


LOG_2PI = log(2 * pi)
