"""Predicted-counts models (reference: jolideco/models/npred.py:31-295).

`NPredModel` holds the edge-corrected exposure and the cached kernel spectrum K-hat of one
(dataset, component) pair on the HIP device; `NPredModels` adds the background.  `forward` /
`evaluate` are differentiable through the HIP convolution (`ops.ConvSameFunction`); the fit loop
uses the fused `fwd_bwd` instead (one C-ABI call per dataset).
"""
import os

import numpy as np
import torch
import torch.nn as nn

from .. import _hip
from ..ops import ConvPlan, ConvSameFunction, default_conv_method, psf_separable_rank

SEPARABLE_MAX_EDGE = 68  # SEP_MAX_K of csrc/kernels.h
DIRECT_FAST_EDGE = 17    # widest PSF the split-fp16 direct kernel covers with one 32-column step per PSF row

__all__ = ["NPredModel", "NPredModels", "NPredCalibration", "NPredCalibrations"]


def _to_device_image(array, device):
    return torch.from_numpy(np.ascontiguousarray(array, dtype=np.float32)).to(device)


def embed_kernel(psf, shape):
    """Zero-embed ``psf`` (kh, kw) in an array of ``shape`` (KH, KW) at the offset that keeps the
    centre crop of the 'same' convolution identical: dy = (KH-1)//2 - (kh-1)//2."""
    kh, kw = psf.shape
    KH, KW = shape
    if KH < kh or KW < kw:
        raise ValueError(f"cannot embed a {psf.shape} kernel into {shape}")
    dy, dx = (KH - 1) // 2 - (kh - 1) // 2, (KW - 1) // 2 - (kw - 1) // 2
    out = np.zeros(shape, dtype=psf.dtype)
    out[dy : dy + kh, dx : dx + kw] = psf
    return out


class NPredModel(nn.Module):
    """Forward model of one component: clip(PSF (*) (flux * exposure), 0, inf)."""

    def __init__(self, exposure, psf, plan, khat, upsampling_factor=None):
        super().__init__()
        self.register_buffer("exposure", exposure)  # (1, 1, H, W), edge corrected
        self.register_buffer("psf", psf)  # (1, 1, kh, kw)
        self.plan = plan
        self.khat = khat
        self.upsampling_factor = upsampling_factor

    @property
    def shape_upsampled(self):
        return tuple(self.exposure.shape)

    @property
    def shape(self):
        shape = list(self.shape_upsampled)
        if self.upsampling_factor:
            shape[-1] //= self.upsampling_factor
            shape[-2] //= self.upsampling_factor
        return tuple(shape)

    @classmethod
    def from_numpy(cls, exposure, psf, upsampling_factor=None, correct_exposure_edges=True, device="cuda",
                   kernel_shape=None, psf_scale=None, allow_separable=True):
        """Upload one dataset's exposure and PSF, compute K-hat once and apply the reference's
        edge correction exposure / conv(1, psf) (models/npred.py:66-115) on the device.

        ``kernel_shape`` (KH, KW) >= psf.shape embeds the PSF in a larger zero array such that the
        'same' crop is unchanged; it lets all components of a dataset share one FFT plan.

        With the convolution method "auto" a PSF that is a sum of at most three outer products (every sampled
        Gaussian is one) takes the separable kernel (csrc/sepconv.hip) unless ``allow_separable`` is False."""
        device = torch.device(device)
        psf = np.ascontiguousarray(psf, dtype=np.float32)
        exposure = np.ascontiguousarray(exposure, dtype=np.float32)
        u = upsampling_factor or 1
        if upsampling_factor:
            # setup-time bilinear up-sampling on the host, PSF divided by u^2 (models/npred.py:96-106) -- of the PSF
            # as given: F.interpolate clamps at the array edge, so a PSF embedded in zeros first would get a different
            # up-sampled border than the reference's
            import torch.nn.functional as F

            up = lambda a: F.interpolate(torch.from_numpy(a)[None, None], scale_factor=upsampling_factor, mode="bilinear")[0, 0]  # noqa: E731
            exposure = up(exposure).numpy()
            psf = (up(psf) / upsampling_factor**2).numpy()
        # the calibration's PSF scale acts on the PSF of the dataset, before it is embedded for the shared plan
        rescaled = rescale_psf(psf, psf_scale) if psf_scale is not None else psf
        if kernel_shape is not None:
            target = (kernel_shape[0] * u, kernel_shape[1] * u)
            if target != psf.shape:
                same = rescaled is psf
                psf = embed_kernel(psf, target)
                rescaled = psf if same else embed_kernel(rescaled, target)
        exposure_t = _to_device_image(exposure, device)
        psf_t = _to_device_image(psf, device)
        H, W = exposure_t.shape
        kh, kw = psf_t.shape
        method = default_conv_method()
        if method == "general":  # every PSF as a general kernel
            method, allow_separable = "auto", False
        if method == "auto" and allow_separable and max(kh, kw) <= SEPARABLE_MAX_EDGE:
            rank = psf_separable_rank(psf)
            if rank and rescaled is not psf:
                rank = max(rank, psf_separable_rank(rescaled)) if psf_separable_rank(rescaled) else 0
            # rank 1: always (18 against 27 us per 2048^2 convolution at 17 x 17, and the joint step batches it).  Rank 2
            # and 3 cost 25 / 31 us there and switch off the LDS aliasing of every rank-1 launch of the process, against
            # 26 us for the split-fp16 direct kernel -- which grows with the PSF area, the separable passes with its
            # edge: beyond 17 taps (two 32-column steps per PSF row) the separable kernel wins again
            # (tools/rank_bench.py)
            if rank == 1 or (rank > 1 and max(kh, kw) > DIRECT_FAST_EDGE):
                method = "separable"
        plan = ConvPlan.get(H, W, kh, kw, device, method=method)
        khat = plan.psf_spectrum(psf_t)
        if correct_exposure_edges:
            # the edge correction uses the UN-rescaled PSF (models/npred.py:108-113 runs before any calibration)
            weights = plan.conv_same(torch.ones_like(exposure_t), None, khat)
            exposure_t = exposure_t / weights
        if rescaled is not psf:
            psf_t = _to_device_image(rescaled, device)
            khat = plan.psf_spectrum(psf_t)
        return cls(
            exposure=exposure_t[None, None], psf=psf_t[None, None], plan=plan, khat=khat,
            upsampling_factor=upsampling_factor,
        )

    @classmethod
    def from_dataset_numpy(cls, dataset, upsampling_factor=None, correct_exposure_edges=True, device="cuda"):
        return cls.from_numpy(
            exposure=dataset["exposure"], psf=dataset["psf"], upsampling_factor=upsampling_factor,
            correct_exposure_edges=correct_exposure_edges, device=device,
        )

    def forward(self, flux, psf_scale=None):
        if psf_scale is not None:
            raise NotImplementedError("psf_scale calibration is not implemented in jolideco_amd")
        conv = ConvSameFunction.apply(flux, self.exposure, self.khat, self.plan)
        if self.upsampling_factor:
            import torch.nn.functional as F

            conv = F.avg_pool2d(conv, kernel_size=self.upsampling_factor, divisor_override=1)
        return torch.clip(conv, 0, torch.inf)


def dataset_psfs(dataset, components):
    """{component name: PSF array} of one dataset (`psf`: one array, or a dict by component, models/npred.py:281-295)."""
    psf = dataset["psf"]
    return {name: np.asarray(psf[name] if isinstance(psf, dict) else psf) for name in components.keys()}


WALK_MAX_EDGE = 33  # widest frame of the strip-walk kernels (csrc/walkconv.hip)

# Estimated cost of a likelihood step (forward model + Poisson pass + adjoint) per (dataset, component), in units of one
# rank-1 PSF of up to 17 taps on the strip-walk kernels, from the per-kernel tables of profiles/r04 at 2048^2: 17-tap frame
# 15 + 10.5 us; 33-tap frame 1.6 x the forward and 2.4 x the adjoint; MFMA Toeplitz 2 x 27 us at 17
# taps, growing with the PSF area; native FFT 72 us per observation whatever the PSF size; separable tile kernel about 1.3
# per rank.  Only the RATIOS between the datasets of one fit matter (`distributed.lpt_assignment`), and the ratio to the
# prior below (`balanced_shares`).
COST_WALK17, COST_WALK33, COST_TILE_PER_RANK, COST_DIRECT17, COST_FFT = 1.0, 1.9, 1.3, 2.1, 2.8
# GMM patch prior of the whole image in the same units: screen 198 us (proportional to the number of components, K = 128)
# + stage, sort, exact evaluation, arg-max, gather 170 us against the 25.5 us unit
COST_PRIOR_FIXED, COST_PRIOR_PER_128_COMPONENTS = 6.7, 7.8


def estimate_dataset_cost(dataset, components, calibrated=False):
    """Estimated cost of one dataset's likelihood step in the units above: decides which rank owns the dataset and how
    many patch rows of the prior that rank evaluates (jolideco_amd/core.py, sharded joint fits).  Host logic only: the
    convolution method "auto" would choose, from the PSF shapes and their separable ranks."""
    method = default_conv_method()
    counts_shape = np.shape(dataset["counts"])
    total, units = 0.0, []
    for name, psf in dataset_psfs(dataset, components).items():
        edge = max(psf.shape[-2:])
        up = components[name].upsampling_factor or 1
        edge *= up
        if method == "fft" or calibrated or up > 1:
            unit = COST_FFT if edge > DIRECT_FAST_EDGE or method == "fft" else COST_DIRECT17
        else:
            rank = psf_separable_rank(psf) if method == "auto" and psf.ndim == 2 else 0
            if rank == 1 and edge <= WALK_MAX_EDGE:
                unit = COST_WALK17 if edge <= 17 else COST_WALK33
            elif rank >= 1 and (rank == 1 or edge > DIRECT_FAST_EDGE):
                unit = COST_TILE_PER_RANK * rank * max(edge / 17.0, 1.0)
            elif edge <= DIRECT_FAST_EDGE or counts_shape[0] * counts_shape[1] < 1 << 20 and edge <= 33:
                unit = COST_DIRECT17 * (edge / 17.0) ** 2
            else:
                unit = COST_FFT
        total += unit * up * up  # (the flux grid has up^2 pixels per counts pixel)
        units.append(unit * up * up)
    # components that share the operator are evaluated as their sum: ONE forward model and ONE adjoint per dataset
    # (`NPredModels.shared_operator`; the host-side conditions here: one non-negative rank-1 PSF for all components, no
    # calibration, no up-sampling, the separable path)
    psfs = list(dataset_psfs(dataset, components).values())
    if (len(psfs) > 1 and not calibrated and method == "auto" and os.environ.get("JOLIDECO_MERGE_COMPONENTS", "1") != "0"
            and all((c.upsampling_factor or 1) == 1 and bool(getattr(c, "use_log_flux", True)) for c in components.values())
            and all(p.shape == psfs[0].shape and np.array_equal(p, psfs[0]) for p in psfs[1:])
            and psfs[0].ndim == 2 and float(np.min(psfs[0])) >= 0.0 and max(psfs[0].shape) <= WALK_MAX_EDGE
            and psf_separable_rank(psfs[0]) == 1):
        return max(units)
    return total


def common_kernel_shape(datasets, components, calibrations=None):
    """PSF array shape to embed every PSF of every dataset in, or None to leave each dataset its own.

    Datasets whose PSF arrays differ in size have different plans, and the batched joint step needs one.  When that is
    all that keeps them apart -- convolution method "auto", every PSF a single outer product (every sampled Gaussian
    is), no calibration, no up-sampling, at most 4 components, nothing wider than 33 taps -- the PSFs are embedded in
    zeros up to the largest array shape: the same 'same' convolution (`embed_kernel`), one separable plan, and the
    kernels work on the non-zero taps of each operator, not on the array size (include/jolideco_hip.h,
    jd_conv_operator_walk_frame).  The same for one flux component whose PSFs all take the FFT path: its batched joint
    step needs one plan too, and an FFT convolution costs the same for every PSF size."""
    method = default_conv_method()
    if method not in ("auto", "general", "fft") or calibrations or len(datasets) < 2 or not 1 <= len(components) <= 4:
        return None
    if any((c.upsampling_factor or 1) != 1 for c in components.values()):
        return None
    psfs = [p for dataset in datasets.values() for p in dataset_psfs(dataset, components).values()]
    shapes = {p.shape for p in psfs}
    if len(shapes) < 2:
        return None
    shape = (max(s[0] for s in shapes), max(s[1] for s in shapes))
    if method == "auto" and max(shape) <= WALK_MAX_EDGE and all(p.ndim == 2 and psf_separable_rank(p) == 1 for p in psfs):
        return shape
    # FFT path (one flux component: its batched joint step): the cost of the convolution does not depend on the PSF size,
    # so PSFs of different sizes share the plan of the largest -- where every one of them takes the FFT path anyway (the
    # method is "fft", or none is low-rank and all are wider than the direct kernel's 17 taps)
    if len(components) == 1 and all(p.ndim == 2 for p in psfs):
        if method == "fft":
            return shape
        general = method == "general" or all(psf_separable_rank(p) == 0 for p in psfs)
        counts_shape = np.shape(next(iter(datasets.values()))["counts"])
        # ("auto" sends a general PSF beyond 17 taps to the FFT path only where the native transforms cover the size --
        # elsewhere it takes the Toeplitz kernel up to 33 taps, whose cost grows with the array size: no embedding then)
        native = bool(_hip.lib().jd_conv_native_fft_supported(int(counts_shape[0]), int(counts_shape[1]), shape[0], shape[1]))
        if (general and native and all(max(p.shape) > DIRECT_FAST_EDGE for p in psfs)
                and counts_shape[0] * counts_shape[1] >= 1 << 20):
            return shape
    return None


def rescale_psf(psf, factor):
    """Setup-time `rescale_image_torch` (jolideco/utils/torch.py:172-193) of a (kh, kw) PSF on the host:
    the PSF scale of a calibration is not trainable (npred.py:333-334), so the rescaled PSF is a
    constant of the fit."""
    import torch.nn.functional as F

    factor = torch.as_tensor(float(factor))
    if torch.isclose(factor, torch.tensor(1.0)):
        return psf
    image = torch.from_numpy(np.ascontiguousarray(psf, dtype=np.float32))[None, None]
    theta = torch.cat([torch.eye(2) / factor, torch.tensor([[0.0], [0.0]])], dim=1)[None]
    grid = F.affine_grid(theta=theta, size=image.size(), align_corners=False)
    return F.grid_sample(image, grid=grid, align_corners=False)[0, 0].numpy()


class NPredCalibration(nn.Module):
    """Dataset calibration parameters (reference: jolideco/models/npred.py:298-402).

    Attributes
    ----------
    shift_xy : `~torch.nn.Parameter`  (1, 2) shift in x / y direction in counts pixels, trainable
    background_norm : background normalisation, trainable as its logarithm
    psf_scale : PSF scale (not trainable, applied once at setup)
    frozen : bool, exclude the calibration from the optimisation (it is still applied)
    weight : likelihood weight (stored; the fit loop of the reference does not use it either)
    """

    def __init__(self, shift_x=0.0, shift_y=0.0, background_norm=1.0, psf_scale=1.0, frozen=False, weight=1.0):
        super().__init__()
        self.shift_xy = nn.Parameter(torch.tensor([[shift_x, shift_y]], dtype=torch.float32))
        self._background_norm = nn.Parameter(torch.log(torch.tensor([background_norm], dtype=torch.float32)))
        self.psf_scale = nn.Parameter(torch.tensor([psf_scale], dtype=torch.float32), requires_grad=False)
        self.frozen = frozen
        self.weight = weight

    @property
    def background_norm(self):
        return torch.exp(self._background_norm)

    @property
    def shift_is_active(self):
        """The reference applies the shift -- and propagates a gradient to it -- only while it is not
        close to zero (`shift_image_torch`, utils/torch.py:211): a shift initialised at exactly 0 never
        moves.  Evaluated once at setup from the host copy of the initial value."""
        shift = self.shift_xy.detach().cpu()
        return not bool(torch.all(torch.isclose(shift, torch.zeros_like(shift))))

    def parameters(self, recurse=True):
        return [] if self.frozen else super().parameters(recurse)

    def to_dict(self):
        shift_xy = self.shift_xy.detach().cpu().numpy()
        return {
            "shift_x": shift_xy[0, 0].item(),
            "shift_y": shift_xy[0, 1].item(),
            "background_norm": self.background_norm.detach().cpu().numpy().item(),
            "psf_scale": self.psf_scale.detach().cpu().numpy().item(),
            "frozen": self.frozen,
            "weight": float(self.weight),
        }

    @classmethod
    def from_dict(cls, data):
        kwargs = {key: (bool(value) if key == "frozen" else float(value)) for key, value in data.items()}
        return cls(**kwargs)


class NPredCalibrations(nn.ModuleDict):
    """Calibrations by dataset name (reference: jolideco/models/npred.py:405-510)."""

    def parameters(self, recurse=True):
        parameters = []
        for model in self.values():
            if not model.frozen:
                parameters.extend(list(model.parameters()))
        return parameters

    def to_dict(self):
        return {name: model.to_dict() for name, model in self.items()}

    @classmethod
    def from_dict(cls, data):
        return cls([(name, NPredCalibration.from_dict(d)) for name, d in data.items()])

    @classmethod
    def read(cls, filename, format=None):
        """Read calibrations; format : {"yaml", "fits"} (reference: npred.py:466-486)."""
        from ..utils.io import IO_FORMATS_NPRED_CALIBRATIONS_READ, get_reader

        return get_reader(filename, format, IO_FORMATS_NPRED_CALIBRATIONS_READ)(filename)

    def write(self, filename, format=None, overwrite=False, **kwargs):
        """Write calibrations; format : {"yaml", "fits"} (reference: npred.py:488-510)."""
        from ..utils.io import IO_FORMATS_NPRED_CALIBRATIONS_WRITE, get_writer

        writer = get_writer(filename, format, IO_FORMATS_NPRED_CALIBRATIONS_WRITE)
        return writer(npred_calibrations=self, filename=filename, overwrite=overwrite, **kwargs)


class NPredModels(nn.ModuleDict):
    """All component models of one dataset plus its background (and its calibration)."""

    def __init__(self, background, calibration=None, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.register_buffer("background", background)
        # not registered as a sub-module: a ModuleDict iterates its modules as flux component models
        object.__setattr__(self, "calibration", calibration)

    def evaluate_per_component(self, fluxes):
        if self.calibration is not None:
            raise NotImplementedError(
                "the autograd seam NPredModels.evaluate does not apply calibrations; use fwd_bwd (the fit loop does)"
            )
        npreds = {name: model(flux=flux) for (name, model), flux in zip(self.items(), fluxes)}
        npreds["background"] = self.background
        return npreds

    def evaluate(self, fluxes):
        """Total predicted counts; differentiable w.r.t. the fluxes (models/npred.py:241-261)."""
        total = torch.zeros(self.background.shape, device=self.background.device)
        for npred in self.evaluate_per_component(fluxes=fluxes).values():
            total = total + npred
        return total

    @classmethod
    def from_dataset_numpy(cls, dataset, components, calibration=None, device="cuda", kernel_shape=None):
        """``kernel_shape``: a PSF array shape (before up-sampling) shared with OTHER datasets -- every component PSF is
        embedded in zeros up to it (same 'same' convolution), so that datasets with different PSF sizes share one plan
        and the batched joint step (`common_kernel_shape`)."""
        values = []
        psfs = dataset_psfs(dataset, components)
        factors = {c.upsampling_factor or 1 for c in components.values()}
        if len(factors) != 1:
            raise NotImplementedError("all components of a fit must share one upsampling_factor in jolideco_amd")
        own = (max(p.shape[0] for p in psfs.values()), max(p.shape[1] for p in psfs.values()))
        kernel_shape = own if kernel_shape is None else (max(own[0], kernel_shape[0]), max(own[1], kernel_shape[1]))
        psf_scale = None if calibration is None else float(calibration.psf_scale.detach().cpu())
        for allow_separable in (True, False):
            values = []
            for name, component in components.items():
                model = NPredModel.from_numpy(
                    exposure=dataset["exposure"], psf=psfs[name], upsampling_factor=component.upsampling_factor,
                    device=device, kernel_shape=kernel_shape, psf_scale=psf_scale, allow_separable=allow_separable,
                )
                values.append((name, model))
            # one fused call per dataset = one plan: if only some of the component PSFs are separable, none uses it
            if len({id(model.plan) for _, model in values}) <= 1:
                break
        background = _to_device_image(dataset["background"], device)[None, None]
        if calibration is not None:
            calibration = calibration.to(device)
        models = cls(background, calibration, values)
        models.shared_operator = cls._components_share_the_operator(psfs, values, calibration)
        return models

    # Several components, ONE operator.  The reference builds every component's model of a dataset from the same exposure
    # and, unless `psf` is a dict by component, the same PSF (models/npred.py:279-295), and adds the clipped convolutions
    # (models/npred.py:241-261, 194).  With a non-negative rank-1 PSF (the separable kernels then add products of
    # non-negative numbers only), a non-negative exposure and non-negative fluxes (exp(theta)) no term is ever clipped, so
    #   sum_c clip(PSF * (flux_c E), 0) = PSF * ((sum_c flux_c) E)      and      d loss / d flux_c = E x corr(g, PSF) for all c:
    # ONE forward model and ONE adjoint per dataset instead of one per component (`PoissonLoss.fwd_bwd_batch`).
    shared_operator = False

    @staticmethod
    def _components_share_the_operator(psfs, values, calibration):
        if len(values) < 2 or calibration is not None:
            return False
        arrays = [np.asarray(p) for p in psfs.values()]
        first = arrays[0]
        if any(a.shape != first.shape or not np.array_equal(a, first) for a in arrays[1:]):
            return False
        if not (np.isfinite(first).all() and first.min() >= 0.0) or psf_separable_rank(first) != 1:
            return False
        models = [m for _, m in values]
        if any(m.plan is not models[0].plan or m.plan.method != "separable" for m in models):
            return False
        if len({m.upsampling_factor or 1 for m in models}) != 1 or (models[0].upsampling_factor or 1) != 1:
            return False
        exposure = models[0].exposure
        return bool(torch.isfinite(exposure).all()) and bool((exposure >= 0).all())

    # fused path ----------------------------------------------------------------------------
    @property
    def plan(self):
        plans = {id(m.plan): m.plan for m in self.values()}
        if len(plans) != 1:
            raise NotImplementedError("components of one dataset must share the PSF shape (one FFT plan)")
        return next(iter(plans.values()))

    def _grad_buffer(self, name, like):
        buffers = self.__dict__.setdefault("_cal_grad_buffers", {})
        buf = buffers.get(name)
        if buf is None or buf.shape != like.shape or buf.device != like.device:
            buf = buffers[name] = torch.zeros_like(like)
        return buf

    def calibration_pointers(self, want_grad):
        """(shift_xy | None, log_background_norm, grad_shift_xy | None, grad_log_background_norm | None) of this dataset's
        calibration -- the device tensors the C-ABI reads and writes -- or None without a calibration.  Gradients land in
        `.grad` (allocated here, so a parameter that takes no part keeps grad None and the optimizer skips it, as in the
        reference)."""
        cal = self.calibration
        if cal is None:
            return None
        want_grad = want_grad and not cal.frozen
        shift = cal.shift_xy if self.shift_active else None
        if want_grad:
            # (the buffers are allocated once and re-attached after every zero_grad(set_to_none=True): the library OVERWRITES
            # both gradients, so no fill launch per parameter and step is needed -- 16 of them per step of 8 observations)
            if shift is not None and cal.shift_xy.grad is None:
                cal.shift_xy.grad = self._grad_buffer("shift_xy", cal.shift_xy)
            if cal._background_norm.grad is None:
                cal._background_norm.grad = self._grad_buffer("background_norm", cal._background_norm)
        return (
            None if shift is None else shift.data,
            cal._background_norm.data,
            cal.shift_xy.grad if (want_grad and shift is not None) else None,
            cal._background_norm.grad if want_grad else None,
        )

    def fwd_bwd(self, fluxes, counts, stirling, loss_out, grads=None, accumulate=False, grad_scale=1.0,
                npred_out=None, flux_nonneg=False):
        """One fused C-ABI call: forward model + Poisson NLL (+ d loss / d flux_c).  ``flux_nonneg``: the caller
        guarantees flux >= 0; components that share the operator (`shared_operator`) are then evaluated as their sum."""
        models = list(self.values())
        if (flux_nonneg and self.shared_operator and len(models) > 1 and not accumulate and npred_out is None
                and os.environ.get("JOLIDECO_MERGE_COMPONENTS", "1") != "0"):
            from ..ops import copy_image_to, sum_images

            fluxes = list(fluxes)
            total = self.__dict__.get("_merged_flux")
            if total is None or total.shape != fluxes[0].shape or total.device != fluxes[0].device:
                total = torch.empty_like(fluxes[0])
                object.__setattr__(self, "_merged_flux", total)
            sum_images(total, fluxes)
            self.plan.npred_poisson_fwd_bwd(
                fluxes=[total], exposures=[models[0].exposure], khats=[models[0].khat], background=self.background,
                counts=counts, stirling=stirling, loss_out=loss_out, grads=None if grads is None else [grads[0]],
                accumulate=False, grad_scale=grad_scale, npred_out=None, upsampling=1, calibration=None,
            )
            if grads is not None:
                copy_image_to(grads[0], list(grads[1:]))
            return
        calibration = self.calibration_pointers(grads is not None)
        self.plan.npred_poisson_fwd_bwd(
            fluxes=list(fluxes), exposures=[m.exposure for m in models], khats=[m.khat for m in models],
            background=self.background, counts=counts, stirling=stirling, loss_out=loss_out, grads=grads,
            accumulate=accumulate, grad_scale=grad_scale, npred_out=npred_out,
            upsampling=models[0].upsampling_factor or 1, calibration=calibration,
        )

    @property
    def shift_active(self):
        if self.calibration is None:
            return False
        if not hasattr(self, "_shift_active"):
            self._shift_active = self.calibration.shift_is_active
        return self._shift_active
