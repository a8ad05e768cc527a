"""GMM patch prior (reference: jolideco/priors/patches/core.py:30-246)."""
import logging

import torch

from ...utils.norms import IdentityImageNorm, ImageNorm, PatchNorm, SubtractMeanPatchNorm
from ...utils.torch import TORCH_DEFAULT_DEVICE, cycle_spin_shifts, cycle_spin_shifts_many, get_default_generator
from ..core import Prior
from .gmm import GaussianMixtureModel

__all__ = ["GMMPatchPrior"]

log = logging.getLogger(__name__)


class GMMPatchPrior(Prior):
    """Patch prior: expected (max or marginal) GMM log-likelihood of all overlapping patches.

    Same constructor as the reference.  Options that are not on the accelerated path
    (``cycle_spin_subpix``, ``jitter``, non-identity ``norm``, other patch norms) raise
    NotImplementedError instead of silently running something else.
    """

    shardable = True

    def __init__(
        self,
        gmm=None,
        stride=None,
        cycle_spin=True,
        cycle_spin_subpix=False,
        generator=None,
        norm=None,
        patch_norm=None,
        jitter=False,
        marginalize=False,
        device=TORCH_DEFAULT_DEVICE,
    ):
        super().__init__()
        if gmm is None:
            gmm = GaussianMixtureModel.from_registry(name="zoran-weiss")
        self.gmm = gmm
        self.stride = gmm.meta.stride if stride is None else stride
        if self.stride is None:
            raise ValueError("stride must be given either explicitly or through gmm.meta.stride")
        self.cycle_spin = cycle_spin
        if cycle_spin_subpix:
            raise NotImplementedError("cycle_spin_subpix is not implemented in jolideco_amd")
        if jitter:
            raise NotImplementedError("jitter is not implemented in jolideco_amd")
        self.cycle_spin_subpix = False
        self.jitter = False
        # shifts are ALWAYS drawn from a host generator (torch default seed), see utils/torch.py
        self.generator = generator if generator is not None else get_default_generator("cpu")
        if self.generator.device.type != "cpu":
            raise ValueError("the cycle-spin generator must be a CPU generator")
        norm = norm if norm is not None else IdentityImageNorm()
        if not isinstance(norm, IdentityImageNorm):
            raise NotImplementedError("only IdentityImageNorm is implemented in jolideco_amd")
        self.norm = norm
        patch_norm = patch_norm if patch_norm is not None else gmm.meta.patch_norm
        if not isinstance(patch_norm, SubtractMeanPatchNorm):
            raise NotImplementedError("only SubtractMeanPatchNorm is implemented in jolideco_amd")
        self.patch_norm = patch_norm
        self.marginalize = marginalize
        self.device = torch.device(device)
        self.last_shifts = None

    @property
    def patch_shape(self):
        return self.gmm.patch_shape

    @property
    def overlap(self):
        return max(self.patch_shape) - self.stride

    @property
    def log_like_weight(self):
        """stride^2 / patch area (priors/patches/core.py:222-225)."""
        return self.stride**2 / (self.patch_shape[0] * self.patch_shape[1])

    def draw_shifts(self):
        """One pair of cycle-spin draws (None when cycle_spin is off)."""
        shifts = cycle_spin_shifts(self.patch_shape, self.generator) if self.cycle_spin else None
        self.last_shifts = shifts
        return shifts

    def draw_shifts_many(self, n):
        """The next `n` pairs of cycle-spin draws, in order (an epoch's worth, `FitSession._plan_epoch`)."""
        if not self.cycle_spin:
            self.last_shifts = None
            return [None] * n
        shifts = cycle_spin_shifts_many(self.patch_shape, self.generator, n)
        self.last_shifts = shifts[-1]
        return shifts

    def __call__(self, flux, mask=None):
        """Differentiable log-prior of a (1, 1, H, W) HIP tensor; draws one pair of shifts."""
        from ...ops import GMMPatchPriorFunction

        shifts = self.draw_shifts()
        scale = self.log_like_weight / flux.numel()
        return GMMPatchPriorFunction.apply(
            flux, self.gmm.handle(flux.device), self.stride, shifts, self.marginalize, scale
        )

    def device_fwd_bwd(self, flux, value_out, grad=None, coef=0.0, patch_rows=None, shifts="draw", band_out=None, phases=3):
        """Fused path: value -> device scalar, ``grad += coef * d logprior / d flux``; with ``band_out`` the gradient of
        the shard ``patch_rows`` goes, un-accumulated, to the band of the rolled frame it covers (sharded joint fit)."""
        if isinstance(shifts, str):
            shifts = self.draw_shifts()
        scale = self.log_like_weight / flux.numel()
        self.gmm.handle(flux.device).prior_fwd_bwd(
            flux.reshape(flux.shape[-2:]), self.stride, shifts, value_out, scale, grad=grad, grad_coef=coef * scale,
            marginalize=self.marginalize, patch_rows=patch_rows or (0, -1), band_out=band_out, phases=phases,
        )

    # the optimizer step of the component can ride in the epilogue of this prior's last kernel (`device_fwd_bwd_step`)
    supports_fused_step = True
    # the pass splits into phase 1 (value + gradient rows: reads the flux only) and phase 2 (gather [+ step]): phase 1 may run
    # on a second stream beside the likelihood launches (`phases` of device_fwd_bwd[_step]; FitSession)
    supports_phases = True

    def device_fwd_bwd_step(self, flux, value_out, coef, step, shifts="draw", phases=3):
        """`device_fwd_bwd` of the WHOLE prior with the component's optimizer step applied by its gather kernel:
        ``step.grad_flux`` holds every other gradient term; theta, the moments and the new flux are written in place
        (jd_gmm_prior_fwd_bwd_step).  Same numbers as `device_fwd_bwd` followed by the stand-alone step."""
        if isinstance(shifts, str):
            shifts = self.draw_shifts()
        scale = self.log_like_weight / flux.numel()
        self.gmm.handle(flux.device).prior_fwd_bwd_step(
            flux.reshape(flux.shape[-2:]), self.stride, shifts, value_out, scale, coef * scale, step,
            marginalize=self.marginalize, phases=phases,
        )

    def hessian_ones(self, flux):
        """Every patch has its mean subtracted before the mixture sees it, so a constant vector is in the null
        space of each patch's quadratic form: Hessian x ones is exactly zero (the reference's double backward
        returns rounding noise there).  One pair of shifts is drawn, as the reference's evaluation does."""
        self.draw_shifts()
        return torch.zeros_like(flux)

    def n_patch_rows(self, shape):
        return (shape[-2] - self.patch_shape[0]) // self.stride + 1

    def to_dict(self):
        data = super().to_dict()
        data.update(
            stride=int(self.stride), cycle_spin=bool(self.cycle_spin), cycle_spin_subpix=False, jitter=False,
            gmm=self.gmm.to_dict(), norm=self.norm.to_dict(), patch_norm=self.patch_norm.to_dict(),
            device=str(self.device),
        )
        if self.marginalize:
            data["marginalize"] = True
        return data

    @classmethod
    def from_dict(cls, data):
        """Rebuild from `to_dict` output / a file header (patches/core.py:110-121): the GMM is looked up
        by name in the user's GMM library."""
        kwargs = dict(data)
        kwargs.pop("type", None)
        kwargs["gmm"] = GaussianMixtureModel.from_dict(kwargs.pop("gmm"))
        kwargs["norm"] = ImageNorm.from_dict(kwargs.pop("norm", {"type": "identity"}))
        if "patch_norm" in kwargs:
            kwargs["patch_norm"] = PatchNorm.from_dict(kwargs["patch_norm"])
        return cls(**kwargs)
