"""Flux components (reference: jolideco/models/core.py:354-607,720-842).

A `SpatialFluxComponent` owns the log-flux parameter theta as an `nn.Parameter` so that user code
that inspects `.parameters()` keeps working.  During a fit the parameter, its flux image, the
gradient accumulator and the optimizer moments live on the HIP device and are updated by the fused
kernels (csrc/elementwise.hip); `flux_upsampled` is the autograd-visible `exp(theta) [* mask]`.
"""
import logging
from pathlib import Path

import numpy as np
import torch
import torch.nn as nn

from ..priors import Prior, Priors, UniformPrior
from ..priors.patches.gmm import GMMNotAvailableError
from ..utils.io import (
    IO_FORMATS_FLUX_COMPONENT_READ,
    IO_FORMATS_FLUX_COMPONENT_WRITE,
    IO_FORMATS_FLUX_COMPONENTS_READ,
    IO_FORMATS_FLUX_COMPONENTS_WRITE,
    get_reader,
    get_writer,
)

__all__ = ["SpatialFluxComponent", "FluxComponents"]

log = logging.getLogger(__name__)


def parse_flux_tensor(value, cls):
    """A flux given as a file name, a 2-D numpy array or a tensor -> (1, 1, H, W) float32 tensor
    (reference: models/core.py:41-51)."""
    if isinstance(value, (str, Path)):
        return cls.read(Path(value)).flux_upsampled.detach()
    if not isinstance(value, torch.Tensor):
        return torch.from_numpy(np.asarray(value)[np.newaxis, np.newaxis].astype(np.float32))
    return value


class SpatialFluxComponent(nn.Module):
    """Dense spatial flux component.

    Parameters
    ----------
    flux_upsampled : `~torch.Tensor`
        Initial flux, shape (1, 1, H, W).
    mask : `~torch.Tensor`
        Optional boolean mask multiplied onto the flux.
    use_log_flux : bool
        Optimise log(flux) (default); False optimises the flux itself (no positivity constraint).
    upsampling_factor : int
        Up-sampling factor of the flux grid w.r.t. the counts grid (None / 1 = none).
    prior : `Prior`
        Prior of this component (default uniform).
    frozen : bool
        Exclude the component from the optimisation.
    """

    is_sparse = False

    def __init__(
        self,
        flux_upsampled,
        flux_upsampled_error=None,
        mask=None,
        use_log_flux=True,
        upsampling_factor=1,
        prior=None,
        frozen=False,
        wcs=None,
    ):
        super().__init__()
        if not flux_upsampled.ndim == 4:
            raise ValueError(f"Flux tensor must be four dimensional. Got {flux_upsampled.ndim}")
        if upsampling_factor is not None and (int(upsampling_factor) != upsampling_factor or upsampling_factor < 1):
            raise ValueError(f"upsampling_factor must be a positive integer, got {upsampling_factor}")
        flux_upsampled = flux_upsampled.to(torch.float32)
        if use_log_flux:
            flux_upsampled = torch.log(flux_upsampled)
        self._flux_upsampled = nn.Parameter(flux_upsampled)
        self._flux_upsampled_error = flux_upsampled_error
        if mask is not None and not mask.shape == flux_upsampled.shape:
            raise ValueError(
                f"Flux and mask need to have the same shape, got {flux_upsampled.shape} and {mask.shape}"
            )
        self.mask = mask
        self._use_log_flux = bool(use_log_flux)
        self.upsampling_factor = None if upsampling_factor is None else int(upsampling_factor)
        self.prior = prior if prior is not None else UniformPrior()
        self.frozen = frozen
        self._wcs = wcs

    @classmethod
    def from_numpy(cls, flux, mask=None, **kwargs):
        """Create from a 2-D numpy flux image on the counts grid; with ``upsampling_factor`` the flux
        (and mask) are bilinearly up-sampled first (reference: models/core.py:505-540)."""
        import torch.nn.functional as F

        upsampling_factor = kwargs.get("upsampling_factor", None)
        flux = torch.from_numpy(np.asarray(flux)[np.newaxis, np.newaxis].astype(np.float32))
        if upsampling_factor:
            flux = F.interpolate(flux, scale_factor=upsampling_factor, mode="bilinear")
        if mask is not None:
            mask = torch.from_numpy(np.asarray(mask)[np.newaxis, np.newaxis].astype(bool))
            if upsampling_factor:
                mask = F.interpolate(mask.type(torch.float32), scale_factor=upsampling_factor, mode="bilinear") > 0.5
        return cls(flux_upsampled=flux, mask=mask, **kwargs)

    @classmethod
    def from_flux_init_datasets(cls, datasets, **kwargs):
        """Average of counts / exposure - background over datasets (models/core.py:542-566)."""
        fluxes = [d["counts"] / d["exposure"] - d["background"] for d in datasets]
        return cls.from_numpy(flux=np.nanmean(fluxes, axis=0), **kwargs)

    def parameters(self, recurse=True):
        return [] if self.frozen else super().parameters(recurse)

    @property
    def wcs(self):
        return self._wcs

    @property
    def shape(self):
        return self._flux_upsampled.shape

    @property
    def shape_image(self):
        return tuple(self.shape[-2:])

    @property
    def use_log_flux(self):
        return self._use_log_flux

    @property
    def flux_upsampled(self):
        """exp(theta) [* mask] as an autograd-visible tensor (models/core.py:583-594)."""
        flux = torch.exp(self._flux_upsampled) if self._use_log_flux else self._flux_upsampled
        if self.mask is not None:
            flux = flux * self.mask.to(flux.device)
        return flux

    @property
    def flux(self):
        """Flux on the counts grid: sum-pool of the up-sampled flux (models/core.py:596-607)."""
        import torch.nn.functional as F

        flux = self.flux_upsampled
        if self.upsampling_factor:
            flux = F.avg_pool2d(flux, kernel_size=self.upsampling_factor, divisor_override=1)
        return flux

    @property
    def flux_upsampled_error(self):
        return self._flux_upsampled_error

    @property
    def flux_upsampled_error_numpy(self):
        """Flux error on the up-sampled grid as a numpy array (models/core.py:626-630)."""
        return self.flux_upsampled_error.detach().cpu().numpy()[0, 0]

    @property
    def flux_numpy(self):
        return self.flux.detach().cpu().numpy()[0, 0]

    @property
    def flux_upsampled_numpy(self):
        return self.flux_upsampled.detach().cpu().numpy()[0, 0]

    def to_dict(self, include_data=None):
        data = {
            "use_log_flux": self._use_log_flux,
            "upsampling_factor": self.upsampling_factor,
            "frozen": self.frozen,
            "prior": self.prior.to_dict(),
        }
        if include_data == "numpy":
            data["flux_upsampled"] = self.flux_upsampled_numpy
            if self.flux_upsampled_error is not None:
                data["flux_upsampled_error"] = self.flux_upsampled_error_numpy
            if self.mask is not None:
                data["mask"] = self.mask.cpu().numpy()
        return data

    @classmethod
    def from_dict(cls, data):
        """Create from `to_dict` output or a file header (reference: models/core.py:455-487).  A GMM
        prior whose model is not in the user's GMM library cannot be rebuilt from its name: the
        component is then created with a uniform prior and a warning."""
        kwargs = dict(data)
        prior_data = kwargs.pop("prior", None)
        if prior_data:
            try:
                kwargs["prior"] = Prior.from_dict(prior_data)
            except GMMNotAvailableError as error:
                log.warning(f"{error}; the component gets a uniform prior instead of {prior_data.get('type')}")
        kwargs["flux_upsampled"] = parse_flux_tensor(kwargs["flux_upsampled"], cls)
        if "flux_upsampled_error" in kwargs:
            kwargs["flux_upsampled_error"] = parse_flux_tensor(kwargs["flux_upsampled_error"], cls)
        if "mask" in kwargs:
            kwargs["mask"] = torch.from_numpy(np.asarray(kwargs["mask"]).astype(bool))
        kwargs.pop("shape", None)
        return cls(**kwargs)

    @classmethod
    def read(cls, filename, format=None):
        """Read a flux component; format : {"fits", "yaml"} (default: from the suffix)."""
        return get_reader(filename, format, IO_FORMATS_FLUX_COMPONENT_READ)(filename)

    def write(self, filename, format=None, overwrite=False, **kwargs):
        """Write the flux component; format : {"fits", "yaml"} (default: from the suffix)."""
        writer = get_writer(filename, format, IO_FORMATS_FLUX_COMPONENT_WRITE)
        return writer(flux_component=self, filename=filename, overwrite=overwrite, **kwargs)


class FluxComponents(nn.ModuleDict):
    """Dict of flux components (reference: models/core.py:720-842)."""

    def parameters(self):
        parameters = []
        for component in self.values():
            if not component.frozen:
                parameters.extend(component.parameters())
        return parameters

    @property
    def priors(self):
        priors = Priors()
        for name, component in self.items():
            priors[name] = component.prior
        return priors

    def to_numpy(self):
        return {name: np.squeeze(c.flux_upsampled.detach().cpu().numpy()) for name, c in self.items()}

    @property
    def fluxes_numpy(self):
        return {name: c.flux_numpy for name, c in self.items()}

    @property
    def fluxes_upsampled_numpy(self):
        return self.to_numpy()

    @property
    def flux_upsampled_total_numpy(self):
        return np.sum([flux for flux in self.fluxes_upsampled_numpy.values()], axis=0)

    @property
    def flux_total_numpy(self):
        return np.sum([flux for flux in self.fluxes_numpy.values()], axis=0)

    def to_flux_tuple(self):
        return tuple(c.flux_upsampled for c in self.values())

    def set_flux_errors(self, flux_errors):
        for name, flux_error in flux_errors.items():
            self[name]._flux_upsampled_error = flux_error

    def to_dict(self, include_data=None):
        return {name: c.to_dict(include_data=include_data) for name, c in self.items()}

    @classmethod
    def from_dict(cls, data):
        return cls([(name, SpatialFluxComponent.from_dict(d)) for name, d in data.items()])

    @classmethod
    def read(cls, filename, format=None):
        """Read flux components; format : {"fits", "yaml"} (default: from the suffix)."""
        return get_reader(filename, format, IO_FORMATS_FLUX_COMPONENTS_READ)(filename)

    def write(self, filename, overwrite=False, format=None, **kwargs):
        """Write flux components; format : {"fits", "yaml"} (default: from the suffix)."""
        writer = get_writer(filename, format, IO_FORMATS_FLUX_COMPONENTS_WRITE)
        return writer(flux_components=self, filename=filename, overwrite=overwrite, **kwargs)
