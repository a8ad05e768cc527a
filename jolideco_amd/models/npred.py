"""Predicted-counts models (reference: jolideco/models/npred.py:31-295).

`NPredModel` holds the edge-corrected exposure and the cached kernel spectrum K-hat of one
(dataset, component) pair on the HIP device; `NPredModels` adds the background.  `forward` /
`evaluate` are differentiable through the HIP convolution (`ops.ConvSameFunction`); the fit loop
uses the fused `fwd_bwd` instead (one C-ABI call per dataset).
"""
import numpy as np
import torch
import torch.nn as nn

from ..ops import ConvPlan, ConvSameFunction

__all__ = ["NPredModel", "NPredModels"]


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
                   kernel_shape=None):
        """Upload one dataset's exposure and PSF, compute K-hat once and apply the reference's
        edge correction exposure / conv(1, psf) (models/npred.py:66-115) on the device.

        ``kernel_shape`` (KH, KW) >= psf.shape embeds the PSF in a larger zero array such that the
        'same' crop is unchanged; it lets all components of a dataset share one FFT plan."""
        device = torch.device(device)
        psf = np.asarray(psf, dtype=np.float32)
        if kernel_shape is not None and tuple(kernel_shape) != psf.shape:
            psf = embed_kernel(psf, kernel_shape)
        exposure = np.ascontiguousarray(exposure, dtype=np.float32)
        if upsampling_factor:
            # setup-time bilinear up-sampling on the host, PSF divided by u^2 (models/npred.py:96-106)
            import torch.nn.functional as F

            up = lambda a: F.interpolate(torch.from_numpy(a)[None, None], scale_factor=upsampling_factor, mode="bilinear")[0, 0]  # noqa: E731
            exposure = up(exposure).numpy()
            psf = (up(np.ascontiguousarray(psf)) / upsampling_factor**2).numpy()
        exposure_t = _to_device_image(exposure, device)
        psf_t = _to_device_image(psf, device)
        H, W = exposure_t.shape
        kh, kw = psf_t.shape
        plan = ConvPlan.get(H, W, kh, kw, device)
        khat = plan.psf_spectrum(psf_t)
        if correct_exposure_edges:
            weights = plan.conv_same(torch.ones_like(exposure_t), None, khat)
            exposure_t = exposure_t / weights
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


class NPredModels(nn.ModuleDict):
    """All component models of one dataset plus its background."""

    def __init__(self, background, calibration=None, *args, **kwargs):
        super().__init__(*args, **kwargs)
        if calibration is not None:
            raise NotImplementedError("NPredCalibration is not implemented in jolideco_amd yet")
        self.register_buffer("background", background)
        self.calibration = None

    def evaluate_per_component(self, fluxes):
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
    def from_dataset_numpy(cls, dataset, components, calibration=None, device="cuda"):
        values = []
        psfs = {}
        for name in components.keys():
            psf = dataset["psf"]
            psfs[name] = np.asarray(psf[name] if isinstance(psf, dict) else psf)
        factors = {c.upsampling_factor or 1 for c in components.values()}
        if len(factors) != 1:
            raise NotImplementedError("all components of a fit must share one upsampling_factor in jolideco_amd")
        kernel_shape = (max(p.shape[0] for p in psfs.values()), max(p.shape[1] for p in psfs.values()))
        for name, component in components.items():
            model = NPredModel.from_numpy(
                exposure=dataset["exposure"], psf=psfs[name], upsampling_factor=component.upsampling_factor,
                device=device, kernel_shape=kernel_shape,
            )
            values.append((name, model))
        background = _to_device_image(dataset["background"], device)[None, None]
        return cls(background, calibration, values)

    # fused path ----------------------------------------------------------------------------
    @property
    def plan(self):
        plans = {id(m.plan): m.plan for m in self.values()}
        if len(plans) != 1:
            raise NotImplementedError("components of one dataset must share the PSF shape (one FFT plan)")
        return next(iter(plans.values()))

    def fwd_bwd(self, fluxes, counts, stirling, loss_out, grads=None, accumulate=False, grad_scale=1.0,
                npred_out=None):
        """One fused C-ABI call: forward model + Poisson NLL (+ d loss / d flux_c)."""
        models = list(self.values())
        self.plan.npred_poisson_fwd_bwd(
            fluxes=list(fluxes), exposures=[m.exposure for m in models], khats=[m.khat for m in models],
            background=self.background, counts=counts, stirling=stirling, loss_out=loss_out, grads=grads,
            accumulate=accumulate, grad_scale=grad_scale, npred_out=npred_out,
            upsampling=models[0].upsampling_factor or 1,
        )
