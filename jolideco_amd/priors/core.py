"""Prior base classes and the element-wise priors (reference: jolideco/priors/core.py).

Every prior exposes two entry points:
  * ``__call__(flux) -> 0-dim tensor`` : differentiable log-prior (autograd seam, HIP backward)
  * ``device_fwd_bwd(flux, value_out, grad, coef)`` : the fused path used by the fit loop, which
    writes the value into a device scalar and accumulates ``coef * d logprior / d flux`` into
    ``grad`` without going through autograd.
"""
import math

import torch
import torch.nn as nn

from .. import _hip
from .._hip import check, ptr, stream_ptr

__all__ = ["Prior", "Priors", "UniformPrior", "InverseGammaPrior", "ExponentialPrior"]


class Prior(nn.Module):
    """Prior base class"""

    # torch.Generator cannot be deep-copied / pickled: carry its state instead
    # (same work-around as jolideco/priors/core.py:28-47)
    def __getstate__(self):
        state = self.__dict__.copy()
        generator = state.pop("generator", None)
        state.pop("_handle", None)
        if generator is not None:
            state["generator"] = generator.get_state()
        return state

    def __setstate__(self, state):
        generator_state = state.pop("generator", None)
        state.pop("generator-device", None)
        if generator_state is not None:
            generator = torch.Generator(device="cpu")
            generator.set_state(generator_state)
            state["generator"] = generator
        self.__dict__ = state

    def to_dict(self):
        from . import PRIOR_REGISTRY

        for name, cls in PRIOR_REGISTRY.items():
            if isinstance(self, cls):
                return {"type": name}
        return {}

    def hessian_ones(self, flux):
        """Hessian of the log-prior times a vector of ones (what `torch.autograd.functional.vhp(..., v=ones)`
        of jolideco/loss.py:263-279 yields for this prior's term).  Zero unless the prior has curvature."""
        return torch.zeros_like(flux)

    @classmethod
    def from_dict(cls, data):
        from . import PRIOR_REGISTRY

        kwargs = dict(data)
        if "type" in kwargs:
            type_ = kwargs.pop("type")
            if type_ not in PRIOR_REGISTRY:
                raise NotImplementedError(f"prior {type_!r} is not implemented in jolideco_amd")
            return PRIOR_REGISTRY[type_].from_dict(kwargs)
        return cls(**kwargs)

    # fused path ----------------------------------------------------------------------------
    def device_fwd_bwd(self, flux, value_out, grad=None, coef=0.0, patch_rows=None):
        raise NotImplementedError

    #: True if the value is a sum over patch rows that can be sharded across ranks
    shardable = False


class Priors(nn.ModuleDict):
    """Dict of multiple priors"""

    def __call__(self, fluxes):
        value = 0
        for idx, prior in enumerate(self.values()):
            value = value + prior(flux=fluxes[idx])
        return value


class UniformPrior(Prior):
    """Uniform prior: log-prior 0, no gradient (jolideco/priors/core.py:110-129)."""

    def __init__(self):
        super().__init__()

    def __call__(self, flux):
        return torch.tensor(0)

    value_is_zero = True  # log-prior 0, gradient 0: a session whose slot for the value already holds 0 skips the call

    def device_fwd_bwd(self, flux, value_out, grad=None, coef=0.0, patch_rows=None):
        value_out.zero_()


class _ElementwisePrior(Prior):
    _kind = 0

    def _params(self):
        raise NotImplementedError

    def __call__(self, flux):
        from ..ops import ElementwisePriorFunction

        alpha, beta, log_const = self._params()
        return ElementwisePriorFunction.apply(flux, self._kind, alpha, beta, log_const)

    def device_fwd_bwd(self, flux, value_out, grad=None, coef=0.0, patch_rows=None):
        alpha, beta, log_const = self._params()
        n = flux.numel()
        check(
            _hip.lib().jd_elementwise_prior_fwd_bwd(
                self._kind, ptr(flux), n, alpha, beta, log_const, ptr(value_out), coef / n, ptr(grad),
                stream_ptr(flux.device),
            )
        )


class InverseGammaPrior(_ElementwisePrior):
    """Product of inverse-Gamma distributions, sparse prior for point sources
    (jolideco/priors/core.py:132-240): mean_i(-beta/x_i - (alpha+1) log x_i) + alpha log beta - lgamma(alpha)."""

    _kind = 1

    def __init__(self, alpha=10, beta=3 / 2, cycle_spin_subpix=False, generator=None):
        super().__init__()
        if cycle_spin_subpix:
            raise NotImplementedError("cycle_spin_subpix is not implemented in jolideco_amd")
        self.alpha = float(alpha)
        self.beta = float(beta)
        self.cycle_spin_subpix = False

    @property
    def mean(self):
        return self.beta / (self.alpha - 1)

    @property
    def mode(self):
        return self.beta / (self.alpha + 1)

    @property
    def log_constant_term(self):
        a, b = torch.tensor([self.alpha]), torch.tensor([self.beta])
        return float(a * torch.log(b) - torch.lgamma(a))

    def _params(self):
        return self.alpha, self.beta, self.log_constant_term

    def hessian_ones(self, flux):
        """The prior is a mean of element-wise terms, so its Hessian is diagonal:
        d2/dx2 (-beta/x - (alpha+1) log x) / n = (-2 beta / x^3 + (alpha+1) / x^2) / n."""
        return (-2.0 * self.beta / flux**3 + (self.alpha + 1.0) / flux**2) / flux.numel()

    def to_dict(self):
        data = super().to_dict()
        data.update(alpha=self.alpha, beta=self.beta, cycle_spin_subpix=False)
        return data


class ExponentialPrior(_ElementwisePrior):
    """Product of exponential distributions (jolideco/priors/core.py:243-339):
    mean_i(-alpha x_i) + log alpha."""

    _kind = 2

    def __init__(self, alpha=10, cycle_spin_subpix=False, generator=None):
        super().__init__()
        if cycle_spin_subpix:
            raise NotImplementedError("cycle_spin_subpix is not implemented in jolideco_amd")
        self.alpha = float(alpha)
        self.cycle_spin_subpix = False

    @property
    def mean(self):
        return 1 / self.alpha

    @property
    def mode(self):
        return 0

    @property
    def log_constant_term(self):
        return float(torch.log(torch.tensor([self.alpha])))

    def _params(self):
        return self.alpha, 0.0, self.log_constant_term

    def to_dict(self):
        data = super().to_dict()
        data.update(alpha=self.alpha, cycle_spin_subpix=False)
        return data
