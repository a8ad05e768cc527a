from .core import ExponentialPrior, InverseGammaPrior, Prior, Priors, UniformPrior
from .patches import GaussianMixtureModel, GMMPatchPrior

PRIOR_REGISTRY = {
    "uniform": UniformPrior,
    "gmm-patches": GMMPatchPrior,
    "inverse-gamma": InverseGammaPrior,
    "exponential": ExponentialPrior,
}

__all__ = [
    "GaussianMixtureModel",
    "GMMPatchPrior",
    "ExponentialPrior",
    "UniformPrior",
    "InverseGammaPrior",
    "Prior",
    "Priors",
    "PRIOR_REGISTRY",
]
