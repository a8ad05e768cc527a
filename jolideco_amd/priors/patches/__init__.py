from .core import GMMPatchPrior
from .gmm import GaussianMixtureModel, GaussianMixtureModelMeta

__all__ = ["GMMPatchPrior", "GaussianMixtureModel", "GaussianMixtureModelMeta"]
