"""jolideco_amd -- MI355X-native MAP deconvolution inner loop with Jolideco's API surface.

The compute path is libjolideco_hip.so (hand-written HIP for gfx950 + rocFFT), bound through
ctypes (`jolideco_amd._hip`).  PyTorch provides device memory, streams and `torch.distributed`.
"""
from .core import MAPDeconvolver, MAPDeconvolverResult
from .loss import PoissonLoss, PriorLoss, TotalLoss
from .models import (
    FluxComponents,
    NPredCalibration,
    NPredCalibrations,
    NPredModel,
    NPredModels,
    SpatialFluxComponent,
)
from .priors import (
    ExponentialPrior,
    GaussianMixtureModel,
    GMMPatchPrior,
    InverseGammaPrior,
    Priors,
    UniformPrior,
)

__version__ = "0.1.0"

__all__ = [
    "MAPDeconvolver",
    "MAPDeconvolverResult",
    "PoissonLoss",
    "PriorLoss",
    "TotalLoss",
    "FluxComponents",
    "SpatialFluxComponent",
    "NPredModel",
    "NPredModels",
    "NPredCalibration",
    "NPredCalibrations",
    "GaussianMixtureModel",
    "GMMPatchPrior",
    "UniformPrior",
    "InverseGammaPrior",
    "ExponentialPrior",
    "Priors",
]
