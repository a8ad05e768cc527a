from .core import FluxComponents, SpatialFluxComponent
from .npred import NPredModel, NPredModels

__all__ = ["FluxComponents", "SpatialFluxComponent", "NPredModel", "NPredModels"]
