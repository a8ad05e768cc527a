from .core import FluxComponents, SpatialFluxComponent
from .npred import NPredCalibration, NPredCalibrations, NPredModel, NPredModels

__all__ = ["FluxComponents", "SpatialFluxComponent", "NPredModel", "NPredModels", "NPredCalibration", "NPredCalibrations"]
