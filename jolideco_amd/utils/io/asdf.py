"""ASDF side of Jolideco's I/O (reference: jolideco/utils/io/asdf.py).

Same functions and the same trees as the reference -- a flux component is its ``to_dict(include_data="numpy")``
(:27), flux components the mapping of those by name (:82-84), a MAP result ``{"components", "components-init",
"trace-loss", "config"}`` (:123-130) -- written and read with the self-contained codec of `_asdffile` instead of the
``asdf`` package (not available here).  A result file additionally carries ``calibrations`` /
``calibrations-init`` (plain mappings; the reference's reader ignores keys it does not know), so that an ASDF
checkpoint holds what a FITS one holds.
"""
import logging
from pathlib import Path

from ._asdffile import read_asdf, write_asdf

log = logging.getLogger(__name__)

__all__ = [
    "write_flux_component_to_asdf",
    "read_flux_component_from_asdf",
    "write_flux_components_to_asdf",
    "read_flux_components_from_asdf",
    "write_map_result_to_asdf",
    "read_map_result_from_asdf",
]


def _write(tree, filename, overwrite):
    path = Path(filename)
    if path.exists() and not overwrite:
        raise OSError(f"{path} already exists!")
    log.info(f"writing {path}")
    write_asdf(path, tree, overwrite=True)


def write_flux_component_to_asdf(flux_component, filename, overwrite, **kwargs):
    """Flux component(s) -> ASDF: the tree is ``to_dict(include_data="numpy")`` (reference: asdf.py:9-39)."""
    _write(flux_component.to_dict(include_data="numpy"), filename, overwrite)


def read_flux_component_from_asdf(filename):
    """ASDF -> `SpatialFluxComponent` (reference: asdf.py:42-63)."""
    from ...models import SpatialFluxComponent

    return SpatialFluxComponent.from_dict(data=read_asdf(filename))


def write_flux_components_to_asdf(flux_components, filename, overwrite, **kwargs):
    """`FluxComponents` -> ASDF, one mapping per component name (reference: asdf.py:66-84)."""
    write_flux_component_to_asdf(flux_component=flux_components, filename=filename, overwrite=overwrite, **kwargs)


def read_flux_components_from_asdf(filename):
    """ASDF -> `FluxComponents` (reference: asdf.py:87-109)."""
    from ...models import FluxComponents

    return FluxComponents.from_dict(data=read_asdf(filename))


def write_map_result_to_asdf(result, filename, overwrite, **kwargs):
    """`MAPDeconvolverResult` -> ASDF (reference: asdf.py:112-142): components, initial components, the loss trace as a
    table, the configuration."""
    tree = {"components": result.components.to_dict(include_data="numpy")}
    if result.components_init is not None:
        tree["components-init"] = result.components_init.to_dict(include_data="numpy")
    tree["trace-loss"] = result.trace_loss
    tree["config"] = result.config
    if getattr(result, "calibrations", None):
        tree["calibrations"] = result.calibrations.to_dict()
        if getattr(result, "calibrations_init", None):
            tree["calibrations-init"] = result.calibrations_init.to_dict()
    _write(tree, filename, overwrite)


def _trace_from_table(table):
    from ..table import TraceTable

    trace = TraceTable(names=table.colnames)
    columns = {name: table[name] for name in table.colnames}
    for i in range(len(table)):
        trace.add_row({name: (str(col[i]) if col.dtype.kind in "US" else col[i].item()) for name, col in columns.items()})
    return trace


def read_map_result_from_asdf(filename):
    """ASDF -> `MAPDeconvolverResult` (reference: asdf.py:145-185)."""
    from ...core import MAPDeconvolverResult
    from ...models import FluxComponents, NPredCalibrations

    log.info(f"Reading {filename}")
    data = read_asdf(filename)
    components = FluxComponents.from_dict(data=data["components"])
    components_init = FluxComponents.from_dict(data=data["components-init"]) if "components-init" in data else None
    calibrations = NPredCalibrations.from_dict(data["calibrations"]) if data.get("calibrations") else None
    calibrations_init = NPredCalibrations.from_dict(data["calibrations-init"]) if data.get("calibrations-init") else None
    return MAPDeconvolverResult(
        config=data["config"], components=components, components_init=components_init,
        trace_loss=_trace_from_table(data["trace-loss"]), calibrations=calibrations, calibrations_init=calibrations_init,
    )
