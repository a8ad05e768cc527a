"""YAML side of Jolideco's I/O (reference: jolideco/utils/io/yaml.py).

A flux component is a YAML mapping of its settings whose ``flux_upsampled`` entry is the absolute
path of a companion ``<name>-data.fits`` image; calibrations are a plain mapping.  The reference
dumps with ruamel.yaml (block style); PyYAML's block style is the same text for these plain trees.
"""
import logging
from pathlib import Path

import numpy as np
import yaml

log = logging.getLogger(__name__)

__all__ = ["to_yaml_str", "from_yaml_str"]


def _plain(value):
    """numpy scalars / tuples -> builtin types, so that the dump carries no python tags."""
    if isinstance(value, dict):
        return {str(k): _plain(v) for k, v in value.items()}
    if isinstance(value, (list, tuple)):
        return [_plain(v) for v in value]
    if isinstance(value, np.generic):
        return value.item()
    return value


def to_yaml_str(data):
    return yaml.safe_dump(_plain(data), default_flow_style=False, sort_keys=False)


def from_yaml_str(yaml_str):
    return yaml.safe_load(yaml_str)


def write_yaml(filename, data, overwrite):
    path = Path(filename)
    if path.exists() and not overwrite:
        raise OSError(f"{filename} already exists!")
    log.info(f"Writing {filename}")
    path.write_text(to_yaml_str(data))


def load_yaml(filename):
    path = Path(filename)
    log.info(f"Reading {path}")
    return from_yaml_str(path.read_text())


def flux_component_to_yaml_dict(flux_component, filename, name=None):
    """Settings + the path of the companion data file (reference: yaml.py:97-119)."""
    path = Path(filename)
    data = flux_component.to_dict()
    data["upsampling_factor"] = int(data["upsampling_factor"] or 1)
    data["flux_upsampled"] = str((path.parent / f"{name or path.stem}-data.fits").absolute())
    return data


def write_flux_component_to_yaml(flux_component, filename, overwrite):
    data = flux_component_to_yaml_dict(flux_component, filename)
    flux_component.write(data["flux_upsampled"], overwrite=overwrite)
    write_yaml(filename, data, overwrite)


def write_flux_components_to_yaml(flux_components, filename, overwrite):
    data = {}
    for name, component in flux_components.items():
        data[name] = flux_component_to_yaml_dict(component, filename, name=name)
        component.write(data[name]["flux_upsampled"], overwrite=overwrite)
    write_yaml(filename, data, overwrite)


def read_flux_component_from_yaml(filename):
    from ...models import SpatialFluxComponent

    return SpatialFluxComponent.from_dict(load_yaml(filename))


def read_flux_components_from_yaml(filename):
    from ...models import FluxComponents

    return FluxComponents.from_dict(load_yaml(filename))


def read_npred_calibrations_from_yaml(filename):
    from ...models import NPredCalibrations

    return NPredCalibrations.from_dict(load_yaml(filename))


def write_npred_calibrations_to_yaml(npred_calibrations, filename, overwrite):
    write_yaml(filename, npred_calibrations.to_dict(), overwrite)
