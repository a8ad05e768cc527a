"""File formats either side of the hot path (reference: jolideco/utils/io/__init__.py).

FITS (`_fitsfile`), ASDF (`_asdffile`) and YAML, each on a self-contained codec: the reference uses astropy, the
``asdf`` package and ruamel.yaml, none of which this image has.
"""
from pathlib import Path

from .asdf import (
    read_flux_component_from_asdf,
    read_flux_components_from_asdf,
    read_map_result_from_asdf,
    write_flux_component_to_asdf,
    write_flux_components_to_asdf,
    write_map_result_to_asdf,
)
from .fits import (
    read_flux_component_from_fits,
    read_flux_components_from_fits,
    read_map_result_from_fits,
    read_npred_calibrations_from_fits,
    write_flux_component_to_fits,
    write_flux_components_to_fits,
    write_map_result_to_fits,
    write_npred_calibrations_to_fits,
)
from .yaml import (
    read_flux_component_from_yaml,
    read_flux_components_from_yaml,
    read_npred_calibrations_from_yaml,
    write_flux_component_to_yaml,
    write_flux_components_to_yaml,
    write_npred_calibrations_to_yaml,
)

__all__ = [
    "guess_format_from_filename",
    "get_reader",
    "get_writer",
    "IO_FORMATS_MAP_RESULT_READ",
    "IO_FORMATS_MAP_RESULT_WRITE",
    "IO_FORMATS_FLUX_COMPONENT_READ",
    "IO_FORMATS_FLUX_COMPONENT_WRITE",
    "IO_FORMATS_FLUX_COMPONENTS_READ",
    "IO_FORMATS_FLUX_COMPONENTS_WRITE",
    "IO_FORMATS_NPRED_CALIBRATIONS_READ",
    "IO_FORMATS_NPRED_CALIBRATIONS_WRITE",
]


def guess_format_from_filename(filename):
    """{"fits", "yaml", "asdf"} from the file suffix."""
    suffix = Path(filename).suffix
    formats = {".fits": "fits", ".asdf": "asdf", ".yml": "yaml", ".yaml": "yaml"}
    if suffix not in formats:
        raise ValueError(f"Cannot guess format from filename {filename}")
    return formats[suffix]


def _dispatch(filename, format, registry):
    if format is None:
        format = guess_format_from_filename(filename)
    if format not in registry:
        raise ValueError(f"Not a valid format '{format}', choose from {list(registry)}")
    return registry[format]


def get_writer(filename, format, registry):
    return _dispatch(filename, format, registry)


def get_reader(filename, format, registry):
    return _dispatch(filename, format, registry)


IO_FORMATS_MAP_RESULT_READ = {"fits": read_map_result_from_fits, "asdf": read_map_result_from_asdf}
IO_FORMATS_MAP_RESULT_WRITE = {"fits": write_map_result_to_fits, "asdf": write_map_result_to_asdf}

IO_FORMATS_FLUX_COMPONENT_READ = {
    "fits": read_flux_component_from_fits,
    "yaml": read_flux_component_from_yaml,
    "asdf": read_flux_component_from_asdf,
}
IO_FORMATS_FLUX_COMPONENT_WRITE = {
    "yaml": write_flux_component_to_yaml,
    "fits": write_flux_component_to_fits,
    "asdf": write_flux_component_to_asdf,
}

IO_FORMATS_FLUX_COMPONENTS_READ = {
    "fits": read_flux_components_from_fits,
    "asdf": read_flux_components_from_asdf,
    "yaml": read_flux_components_from_yaml,
}
IO_FORMATS_FLUX_COMPONENTS_WRITE = {
    "fits": write_flux_components_to_fits,
    "asdf": write_flux_components_to_asdf,
    "yaml": write_flux_components_to_yaml,
}

IO_FORMATS_NPRED_CALIBRATIONS_READ = {
    "yaml": read_npred_calibrations_from_yaml,
    "fits": read_npred_calibrations_from_fits,
}
IO_FORMATS_NPRED_CALIBRATIONS_WRITE = {
    "yaml": write_npred_calibrations_to_yaml,
    "fits": write_npred_calibrations_to_fits,
}
