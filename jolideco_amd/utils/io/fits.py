"""Jolideco's FITS file layouts on top of the self-contained codec (`_fitsfile`).

Mirrors jolideco/utils/io/fits.py function by function -- same HDU names, header keywords
(`FITS_META`, fits.py:16-37) and table columns -- so that results, flux components and calibrations
written by either package open in the other:

* MAP result (fits.py:421-510): PRIMARY, one IMAGE HDU per component (``NAME``), one per initial
  component (``NAME-INIT``), optional ``CALIBRATIONS`` / ``CALIBRATIONS-INIT`` tables, ``TRACE_LOSS``
  and a one-row ``CONFIG`` table;
* flux components (fits.py:299-333): PRIMARY + one IMAGE HDU per component;
* one flux component (fits.py:335-383): the image in the primary HDU (``EXTNAME = 'PRIMARY'``);
* calibrations (fits.py:386-418): a table with ``name`` and one column per calibration parameter.

Sparse (point source list) components are not on the accelerated path; their table HDUs raise.
"""
import logging

import numpy as np

from ..misc import flatten_dict, unflatten_dict
from ..table import TraceTable
from ._fitsfile import HDU, FitsTable, Header, read_fits, write_fits

log = logging.getLogger(__name__)

SUFFIX_INIT = "-INIT"
META_SEP = "."

# reference keys (fits.py:16-37) ...
FITS_META = {
    "use_log_flux": "LOG_FLUX",
    "upsampling_factor": "UPSAMPLE",
    "frozen": "FROZEN",
    "shape": "SHAPE",
    "prior.type": "PTYPE",
    "prior.stride": "PSTRIDE",
    "prior.cycle_spin": "PSPIN",
    "prior.cycle_spin_subpix": "PSUBSPIN",
    "prior.jitter": "PJITTER",
    "prior.alpha": "PALPHA",
    "prior.beta": "PBETA",
    "prior.width": "PWIDTH",
    "prior.gmm.type": "PGMMTYPE",
    "prior.gmm.stride": "PGMMSTRI",
    "prior.norm.type": "PNORMTYP",
    "prior.norm.max_value": "PNORMMAX",
    "prior.norm.alpha": "PNORMALP",
    "prior.norm.beta": "PNORMBET",
    "prior.patch_norm.type": "PNPTYPE",
    "prior.device": "PDEVICE",
    # ... plus keys only this package writes (ignored by the reference reader)
    "prior.marginalize": "PMARGIN",
    "prior.gmm.n_components": "PGMMNCMP",
    "prior.gmm.n_features": "PGMMNFEA",
}

FITS_META_INVERSE = {value: key for key, value in FITS_META.items()}

# header keywords of a linear celestial WCS that travel with a component image (the reference
# stores an astropy WCS; here it is the plain keyword -> value mapping)
_WCS_KEY = ("WCSAXES", "CRPIX", "CRVAL", "CDELT", "CUNIT", "CTYPE", "CROTA", "PC", "CD", "PV", "LONPOLE", "LATPOLE",
            "RADESYS", "EQUINOX", "MJDREF", "MJD-OBS", "DATE-OBS", "DATEREF")

_NOT_COMPONENTS = ("config", "trace_loss", "calibrations")


def _is_wcs_key(key):
    return any(key == k or (key.startswith(k) and key[len(k):].replace("_", "").isdigit()) for k in _WCS_KEY)


def wcs_from_header(header):
    """The WCS keywords of a header as a dict, or None when there are none."""
    wcs = {key: value for key, value in header.items() if _is_wcs_key(key)}
    return wcs or None


def flux_component_to_image_hdu(flux_component, name):
    """Flux component -> image HDU (reference: fits.py:115-143)."""
    header = Header()
    for key, value in (flux_component.wcs or {}).items():
        header[key] = value
    data = flatten_dict(flux_component.to_dict(), sep=META_SEP)
    for key, value in data.items():
        if key not in FITS_META:
            raise KeyError(f"no FITS keyword is defined for the component setting {key!r}")
        if key == "upsampling_factor":
            value = int(value or 1)
        header[FITS_META[key]] = value
    return HDU(data=flux_component.flux_upsampled_numpy, header=header, name=name.upper(), kind="image")


def flux_component_from_image_hdu(hdu):
    """Image HDU -> flux component (reference: fits.py:146-172)."""
    from ...models import SpatialFluxComponent

    data = {"wcs": wcs_from_header(hdu.header), "flux_upsampled": hdu.data}
    for fits_key, key in FITS_META_INVERSE.items():
        value = hdu.header.get(fits_key, None)
        if value is not None:
            data[key] = value
    return SpatialFluxComponent.from_dict(unflatten_dict(data, sep=META_SEP))


def sparse_flux_component_from_table_hdu(hdu):
    raise NotImplementedError(
        f"HDU {hdu.name!r} holds a sparse (point source list) flux component (reference fits.py:87-112); "
        "sparse components are not implemented in jolideco_amd"
    )


def flux_components_to_hdulist(flux_components, name_suffix=""):
    """Reference: fits.py:175-204."""
    hdus = []
    for name, component in flux_components.items():
        if getattr(component, "is_sparse", False):
            raise NotImplementedError("sparse flux components are not implemented in jolideco_amd")
        hdus.append(flux_component_to_image_hdu(component, name=name + name_suffix))
    return hdus


def flux_components_from_hdulist(hdulist):
    """Reference: fits.py:207-238: every image extension that is not a bookkeeping table."""
    from ...models import FluxComponents

    components = FluxComponents()
    for hdu in hdulist:
        name = hdu.name.replace(SUFFIX_INIT, "").lower()
        if name in _NOT_COMPONENTS:
            continue
        if hdu.kind == "image":
            components[name] = flux_component_from_image_hdu(hdu)
        elif hdu.kind == "bintable":
            components[name] = sparse_flux_component_from_table_hdu(hdu)
    return components


def npred_calibrations_to_table(npred_calibrations):
    """Reference: fits.py:241-264: one row per dataset, ``name`` first."""
    rows = []
    for name, value in npred_calibrations.to_dict().items():
        row = {"name": name}
        row.update(value)
        rows.append(row)
    return FitsTable.from_rows(rows)


def npred_calibrations_from_table(table):
    """Reference: fits.py:267-296."""
    from ...models import NPredCalibrations

    data = {}
    for row in table:
        row = dict(row)
        name = row.pop("name")
        data[name.decode("utf-8") if isinstance(name, bytes) else str(name)] = row
    return NPredCalibrations.from_dict(data)


def write_flux_components_to_fits(flux_components, filename, overwrite):
    log.info(f"writing {filename}")
    write_fits(filename, [HDU(kind="primary")] + flux_components_to_hdulist(flux_components), overwrite=overwrite)


def read_flux_components_from_fits(filename):
    return flux_components_from_hdulist(read_fits(filename))


def write_flux_component_to_fits(flux_component, filename, overwrite):
    """The component image goes into the primary HDU (reference: fits.py:335-359)."""
    if getattr(flux_component, "is_sparse", False):
        raise NotImplementedError("sparse flux components are not implemented in jolideco_amd")
    hdu = flux_component_to_image_hdu(flux_component, name="primary")
    hdu.kind = "primary"
    log.info(f"writing {filename}")
    write_fits(filename, [hdu], overwrite=overwrite)


def read_flux_component_from_fits(filename, hdu_name=0):
    """Reference: fits.py:362-383."""
    hdulist = read_fits(filename)
    if isinstance(hdu_name, str):
        matches = [hdu for hdu in hdulist if hdu.name.upper() == hdu_name.upper()]
        if not matches:
            raise KeyError(f"Extension {hdu_name!r} not found.")
        hdu = matches[0]
    else:
        hdu = hdulist[hdu_name]
    if hdu.is_image:
        return flux_component_from_image_hdu(hdu)
    return sparse_flux_component_from_table_hdu(hdu)


def read_npred_calibrations_from_fits(filename):
    log.info(f"Reading {filename}")
    tables = [hdu for hdu in read_fits(filename) if hdu.kind == "bintable"]
    if not tables:
        raise ValueError(f"{filename} holds no table")
    return npred_calibrations_from_table(tables[0].data)


def write_npred_calibrations_to_fits(npred_calibrations, filename, overwrite):
    write_fits(filename, [HDU(kind="primary"), HDU(npred_calibrations_to_table(npred_calibrations))],
               overwrite=overwrite)


def trace_to_table(trace_loss):
    """Loss trace -> table: float64 columns and the ``filename`` string column (loss.py:201-210)."""
    table = FitsTable()
    for name in trace_loss.colnames:
        values = trace_loss[name]
        if name == "filename":
            table[name] = np.array([str(v) for v in values], dtype=str) if len(values) else np.zeros(0, dtype="U1")
        else:
            table[name] = np.asarray(values, dtype=np.float64)
    return table


def trace_from_table(table):
    trace = TraceTable(names=table.colnames)
    for row in table:
        trace.add_row(row)
    return trace


def config_to_table(config):
    """One-row table, one column per setting (reference: core.py:425-433)."""
    table = FitsTable()
    for key, value in config.items():
        table[key] = [value if isinstance(value, (bool, int, float, str, np.generic)) else str(value)]
    return table


def write_map_result_to_fits(result, filename, overwrite):
    """Reference: fits.py:421-459.  One deliberate difference: the ``-INIT`` HDUs hold
    ``result.components_init``; the reference passes ``result.components`` there a second time
    (fits.py:438-439), so its ``-INIT`` images are copies of the final flux."""
    hdulist = [HDU(kind="primary")]
    hdulist.extend(flux_components_to_hdulist(result.components))
    if result.components_init is not None:
        hdulist.extend(flux_components_to_hdulist(result.components_init, name_suffix=SUFFIX_INIT))
    if result.calibrations:
        hdulist.append(HDU(npred_calibrations_to_table(result.calibrations), name="CALIBRATIONS"))
        if result.calibrations_init:
            table = npred_calibrations_to_table(result.calibrations_init)
            hdulist.append(HDU(table, name="CALIBRATIONS" + SUFFIX_INIT))
    hdulist.append(HDU(trace_to_table(result.trace_loss), name="TRACE_LOSS"))
    hdulist.append(HDU(config_to_table(result.config), name="CONFIG"))
    log.info(f"writing {filename}")
    write_fits(filename, hdulist, overwrite=overwrite)


def read_map_result_from_fits(filename):
    """Reference: fits.py:462-510."""
    from ...core import MAPDeconvolverResult

    log.info(f"Reading {filename}")
    hdulist = read_fits(filename)
    by_name = {hdu.name.upper(): hdu for hdu in hdulist}
    config = dict(by_name["CONFIG"].data[0])
    trace_loss = trace_from_table(by_name["TRACE_LOSS"].data)
    components = flux_components_from_hdulist([hdu for hdu in hdulist if SUFFIX_INIT not in hdu.name])
    components_init = flux_components_from_hdulist([hdu for hdu in hdulist if SUFFIX_INIT in hdu.name])
    calibrations = calibrations_init = None
    if "CALIBRATIONS" in by_name:
        calibrations = npred_calibrations_from_table(by_name["CALIBRATIONS"].data)
    if "CALIBRATIONS" + SUFFIX_INIT in by_name:
        calibrations_init = npred_calibrations_from_table(by_name["CALIBRATIONS" + SUFFIX_INIT].data)
    return MAPDeconvolverResult(
        config=config,
        components=components,
        components_init=components_init,
        calibrations=calibrations,
        calibrations_init=calibrations_init,
        trace_loss=trace_loss,
    )
