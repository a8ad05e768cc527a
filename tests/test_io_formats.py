"""CPU: the file formats either side of the hot path: Jolideco's FITS layouts
for results / flux components / calibrations / GMM libraries, and the YAML side.

Pinned three ways:
  * tests/golden/io/*.fits were written by the REAL astropy from the HDUs the REFERENCE's own writers
    built (oracle/refload/make_golden_fits.py + hdus_to_fits.py); jolideco_amd must read them and
    recover the recorded values (tests/golden/io/*.hdus.npz) exactly;
  * files written by jolideco_amd must have the same structure (HDU names and kinds, header keywords and
    values, table columns and TFORMs) as those golden files, byte-for-byte in the data;
  * where an interpreter with astropy exists (/opt/conda/bin/python3.9 in the build image) astropy itself
    opens and verifies the files jolideco_amd writes.
The round-trip cases mirror the reference's own I/O tests (models/tests/test_core.py:126-214,
models/tests/test_npred.py, tests/test_core.py:82-91).
"""
import json
import os
import subprocess
from pathlib import Path

import numpy as np
import pytest
import torch

from jolideco_amd import (
    ExponentialPrior,
    FluxComponents,
    GaussianMixtureModel,
    GMMPatchPrior,
    InverseGammaPrior,
    MAPDeconvolverResult,
    NPredCalibration,
    NPredCalibrations,
    SpatialFluxComponent,
    UniformPrior,
)
from jolideco_amd.priors import PRIOR_REGISTRY
from jolideco_amd.priors.patches import GaussianMixtureModelMeta
from jolideco_amd.utils.io import guess_format_from_filename
from jolideco_amd.utils.io._fitsfile import HDU, FitsTable, Header, read_fits, write_fits
from jolideco_amd.utils.table import TraceTable
from oracle import cpu_ref

REPO = Path(__file__).resolve().parent.parent
GOLDEN_IO = REPO / "tests" / "golden" / "io"
ASTROPY_PYTHON = Path("/opt/conda/bin/python3.9")
CHECKER = REPO / "oracle" / "refload" / "hdus_to_fits.py"


def recorded(case):
    data = np.load(GOLDEN_IO / f"{case}.hdus.npz")
    return json.loads(str(data["layout"])), data


@pytest.fixture()
def gmm_library(tmp_path, monkeypatch):
    """A user's GMM library directory with one synthetic model registered as "zoran-weiss"."""
    means, covs, weights = cpu_ref.synthetic_gmm(5, 64, seed=11, zero_means=False)
    # the model keeps fp32 arrays (as the reference does): start from fp32-representable values so that
    # the file holds exactly what the model was built from
    means, covs, weights = (a.astype(np.float32).astype(np.float64) for a in (means, covs, weights))
    gmm = GaussianMixtureModel.from_numpy(means, covs, weights, meta=GaussianMixtureModelMeta(stride=4))
    gmm.write(tmp_path / "zw.fits")
    index = {"zoran-weiss": {"filename": "$JOLIDECO_GMM_LIBRARY/zw.fits", "format": "table"}}
    (tmp_path / "jolideco-gmm-library-index.json").write_text(json.dumps(index))
    monkeypatch.setenv("JOLIDECO_GMM_LIBRARY", str(tmp_path))
    return gmm


# ------------------------------------------------------------------------------------------ the codec
def test_fits_codec_round_trip(tmp_path):
    header = Header()
    header["LOG_FLUX"] = True
    header["UPSAMPLE"] = 2
    header["PTYPE"] = "gmm-patches"
    header["PALPHA"] = 10.5
    header["TINY"] = 1e-30
    header["QUOTE"] = "it's"
    header["LONGSTR"] = "x" * 150 + "'end"
    image = np.arange(12, dtype=np.float32).reshape(3, 4)
    columns = {
        "total": np.array([1.5, 2.5]),
        "filename": np.array(["", ""]),
        "n": np.array([1, -2]),
        "ok": np.array([True, False]),
        "cov": np.arange(18, dtype=np.float64).reshape(2, 3, 3),
        "vec": np.arange(6, dtype=np.float32).reshape(2, 3),
        "name": np.array(["obs-1", "observation-22"]),
        "small": np.array([3, 4], dtype=np.int16),
        "byte": np.array([3, 255], dtype=np.uint8),
    }
    path = tmp_path / "a.fits"
    write_fits(path, [HDU(kind="primary"), HDU(image, header, name="flux"), HDU(FitsTable(columns), name="TAB")])
    assert path.stat().st_size % 2880 == 0
    with pytest.raises(OSError):
        write_fits(path, [HDU(kind="primary")])
    primary, img, tab = read_fits(path)
    assert (primary.kind, img.kind, tab.kind) == ("primary", "image", "bintable")
    assert (primary.name, img.name, tab.name) == ("PRIMARY", "FLUX", "TAB")
    assert img.data.dtype == np.float32 and np.array_equal(img.data, image)
    for key, value in header.items():
        assert img.header[key] == value and type(img.header[key]) is type(value)
    for name, values in columns.items():
        got = tab.data[name]
        assert got.shape == values.shape and np.array_equal(got, values), name
        assert got.dtype.kind == values.dtype.kind and (got.dtype.kind == "U" or got.dtype == values.dtype), name
    assert tab.data[1]["name"] == "observation-22" and tab.data[1]["ok"] is False
    assert tab.header["TFORM5"] == "9D" and tab.header["TDIM5"] == "(3,3)"

    # integer images, 3-d cubes, an image in the primary HDU
    cube = np.arange(24, dtype=np.int32).reshape(2, 3, 4)
    write_fits(tmp_path / "b.fits", [HDU(cube, kind="primary"), HDU(cube.astype(np.float64)[0])])
    first, second = read_fits(tmp_path / "b.fits")
    assert first.kind == "primary" and first.data.dtype == np.int32 and np.array_equal(first.data, cube)
    assert second.data.dtype == np.float64 and second.header["NAXIS1"] == 4 and second.header["NAXIS2"] == 3

    # a table without rows (the loss trace of the first checkpoint)
    empty = FitsTable({"total": np.zeros(0), "filename": np.zeros(0, dtype="U1"), "ok": np.zeros(0, bool)})
    write_fits(tmp_path / "e.fits", [HDU(empty, name="TRACE_LOSS")])
    back = read_fits(tmp_path / "e.fits")[1].data
    assert back.colnames == ["total", "filename", "ok"] and len(back) == 0 and back["total"].dtype == np.float64

    # a table first -> an empty primary HDU is inserted (astropy does the same)
    write_fits(tmp_path / "c.fits", [HDU(FitsTable({"a": [1.0]}))])
    assert [h.kind for h in read_fits(tmp_path / "c.fits")] == ["primary", "bintable"]
    with pytest.raises(ValueError):
        (tmp_path / "junk.fits").write_bytes(b"not a fits file".ljust(2880))
        read_fits(tmp_path / "junk.fits")
    with pytest.raises(ValueError):
        Header([("TOOLONGKEY", 1, "")]) and write_fits(tmp_path / "d.fits", [HDU(image, Header([("TOOLONGKEY", 1, "")]))])


# ------------------------------------------------------------------- golden files (reference + astropy)
def test_read_reference_result_fits():
    layout, data = recorded("result")
    result = MAPDeconvolverResult.read(GOLDEN_IO / "result.fits")
    names = [entry["name"] for entry in layout]
    assert names == ["", "FLUX", "FLUX-INIT", "CALIBRATIONS", "CALIBRATIONS-INIT", "TRACE_LOSS", "CONFIG"]

    assert list(result.components) == ["flux"] and list(result.components_init) == ["flux"]
    component = result.components["flux"]
    assert np.array_equal(component.flux_upsampled_numpy, data["hdu1/data"])
    assert np.array_equal(result.components_init["flux"].flux_upsampled_numpy, data["hdu2/data"])
    assert np.array_equal(result.flux_total, data["hdu1/data"])
    assert component.use_log_flux is True and component.frozen is False and component.upsampling_factor == 1
    assert isinstance(component.prior, InverseGammaPrior)
    assert component.prior.alpha == 10.0 and component.prior.beta == 1.5

    trace = result.trace_loss
    columns = layout[5]["columns"]
    assert trace.colnames == columns and columns[-1] == "filename" and len(trace) == 4
    for name in columns[:-1]:
        assert np.array_equal(trace[name], data[f"hdu5/{name}"]), name
    assert list(trace["filename"]) == ["", "", "", ""]
    assert trace[-1]["total"] == data["hdu5/total"][-1]

    for key in layout[6]["columns"]:
        expected = data[f"hdu6/{key}"][0]
        got = result.config[key]
        assert got == (str(expected) if expected.dtype.kind == "U" else expected.item()), key
    assert result.config["n_epochs"] == 4 and result.config["checkpoint_path"] == "None"

    for hdu_index, calibrations in ((3, result.calibrations), (4, result.calibrations_init)):
        assert list(calibrations) == ["obs-0", "obs-1"]
        for row, (name, model) in enumerate(calibrations.items()):
            values = model.to_dict()
            for key in ("shift_x", "shift_y", "background_norm", "psf_scale", "weight"):
                assert values[key] == pytest.approx(float(data[f"hdu{hdu_index}/{key}"][row]), rel=1e-7), (name, key)
            assert values["frozen"] == bool(data[f"hdu{hdu_index}/frozen"][row])
    assert result.calibrations["obs-1"].frozen is True
    # the fit moved the free calibration, the initial copy kept the start values
    assert result.calibrations_init["obs-0"].to_dict()["shift_x"] == pytest.approx(0.3, rel=1e-6)
    assert result.calibrations["obs-0"].to_dict()["shift_x"] != result.calibrations_init["obs-0"].to_dict()["shift_x"]


def test_read_reference_components_and_calibrations_fits():
    layout, data = recorded("components")
    components = FluxComponents.read(GOLDEN_IO / "components.fits")
    assert list(components) == ["flux-uniform", "flux-point"]
    assert np.array_equal(components["flux-uniform"].flux_upsampled_numpy, data["hdu1/data"])
    assert np.array_equal(components["flux-point"].flux_upsampled_numpy, data["hdu2/data"])
    uniform, point = components["flux-uniform"], components["flux-point"]
    assert (uniform.use_log_flux, uniform.frozen, uniform.upsampling_factor) == (False, False, 2)
    assert (point.use_log_flux, point.frozen, point.upsampling_factor) == (True, True, 2)
    assert isinstance(uniform.prior, UniformPrior) and isinstance(point.prior, ExponentialPrior)
    assert point.prior.alpha == 3.0
    assert point.flux_numpy.shape == (8, 12)

    _, data = recorded("component")
    component = SpatialFluxComponent.read(GOLDEN_IO / "component.fits")
    assert np.array_equal(component.flux_upsampled_numpy, data["hdu0/data"])
    assert isinstance(component.prior, ExponentialPrior) and component.frozen and component.upsampling_factor == 2

    _, data = recorded("calibrations")
    calibrations = NPredCalibrations.read(GOLDEN_IO / "calibrations.fits")
    assert list(calibrations) == [str(n) for n in data["hdu1/name"]]
    for row, model in enumerate(calibrations.values()):
        assert model.to_dict()["shift_y"] == pytest.approx(float(data["hdu1/shift_y"][row]), rel=1e-7)


def _structure(path):
    """What must agree between a file of ours and the reference's: HDU kinds / names, the header keywords
    a user of the file sees (and their values), table column names and TFORMs."""
    out = []
    for hdu in read_fits(path):
        entry = {"kind": hdu.kind, "name": hdu.name}
        skip = ("EXTNAME",)
        if hdu.is_image:
            entry["header"] = {k: v for k, v in hdu.header.items()
                               if k not in skip and not k.startswith(("NAXIS", "BITPIX", "PCOUNT", "GCOUNT"))
                               and k not in ("SIMPLE", "XTENSION", "EXTEND")}
            entry["bitpix"] = hdu.header["BITPIX"]
            entry["shape"] = None if hdu.data is None else hdu.data.shape
        else:
            n = hdu.header["TFIELDS"]
            entry["columns"] = [(hdu.header[f"TTYPE{i}"], hdu.header[f"TFORM{i}"].strip()) for i in range(1, n + 1)]
            entry["rows"] = len(hdu.data)
        out.append(entry)
    return out


def test_written_files_match_reference_structure(tmp_path):
    """Read a golden file, write it back with our writers: same structure, same data."""
    result = MAPDeconvolverResult.read(GOLDEN_IO / "result.fits")
    result.write(tmp_path / "result.fits")
    mine, golden = _structure(tmp_path / "result.fits"), _structure(GOLDEN_IO / "result.fits")
    assert mine == golden
    for ours, theirs in zip(read_fits(tmp_path / "result.fits"), read_fits(GOLDEN_IO / "result.fits")):
        if ours.is_image:
            assert (ours.data is None and theirs.data is None) or np.array_equal(ours.data, theirs.data)
        else:
            for name in theirs.data.colnames:
                a, b = ours.data[name], theirs.data[name]
                if a.dtype.kind == "f":  # calibrations pass through fp32 parameters
                    assert np.allclose(a, b, rtol=1e-7, atol=0), name
                else:
                    assert np.array_equal(a, b), name

    FluxComponents.read(GOLDEN_IO / "components.fits").write(tmp_path / "components.fits")
    assert _structure(tmp_path / "components.fits") == _structure(GOLDEN_IO / "components.fits")
    SpatialFluxComponent.read(GOLDEN_IO / "component.fits").write(tmp_path / "component.fits")
    assert _structure(tmp_path / "component.fits") == _structure(GOLDEN_IO / "component.fits")
    NPredCalibrations.read(GOLDEN_IO / "calibrations.fits").write(tmp_path / "calibrations.fits")
    assert _structure(tmp_path / "calibrations.fits") == _structure(GOLDEN_IO / "calibrations.fits")


@pytest.mark.skipif(not ASTROPY_PYTHON.exists(), reason="no interpreter with astropy on this machine")
def test_astropy_opens_what_we_write(tmp_path):
    """astropy 4.3 verifies our files against the standard and sees what it sees in the reference's."""

    def report(path):
        out = subprocess.run([str(ASTROPY_PYTHON), str(CHECKER), "--check", str(path)], capture_output=True,
                             text=True, timeout=300, env={k: v for k, v in os.environ.items() if k != "PYTHONPATH"})
        if out.returncode != 0 and "No module named" in out.stderr:
            pytest.skip(f"astropy interpreter is not usable: {out.stderr.strip().splitlines()[-1]}")
        assert out.returncode == 0, out.stderr[-2000:]
        return json.loads(out.stdout.strip().splitlines()[-1])

    result = MAPDeconvolverResult.read(GOLDEN_IO / "result.fits")
    result.write(tmp_path / "result.fits")
    mine, golden = report(tmp_path / "result.fits"), report(GOLDEN_IO / "result.fits")
    assert [h["name"] for h in mine] == [h["name"] for h in golden]
    assert [h["type"] for h in mine] == [h["type"] for h in golden]
    for ours, theirs in zip(mine, golden):
        if "columns" in theirs:
            assert ours["columns"] == theirs["columns"], ours["name"]
            if ours["name"] in ("TRACE_LOSS", "CONFIG"):
                assert ours["rows"] == theirs["rows"]
        else:
            assert ours.get("shape") == theirs.get("shape") and ours.get("dtype") == theirs.get("dtype")
            assert ours.get("sum") == theirs.get("sum")
            user_keys = [k for k in theirs["header"] if k in ("LOG_FLUX", "UPSAMPLE", "FROZEN", "PTYPE", "PALPHA", "PBETA")]
            assert {k: ours["header"][k] for k in user_keys} == {k: theirs["header"][k] for k in user_keys}

    # a GMM library table with (K, D, D) covariances: TDIM round trip through astropy
    means, covs, weights = cpu_ref.synthetic_gmm(3, 64, seed=2, zero_means=False)
    GaussianMixtureModel.from_numpy(means, covs, weights).write(tmp_path / "gmm.fits")
    gmm = report(tmp_path / "gmm.fits")
    assert gmm[1]["columns"] == {"means": [">f8", [3, 64]], "weights": [">f8", [3]],
                                 "covariances": [">f8", [3, 64, 64]]}
    assert gmm[1]["header"]["PNPTYPE"] == "subtract-mean"


# ------------------------------------------------------------- the reference's own round-trip tests
@pytest.mark.parametrize("prior_class", list(PRIOR_REGISTRY.values()))
@pytest.mark.parametrize("format", ["fits", "yaml"])
def test_flux_component_io(prior_class, format, tmp_path, gmm_library):
    """models/tests/test_core.py:126-149"""
    component = SpatialFluxComponent(
        flux_upsampled=torch.ones((1, 1, 32, 32)), upsampling_factor=2, use_log_flux=False, frozen=True,
        prior=prior_class(),
    )
    filename = tmp_path / f"test.{format}"
    component.write(filename=filename, format=format)
    component_new = SpatialFluxComponent.read(filename=filename, format=format)
    assert component.shape == component_new.shape
    assert component.upsampling_factor == component_new.upsampling_factor
    assert component.use_log_flux == component_new.use_log_flux
    assert component_new.frozen is True
    assert isinstance(component_new.prior, prior_class)
    if prior_class is GMMPatchPrior:
        assert component_new.prior.gmm.registry_name == "zoran-weiss" and component_new.prior.stride == 4
        assert np.array_equal(component_new.prior.gmm.covariances_numpy, gmm_library.covariances_numpy)
    with pytest.raises(OSError):
        component.write(filename=filename, format=format)
    component.write(filename=filename, format=format, overwrite=True)


@pytest.mark.parametrize("prior_class", list(PRIOR_REGISTRY.values()))
@pytest.mark.parametrize("format", ["fits", "yaml"])
def test_flux_components_io(prior_class, format, tmp_path, gmm_library):
    """models/tests/test_core.py:152-181"""
    components = FluxComponents()
    flux_init = torch.ones((1, 1, 32, 32))
    components["flux-uniform"] = SpatialFluxComponent(
        flux_upsampled=flux_init, upsampling_factor=2, use_log_flux=False, frozen=False, prior=UniformPrior()
    )
    components["flux-point"] = SpatialFluxComponent(
        flux_upsampled=3 * flux_init, upsampling_factor=2, use_log_flux=False, frozen=False, prior=prior_class()
    )
    filename = tmp_path / f"test.{format}"
    components.write(filename=filename, format=format)
    components_new = FluxComponents.read(filename=filename, format=format)
    assert list(components_new) == ["flux-uniform", "flux-point"]
    assert isinstance(components_new["flux-point"].prior, prior_class)
    assert np.array_equal(components_new["flux-point"].flux_upsampled_numpy, np.full((32, 32), 3.0, np.float32))
    if format == "yaml":  # settings in the YAML file, images in companion FITS files next to it
        text = filename.read_text()
        assert "flux-point:" in text and "flux_upsampled: " in text and "!!python" not in text
        assert (tmp_path / "flux-point-data.fits").exists() and (tmp_path / "flux-uniform-data.fits").exists()


@pytest.mark.parametrize("format", ["yaml", "fits"])
def test_npred_calibrations_io(format, tmp_path):
    """models/tests/test_npred.py:6-37"""
    calibrations = NPredCalibrations()
    calibrations["dataset-1"] = NPredCalibration(shift_x=0.1, shift_y=0.1, background_norm=0.9)
    calibrations["dataset-2"] = NPredCalibration(shift_x=-0.2, shift_y=0.23, background_norm=1.05, frozen=True)
    filename = tmp_path / f"test.{format}"
    calibrations.write(filename=filename, format=format)
    calibrations_new = NPredCalibrations.read(filename=filename, format=format)
    for name in ("dataset-1", "dataset-2"):
        data, data_new = calibrations.to_dict()[name], calibrations_new.to_dict()[name]
        assert data == data_new
    assert calibrations_new["dataset-2"].frozen is True and calibrations_new["dataset-1"].frozen is False


def make_result(gmm=None):
    components = FluxComponents()
    components["flux"] = SpatialFluxComponent.from_numpy(
        np.random.RandomState(3).gamma(5, size=(16, 16)), prior=GMMPatchPrior(gmm=gmm) if gmm else UniformPrior()
    )
    names = ["total", "datasets-total", "priors-total", "prior-flux", "dataset-obs-0", "filename"]
    trace = TraceTable(names=names)
    for i in range(3):
        trace.add_row({"total": 3.0 - i, "datasets-total": 2.0 - i, "priors-total": 1.0, "prior-flux": 1.0,
                       "dataset-obs-0": 2.0 - i, "filename": f"checkpoint-epoch-{i}.fits"})
    config = {"n_epochs": 100, "beta": 1, "learning_rate": 0.1, "compute_error": False, "stop_early": False,
              "stop_early_n_average": 10, "display_progress": True, "device": "cuda:0", "optimizer_type": "adam",
              "checkpoint_path": "None", "fit_mode": "sequential"}
    return MAPDeconvolverResult(config=config, components=components, components_init=components, trace_loss=trace)


@pytest.mark.parametrize("format", ["fits", "npz"])
def test_map_deconvolver_result_io(format, tmp_path):
    """tests/test_core.py:82-91"""
    result = make_result()
    filename = tmp_path / f"result.{format}"
    result.write(filename, format=format)
    new = MAPDeconvolverResult.read(filename=filename, format=format)
    assert np.array_equal(new.flux_total, result.flux_total)
    assert np.array_equal(new.trace_loss["total"], [3.0, 2.0, 1.0])
    if format == "fits":
        assert new.config == result.config
        assert new.config["n_epochs"] == 100 and new.config["learning_rate"] == 0.1
        assert list(new.trace_loss["filename"]) == [f"checkpoint-epoch-{i}.fits" for i in range(3)]
        assert new.calibrations is None and list(new.components_init) == ["flux"]
    with pytest.raises(OSError):
        result.write(filename, format=format)


def test_result_with_unnamed_gmm_reads_back_with_a_warning(tmp_path, caplog):
    """A GMM built from explicit arrays has no library name: the file records its size, reading it back
    keeps the flux and falls back to a uniform prior with a warning (the reference cannot write it at
    all, gmm.py:458-471)."""
    means, covs, weights = cpu_ref.synthetic_gmm(4, 64, seed=5)
    gmm = GaussianMixtureModel.from_numpy(means, covs, weights, meta=GaussianMixtureModelMeta(stride=4))
    result = make_result(gmm=gmm)
    result.write(tmp_path / "result.fits")
    header = read_fits(tmp_path / "result.fits")[1].header
    assert header["PTYPE"] == "gmm-patches" and header["PGMMTYPE"] == "custom" and header["PGMMNCMP"] == 4
    assert header["PSTRIDE"] == 4 and header["PNPTYPE"] == "subtract-mean" and header["PNORMTYP"] == "identity"
    with caplog.at_level("WARNING"):
        new = MAPDeconvolverResult.read(tmp_path / "result.fits")
    assert isinstance(new.components["flux"].prior, UniformPrior)
    assert "uniform prior" in caplog.text
    assert np.array_equal(new.flux_total, result.flux_total)


def test_formats_and_errors(tmp_path):
    assert guess_format_from_filename("a/b.fits") == "fits"
    assert guess_format_from_filename("b.yml") == "yaml" and guess_format_from_filename("b.yaml") == "yaml"
    assert guess_format_from_filename("b.asdf") == "asdf"
    with pytest.raises(ValueError):
        guess_format_from_filename("b.txt")
    result = make_result()
    with pytest.raises(ValueError, match="Not a valid format"):
        result.write(tmp_path / "result.fits", format="hdf5")
    with pytest.raises(ValueError, match="Not a valid format"):
        NPredCalibrations().write(tmp_path / "c.asdf")


# ----------------------------------------------------------------------------------- GMM library files
def test_gmm_table_and_registry(tmp_path, gmm_library):
    """gmm.py:301-391: by name through the library index, and from a table file directly."""
    gmm = GaussianMixtureModel.from_registry("zoran-weiss")
    assert gmm.registry_name == "zoran-weiss" and gmm.to_dict() == {"type": "zoran-weiss"}
    assert gmm.n_components == 5 and gmm.n_features == 64 and gmm.meta.stride == 4
    for name in ("means_numpy", "covariances_numpy", "weights_numpy", "precisions_cholesky_numpy"):
        assert np.array_equal(getattr(gmm, name), getattr(gmm_library, name)), name
    direct = GaussianMixtureModel.read(tmp_path / "zw.fits", format="table")
    assert np.array_equal(direct.covariances_numpy, gmm.covariances_numpy) and direct.registry_name is None
    assert GMMPatchPrior().gmm.registry_name == "zoran-weiss"  # the default prior of the reference
    with pytest.raises(ValueError, match="Not a supported GMM"):
        GaussianMixtureModel.from_registry("gleam")
    with pytest.raises(ValueError, match="Not a supported format"):
        GaussianMixtureModel.read(tmp_path / "zw.fits", format="hdf5")


def test_gmm_epll_matlab(tmp_path):
    """The Zoran & Weiss EPLL .mat layout (gmm.py:360-381): GS.means (D, K), GS.covs (D, D, K),
    GS.mixweights (K, 1)."""
    import scipy.io as sio

    means, covs, weights = cpu_ref.synthetic_gmm(4, 64, seed=9, zero_means=False)
    record = {"means": means.T, "covs": covs.T, "mixweights": weights[:, None]}
    sio.savemat(tmp_path / "GSModel_8x8_200_2M_noDC_zeromean.mat", {"GS": record})
    gmm = GaussianMixtureModel.read(tmp_path / "GSModel_8x8_200_2M_noDC_zeromean.mat", format="epll-matlab")
    ref = GaussianMixtureModel.from_numpy(means, np.transpose(covs, (0, 2, 1)), weights)
    assert gmm.meta.stride == 4 and gmm.n_components == 4
    assert np.array_equal(gmm.means_numpy, ref.means_numpy)
    assert np.array_equal(gmm.covariances_numpy, ref.covariances_numpy)
    assert np.array_equal(gmm.precisions_cholesky_numpy, ref.precisions_cholesky_numpy)

    sio.savemat(tmp_path / "gmm16.mat", {"GMM": {"covs": np.tile(np.eye(256)[:, :, None], (1, 1, 200)),
                                                  "mixweights": np.full((200, 1), 1 / 200)}})
    gmm16 = GaussianMixtureModel.read(tmp_path / "gmm16.mat", format="epll-matlab-16x16")
    assert gmm16.n_features == 256 and gmm16.meta.stride == 8 and gmm16.patch_shape == (16, 16)
