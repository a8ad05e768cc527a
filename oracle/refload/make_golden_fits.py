"""Generate tests/golden/io/*.fits from the LIVE reference's FITS writers (build container only).

Two steps, because no interpreter in the image has both PyTorch and astropy:

  python3    oracle/refload/make_golden_fits.py           # this file, python3.10 + PyTorch
      runs a small reference fit and calls the reference's own writers
      (MAPDeconvolverResult.write, FluxComponents.write, SpatialFluxComponent.write,
      NPredCalibrations.write, jolideco/utils/io/fits.py); the loader's recording stand-in for
      astropy.io.fits captures the HDUs they build into tests/golden/io/<case>.hdus.npz
  /opt/conda/bin/python3.9 oracle/refload/hdus_to_fits.py  # real astropy 4.3
      replays every <case>.hdus.npz through astropy.io.fits / astropy.table -> <case>.fits

The .hdus.npz files double as the expected values of the read-back tests (tests/test_io_fits.py).
"""
import sys
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
REPO = HERE.parent.parent
sys.path.insert(0, str(HERE))
sys.path.insert(0, str(REPO))

from load_ref import load_reference  # noqa: E402

load_reference()

from jolideco.core import MAPDeconvolver  # noqa: E402
from jolideco.data import point_source_gauss_psf  # noqa: E402
from jolideco.models import (  # noqa: E402
    FluxComponents,
    NPredCalibration,
    NPredCalibrations,
    SpatialFluxComponent,
)
from jolideco.priors import ExponentialPrior, InverseGammaPrior, UniformPrior  # noqa: E402

OUT = REPO / "tests" / "golden" / "io"
OUT.mkdir(parents=True, exist_ok=True)


def main():
    rs = np.random.RandomState(20240607)
    datasets = {
        f"obs-{i}": point_source_gauss_psf(shape=(32, 32), sigma_psf=1.5 + 0.5 * i, random_state=rs) for i in range(2)
    }
    calibrations = NPredCalibrations()
    calibrations["obs-0"] = NPredCalibration(shift_x=0.3, shift_y=-0.2, background_norm=1.1)
    calibrations["obs-1"] = NPredCalibration(shift_x=-0.15, shift_y=0.25, background_norm=0.9, frozen=True)

    flux_init = rs.gamma(30, size=(32, 32))
    component = SpatialFluxComponent.from_numpy(flux=flux_init, prior=InverseGammaPrior(alpha=10.0, beta=1.5))
    # inputs of the fit, so that the GPU tests can run the same fit and compare the FILE they write
    inputs = {"flux_init": flux_init}
    for name, d in datasets.items():
        for key in ("counts", "psf", "exposure", "background"):
            inputs[f"data/{name}/{key}"] = d[key]
    np.savez_compressed(OUT / "result_inputs.npz", **inputs)
    torch.manual_seed(0)
    deconvolver = MAPDeconvolver(n_epochs=4, display_progress=False)
    result = deconvolver.run(datasets=datasets, components=component, calibrations=calibrations)
    result.write(OUT / "result.hdus.npz", format="fits", overwrite=True)

    components = FluxComponents()
    components["flux-uniform"] = SpatialFluxComponent(
        flux_upsampled=torch.from_numpy(rs.gamma(5, size=(1, 1, 16, 24)).astype(np.float32)),
        upsampling_factor=2, use_log_flux=False, frozen=False, prior=UniformPrior(),
    )
    components["flux-point"] = SpatialFluxComponent(
        flux_upsampled=torch.from_numpy(rs.gamma(5, size=(1, 1, 16, 24)).astype(np.float32)),
        upsampling_factor=2, use_log_flux=True, frozen=True, prior=ExponentialPrior(alpha=3.0),
    )
    components.write(OUT / "components.hdus.npz", format="fits", overwrite=True)
    components["flux-point"].write(OUT / "component.hdus.npz", format="fits", overwrite=True)
    calibrations.write(OUT / "calibrations.hdus.npz", format="fits", overwrite=True)

    for path in sorted(OUT.glob("*.hdus.npz")):
        layout = str(np.load(path)["layout"])
        print(path.name, layout[:300])


if __name__ == "__main__":
    main()
