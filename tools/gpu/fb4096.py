"""Batched against per-dataset FFT joint step at 4096^2 x 4 datasets and 1024^2 x 8 (JOLIDECO_CONV_METHOD=fft)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
os.environ["JOLIDECO_CONV_METHOD"] = "fft"
import numpy as np, torch
from jolideco_amd import MAPDeconvolver, SpatialFluxComponent, UniformPrior, _hip
from jolideco_amd.data import instrument_like_psf, synthetic_observations
DEV = "cuda:0"
for shape, n_obs in (((4096, 4096), 4), ((1024, 1024), 8)):
    datasets, _, flux_init = synthetic_observations(shape=shape, n_obs=n_obs, seed=2)
    for i, d in enumerate(datasets.values()):
        d["psf"] = instrument_like_psf(i, (33, 33))
    comp = SpatialFluxComponent.from_numpy(flux=flux_init, prior=UniformPrior())
    session = MAPDeconvolver(n_epochs=1, display_progress=False, device=DEV, fit_mode="joint").session(datasets, components=comp)
    assert session.batch_joint
    for rnd in range(2):
        for batched in (1, 0):
            _hip.set_option("JD_FFT_BATCH", batched)
            for _ in range(3): session.epoch()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): session.epoch()
            e1.record(); torch.cuda.synchronize()
            print(shape, n_obs, "batched" if batched else "loop", f"{e0.elapsed_time(e1) / 20 * 1e3:.1f} us/step")
    del session
