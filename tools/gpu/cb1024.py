"""Batched against per-dataset CALIBRATED + up-sampled FFT joint step (c6 shape) at several sizes."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
from jolideco_amd import MAPDeconvolver, NPredCalibration, NPredCalibrations, SpatialFluxComponent, UniformPrior, _hip
from jolideco_amd.data import instrument_observations
DEV = "cuda:0"
for counts_shape, n_obs in (((256, 256), 8), ((512, 512), 8), ((1024, 1024), 8)):
    datasets, _, flux_init, cal = instrument_observations(shape=counts_shape, n_obs=n_obs, seed=0, psf_shape=(33, 33))
    comp = SpatialFluxComponent.from_numpy(flux=flux_init, upsampling_factor=2, prior=UniformPrior())
    cals = NPredCalibrations()
    for name, (sx, sy, norm) in cal.items():
        cals[name] = NPredCalibration(shift_x=sx, shift_y=sy, background_norm=norm)
    session = MAPDeconvolver(n_epochs=1, display_progress=False, device=DEV, fit_mode="joint").session(datasets, components=comp, calibrations=cals)
    assert session.batch_joint_calibrated
    for rnd in range(2):
        for batched in (2, 0):
            _hip.set_option("JD_FFT_BATCH", batched)
            for _ in range(3): session.epoch()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): session.epoch()
            e1.record(); torch.cuda.synchronize()
            print("flux grid", 2 * counts_shape[0], n_obs, "batched" if batched else "loop", f"{e0.elapsed_time(e1) / 20 * 1e3:.1f} us/step")
    del session
