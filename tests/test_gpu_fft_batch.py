"""The batched joint step on the native FFT path (jd_npred_poisson_batch_multi_fwd_bwd with a native FFT plan: every launch
of the likelihood step covers all datasets, the last one adds their gradients in dataset order inside its blocks)
against the per-dataset calls it stands for -- bit for bit -- and against the oracle."""
import numpy as np
import pytest
import torch

from conftest import rel_linf
from oracle import cpu_ref

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _session(shape, n_obs, psf_shapes, monkeypatch):
    from jolideco_amd import MAPDeconvolver, SpatialFluxComponent, UniformPrior
    from jolideco_amd.data import instrument_like_psf, synthetic_observations

    monkeypatch.setenv("JOLIDECO_CONV_METHOD", "fft")
    datasets, _, flux_init = synthetic_observations(shape=shape, n_obs=n_obs, seed=2)
    for i, d in enumerate(datasets.values()):
        d["psf"] = instrument_like_psf(i, psf_shapes[i % len(psf_shapes)])
    comp = SpatialFluxComponent.from_numpy(flux=flux_init, prior=UniformPrior())
    deco = MAPDeconvolver(n_epochs=1, display_progress=False, device=DEV, fit_mode="joint")
    session = deco.session(datasets, components=comp)
    session.cfg._optimizer_step = lambda states, step: None  # keep the gradient buffer, no update
    return session, datasets


@pytest.mark.parametrize("shape,n_obs", [((1024, 1024), 5), ((512, 256), 3)])
def test_batched_fft_step_equals_the_per_dataset_calls_bit_for_bit(shape, n_obs, monkeypatch, jd_option):
    """(1024^2: compile-time row / column schedules; 512 x 256: the generic kernels.)  PSFs of two sizes share the plan of
    the larger one (`common_kernel_shape`)."""
    session, _ = _session(shape, n_obs, [(17, 17), (33, 33)], monkeypatch)
    plans = {m.plan for mm in session.total_loss.poisson_loss.npred_models_all for m in mm.values()}
    assert len(plans) == 1 and all(p.method == "fft" and p.native_fft for p in plans)
    assert session.batch_joint
    out = {}
    for batched in (1, 0):
        jd_option("JD_FFT_BATCH", batched)
        session.epoch()
        torch.cuda.synchronize()
        out[batched] = session.comm.cpu().numpy().copy()
    n = shape[0] * shape[1]
    assert np.all(np.isfinite(out[1])) and np.any(out[1][:n] != 0)
    np.testing.assert_array_equal(out[1], out[0])  # gradient image and every dataset loss


def test_batched_fft_step_matches_the_oracle(monkeypatch):
    """The gradient and the losses of one batched joint step (4 datasets, 1024^2, general 33x33 PSFs) against autograd
    of the oracle (`cpu_ref.DatasetRef.loss`)."""
    shape, n_obs = (1024, 1024), 4
    session, datasets = _session(shape, n_obs, [(33, 33)], monkeypatch)
    assert session.batch_joint
    session.epoch()
    torch.cuda.synchronize()
    n = shape[0] * shape[1]
    comm = session.comm.cpu().numpy()
    grad, scalars = comm[:n].reshape(shape), comm[n : n + n_obs]
    seen = session.states[0].flux_cur.cpu().numpy()
    flux = torch.from_numpy(np.ascontiguousarray(seen))[None, None].requires_grad_(True)
    losses = []
    for d in datasets.values():
        loss = cpu_ref.DatasetRef.from_numpy(d, ["flux"]).loss((flux,))
        loss.backward()
        losses.append(float(loss.detach()))
    np.testing.assert_allclose(scalars, np.array(losses), rtol=5e-6)
    err = rel_linf(grad, flux.grad.numpy()[0, 0])
    assert err < 1e-5, err


def test_batched_calibrated_upsampled_step_equals_the_per_dataset_calls_bit_for_bit(monkeypatch, jd_option):
    """jd_npred_poisson_calibrated_batch_fwd_bwd (calibrations + up-sampling x2, general PSFs: the c6 shape at 512^2 flux
    pixels x 4 observations, one of them without a trained shift) against the per-dataset calls: flux gradient, losses and
    the gradients of every calibration parameter, bit for bit.  (Oracle parity of the same step:
    test_gpu_baseline_parity.py::test_c6_shaped_calibrated_upsampled_joint_step_1024_4obs, which runs this path.)"""
    from jolideco_amd import MAPDeconvolver, NPredCalibration, NPredCalibrations, SpatialFluxComponent, UniformPrior
    from jolideco_amd.data import instrument_observations

    datasets, _, flux_init, cal = instrument_observations(shape=(256, 256), n_obs=4, seed=1, psf_shape=(17, 17))
    out = {}
    for batched in (1, 0):
        jd_option("JD_FFT_BATCH", batched)
        comp = SpatialFluxComponent.from_numpy(flux=flux_init, upsampling_factor=2, prior=UniformPrior())
        cals = NPredCalibrations()
        for i, (name, (sx, sy, norm)) in enumerate(cal.items()):
            cals[name] = NPredCalibration(shift_x=sx if i else 0.0, shift_y=sy if i else 0.0, background_norm=norm)
        first = next(iter(cals.values()))
        first.shift_xy.requires_grad = False  # (the reference observation of the Chandra example)
        deco = MAPDeconvolver(n_epochs=1, display_progress=False, device=DEV, fit_mode="joint")
        session = deco.session(datasets, components=comp, calibrations=cals)
        assert session.batch_joint_calibrated and not session.batch_joint
        session.cfg._optimizer_step = lambda states, step: None
        session.epoch()
        torch.cuda.synchronize()
        grads = []
        for c in session.calibrations.values():
            grads.append(None if c.shift_xy.grad is None else c.shift_xy.grad.cpu().numpy().copy())
            grads.append(c._background_norm.grad.cpu().numpy().copy())
        out[batched] = (session.comm.cpu().numpy().copy(), grads)
    assert np.all(np.isfinite(out[1][0])) and np.any(out[1][0] != 0)
    np.testing.assert_array_equal(out[1][0], out[0][0])
    for a, b in zip(out[1][1], out[0][1]):
        assert (a is None) == (b is None)
        if a is not None:
            np.testing.assert_array_equal(a, b)
