"""Planned epochs (round 5): the per-step scalars of an epoch -- cycle-spin shifts of every prior evaluation, Adam bias terms
of every optimizer step -- live in device memory (`core.StepScalars`; include/jolideco_hip.h: device-resident step scalars),
so an epoch's launch arguments never change and the epoch is captured in a hipGraph and replayed.  Everything here compares
with the by-value form of rounds 1-4 (JOLIDECO_STEP_SCALARS=host), which the parity tests hold against the oracle: same
kernels, same order, same arithmetic -- the results must agree BIT FOR BIT, replayed or not."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _gmm(k=16, seed=2):
    from jolideco_amd.data import synthetic_gmm
    from jolideco_amd.priors.patches import GaussianMixtureModel, GaussianMixtureModelMeta

    means, covs, weights = synthetic_gmm(k, 64, seed=seed)
    return GaussianMixtureModel.from_numpy(means, covs, weights, meta=GaussianMixtureModelMeta(stride=4))


def _fit(monkeypatch, mode, build, n_epochs, fit_mode):
    """Run `n_epochs` epochs of the fit `build()` describes; mode: "host" (by value), "device" (planned, no capture),
    "graph" (planned + captured).  Returns (fluxes, trace scalars per epoch, calibration values, graphs captured)."""
    from jolideco_amd import MAPDeconvolver

    monkeypatch.setenv("JOLIDECO_STEP_SCALARS", "host" if mode == "host" else "device")
    monkeypatch.setenv("JOLIDECO_GRAPH", "1" if mode == "graph" else "0")
    datasets, components, calibrations = build()
    deco = MAPDeconvolver(n_epochs=n_epochs, display_progress=False, device=DEV, fit_mode=fit_mode)
    session = deco.session(datasets, components=components, calibrations=calibrations)
    rows = []
    for _ in range(n_epochs):
        session.epoch()
        rows.append(session.scalars.clone())
    torch.cuda.synchronize()
    fluxes = [st.flux_cur.cpu().numpy().copy() for st in session.states]
    thetas = [st.theta.cpu().numpy().copy() for st in session.states]
    cal = []
    if calibrations is not None:
        for c in calibrations.values():
            cal.append(np.concatenate([c.shift_xy.detach().cpu().numpy().ravel(), c._background_norm.detach().cpu().numpy().ravel()]))
    last = [getattr(p, "last_shifts", None) for p in session.priors]
    return fluxes + thetas, torch.stack(rows).cpu().numpy(), cal, len(session._graphs), (session.step, last)


def _assert_same(a, b):
    for x, y in zip(a[0], b[0]):
        np.testing.assert_array_equal(x, y)
    np.testing.assert_array_equal(a[1], b[1])
    for x, y in zip(a[2], b[2]):
        np.testing.assert_array_equal(x, y)
    assert a[4] == b[4]  # step count and the last shifts drawn: the host state of a replayed epoch


def _build_joint():
    from jolideco_amd import GMMPatchPrior, SpatialFluxComponent
    from jolideco_amd.data import synthetic_observations

    datasets, _, flux_init = synthetic_observations(shape=(96, 132), n_obs=4, seed=3)
    comp = SpatialFluxComponent.from_numpy(flux=flux_init, prior=GMMPatchPrior(gmm=_gmm(), generator=torch.Generator().manual_seed(5)))
    return datasets, comp, None


def _build_two_components():
    from jolideco_amd import FluxComponents, GMMPatchPrior, InverseGammaPrior, SpatialFluxComponent, UniformPrior
    from jolideco_amd.data import gaussian_kernel, synthetic_observations

    datasets, _, flux_init = synthetic_observations(shape=(72, 88), n_obs=3, seed=4)
    comps = FluxComponents()
    generator = torch.Generator().manual_seed(11)  # ONE generator behind two priors: the order of the draws matters
    comps["extended"] = SpatialFluxComponent.from_numpy(flux=flux_init, prior=GMMPatchPrior(gmm=_gmm(8, 3), generator=generator))
    comps["second"] = SpatialFluxComponent.from_numpy(flux=0.5 * flux_init, prior=GMMPatchPrior(gmm=_gmm(8, 4), generator=generator))
    comps["points"] = SpatialFluxComponent.from_numpy(flux=0.05 * flux_init, prior=InverseGammaPrior(alpha=10, beta=1.5))
    comps["flat"] = SpatialFluxComponent.from_numpy(flux=0.02 * flux_init, prior=UniformPrior(), frozen=True)
    for i, d in enumerate(datasets.values()):
        d["psf"] = {"extended": d["psf"], "second": d["psf"], "points": gaussian_kernel(1.0 + 0.1 * i, (9, 9)).astype(np.float32),
                    "flat": d["psf"]}
    return datasets, comps, None


def _build_calibrated():
    from jolideco_amd import GMMPatchPrior, NPredCalibration, NPredCalibrations, SpatialFluxComponent
    from jolideco_amd.data import instrument_observations

    datasets, _, flux_init, cal = instrument_observations(shape=(64, 80), n_obs=3, seed=1, psf_shape=(17, 17))
    comp = SpatialFluxComponent.from_numpy(flux=flux_init, upsampling_factor=2,
                                           prior=GMMPatchPrior(gmm=_gmm(), generator=torch.Generator().manual_seed(7)))
    cals = NPredCalibrations()
    for name, (sx, sy, norm) in cal.items():
        cals[name] = NPredCalibration(shift_x=sx, shift_y=sy, background_norm=norm)
    return datasets, comp, cals


CASES = {"joint": (_build_joint, "joint"), "sequential": (_build_joint, "sequential"),
         "components-joint": (_build_two_components, "joint"), "components-sequential": (_build_two_components, "sequential"),
         "calibrated-joint": (_build_calibrated, "joint"), "calibrated-sequential": (_build_calibrated, "sequential")}


@pytest.mark.parametrize("case", sorted(CASES))
def test_planned_and_replayed_epochs_equal_the_by_value_epochs_bit_for_bit(monkeypatch, case):
    build, fit_mode = CASES[case]
    n_epochs = 9  # three eager epochs, a capture per flux-buffer parity, replays
    by_value = _fit(monkeypatch, "host", build, n_epochs, fit_mode)
    planned = _fit(monkeypatch, "device", build, n_epochs, fit_mode)
    replayed = _fit(monkeypatch, "graph", build, n_epochs, fit_mode)
    assert by_value[3] == 0 and planned[3] == 0
    assert replayed[3] >= 1, "no epoch was captured"
    _assert_same(planned, by_value)
    _assert_same(replayed, by_value)


def test_run_under_the_auto_policy_matches_the_oracle(monkeypatch):
    """`MAPDeconvolver.run` (the reference's entry point) under the default policy -- eight timed by-value epochs, then by
    value or planned + captured, whichever the fit's host / device balance asks for -- gives the fit the oracle computes (the
    anchor-B shape: sequential mode, GMM prior, three observations; 16 epochs: the switch lies inside the fit)."""
    from conftest import rel_linf
    from jolideco_amd import GMMPatchPrior, MAPDeconvolver, SpatialFluxComponent
    from jolideco_amd.data import point_source_gauss_psf, synthetic_gmm
    from jolideco_amd.priors.patches import GaussianMixtureModel, GaussianMixtureModelMeta
    from oracle import cpu_ref

    monkeypatch.delenv("JOLIDECO_GRAPH", raising=False)
    monkeypatch.delenv("JOLIDECO_STEP_SCALARS", raising=False)
    rs = np.random.RandomState(11)
    datasets = {f"obs-{i}": point_source_gauss_psf(shape=(48, 40), sigma_psf=2 + i, random_state=rs) for i in range(3)}
    for d in datasets.values():
        d.pop("flux")
    flux_init = rs.gamma(30, size=(48, 40))
    means, covs, weights = synthetic_gmm(4, 64, seed=2)
    gmm = GaussianMixtureModel.from_numpy(means, covs, weights, meta=GaussianMixtureModelMeta(stride=4))
    comp = SpatialFluxComponent.from_numpy(flux=flux_init, prior=GMMPatchPrior(gmm=gmm))
    deco = MAPDeconvolver(n_epochs=16, display_progress=False, device=DEV)
    res = deco.run(datasets, components=comp)
    gmm_o = cpu_ref.GMM.from_numpy(means, covs, weights, stride=4)
    final, trace = cpu_ref.map_fit_sequential(datasets, {"flux": flux_init}, {"flux": cpu_ref.GMMPatchPriorRef(gmm_o)}, n_epochs=16)
    assert rel_linf(res.flux_total, final["flux"]) < 1e-5
    np.testing.assert_allclose(res.trace_loss["total"], [row["total"] for row in trace], rtol=1e-4)


def test_an_option_set_between_epochs_invalidates_the_captured_epochs(monkeypatch, jd_option):
    """A captured epoch holds the kernels that were launched when it was captured: `_hip.set_option` bumps a generation
    counter and the session captures anew (bench.py times the dense-GMM side run on the session of the headline run)."""
    from jolideco_amd import MAPDeconvolver

    monkeypatch.setenv("JOLIDECO_GRAPH", "1")
    datasets, comp, _ = _build_joint()
    session = MAPDeconvolver(n_epochs=1, display_progress=False, device=DEV, fit_mode="joint").session(datasets, components=comp)
    for _ in range(6):
        session.epoch()
    assert len(session._graphs) == 2
    jd_option("JD_GMM_SCREEN", 0)
    session.epoch()
    assert len(session._graphs) == 0 and session._epochs_done == 1
    for _ in range(5):
        session.epoch()
    torch.cuda.synchronize()
    assert len(session._graphs) == 2


def test_the_auto_policy_decides_after_its_probe_epochs(monkeypatch):
    """Default policy: epochs 1 .. 8 by value with the GMM prior's first phase beside the likelihood, epochs 9 .. 16 on one
    stream, both timed: the faster stays; a fit whose host time is a good part of the device's is then captured in both
    stream forms, 8 replays of each are timed, and the fastest of the three stays.  Whichever way it goes the fit is the by-value fit bit for bit."""
    from jolideco_amd import MAPDeconvolver

    n_epochs = 56  # first epoch, 2 x 8 probe epochs, per stream form 3 eager planned epochs + 2 captures + 8 timed replays, 5 to
    # capture the winner again if it was not the last one, 8 more
    monkeypatch.delenv("JOLIDECO_PRIOR_OVERLAP", raising=False)
    by_value = _fit(monkeypatch, "host", _build_joint, n_epochs, "joint")
    monkeypatch.delenv("JOLIDECO_GRAPH", raising=False)
    monkeypatch.delenv("JOLIDECO_STEP_SCALARS", raising=False)
    datasets, comp, _ = _build_joint()
    session = MAPDeconvolver(n_epochs=1, display_progress=False, device=DEV, fit_mode="joint").session(datasets, components=comp)
    rows, policies = [], []
    for i in range(n_epochs):
        session.epoch()
        rows.append(session.scalars.clone())
        policies.append(session.graph_policy)
        assert (session.graph_policy == "undecided") == (i < 16), (i, session.graph_policy)
    torch.cuda.synchronize()
    print("auto policy on the 96 x 132 joint fit:", policies[16], "->", session.graph_policy)
    assert session.graph_policy.startswith(("by value", "captured epochs")) and "on two streams" in session.graph_policy
    if "on trial" in policies[16]:
        assert "measured" in session.graph_policy and session._trial is None
    assert bool(session._graphs) == session.graph_policy.startswith("captured")
    np.testing.assert_array_equal(torch.stack(rows).cpu().numpy(), by_value[1])
    np.testing.assert_array_equal(session.states[0].flux_cur.cpu().numpy(), by_value[0][0])
