"""CPU: host-side logic of jolideco_amd that runs before / around the HIP calls -- constants handed
to the library, the trace table, optimizer scalars, synthetic data, API surface."""
import math

import numpy as np
import pytest
import torch

from oracle import cpu_ref


def test_gmm_constants_match_oracle():
    """precisions_cholesky, mu P, log|P|, log pi and the pixel weights handed to jd_gmm_create are
    the reference's (patches/gmm.py:119-149,217-240,283-299; utils/numpy.py:16-79)."""
    from jolideco_amd.priors.patches import GaussianMixtureModel, GaussianMixtureModelMeta

    means, covs, weights = cpu_ref.synthetic_gmm(6, 64, seed=4, zero_means=False)
    gmm = GaussianMixtureModel.from_numpy(means, covs, weights, meta=GaussianMixtureModelMeta(stride=4))
    ref = cpu_ref.GMM.from_numpy(means, covs, weights, stride=4)
    assert np.array_equal(gmm.precisions_cholesky_numpy, ref.precisions_cholesky.numpy())
    assert np.array_equal(gmm.means_precisions_cholesky_numpy, ref.means_precisions_cholesky.numpy())
    assert np.array_equal(gmm.log_det_cholesky_numpy, ref.log_det_cholesky.numpy())
    assert np.array_equal(gmm.log_weights_numpy, ref.log_weights.numpy())
    assert np.array_equal(gmm.pixel_weights_numpy.astype(np.float32), ref.pixel_weights.numpy())
    assert gmm.patch_shape == (8, 8) and gmm.n_components == 6 and gmm.n_features == 64
    # stride None => unit weights (gmm.py:290-293)
    gmm1 = GaussianMixtureModel.from_numpy(means, covs, weights)
    assert np.array_equal(gmm1.pixel_weights_numpy, np.ones((1, 64)))


def test_pixel_weights_and_precision_cholesky():
    from jolideco_amd.utils.numpy import compute_precision_cholesky, get_pixel_weights

    w = get_pixel_weights((8, 8), 4)
    assert np.allclose(w, cpu_ref.pixel_weights((8, 8), 4), rtol=0, atol=0)
    _, covs, _ = cpu_ref.synthetic_gmm(3, 64, seed=1)
    pc = compute_precision_cholesky(covs)
    assert np.array_equal(pc, cpu_ref.precision_cholesky(covs))
    # P P^T = Sigma^-1
    assert np.allclose(pc[0] @ pc[0].T, np.linalg.inv(covs[0]), rtol=1e-6, atol=1e-6)


def test_cycle_spin_shift_draws_follow_the_reference_order(golden):
    from jolideco_amd.utils.torch import cycle_spin_shifts, get_default_generator

    gen = get_default_generator("cuda")  # always a host generator
    assert gen.device.type == "cpu" and gen.initial_seed() == cpu_ref.TORCH_DEFAULT_GENERATOR_SEED
    mine = np.array([cycle_spin_shifts((8, 8), gen) for _ in range(32)])
    assert np.array_equal(mine, golden("rng_draws")["shifts"])


def test_an_epoch_worth_of_cycle_spin_draws_is_the_same_as_single_draws(golden):
    """`cycle_spin_shifts_many` (planned epochs draw all shifts of an epoch at once): the reference's draw sequence -- the
    live-reference fixture -- and the generator left in the state n single draws leave it in, for every batch size."""
    import torch

    from jolideco_amd.utils.torch import cycle_spin_shifts, cycle_spin_shifts_many, get_default_generator

    gen = get_default_generator("cpu")
    many = cycle_spin_shifts_many((8, 8), gen, 32)
    assert np.array_equal(np.array(many), golden("rng_draws")["shifts"])
    for n in (1, 2, 7, 16, 25, 64, 501):
        a, b = torch.Generator().manual_seed(n), torch.Generator().manual_seed(n)
        single = [cycle_spin_shifts((8, 8), a) for _ in range(n)]
        assert cycle_spin_shifts_many((8, 8), b, n) == single
        assert torch.equal(a.get_state(), b.get_state())
    # patches that are not square draw from two ranges: single draws
    a, b = torch.Generator().manual_seed(3), torch.Generator().manual_seed(3)
    assert cycle_spin_shifts_many((8, 16), b, 5) == [cycle_spin_shifts((8, 16), a) for _ in range(5)]


def test_adam_scalars_match_torch():
    """step_size / sqrt(bias2) as torch.optim.Adam computes them in python floats."""
    from jolideco_amd.ops import adam_bias_terms

    for step in (1, 2, 10, 1000):
        step_size, bias2_sqrt = adam_bias_terms(step, 0.1, 0.9, 0.999)
        assert step_size == 0.1 / (1 - 0.9**step)
        assert bias2_sqrt == math.sqrt(1 - 0.999**step)


def test_stirling_mean():
    from jolideco_amd.ops import stirling_mean

    counts = np.array([[0, 1, 2], [5, 100, 3]], dtype=np.float32)
    npred = np.full(counts.shape, 2.5)
    full = cpu_ref.poisson_nll_numpy(npred, counts)
    plain = float(np.mean(npred - counts * np.log(npred + 1e-25)))
    assert abs((full - plain) - stirling_mean(counts)) < 1e-12


def test_trace_row_layout_and_signs():
    """Columns, order and signs of jolideco/loss.py:192-250."""
    from jolideco_amd.loss import PriorLoss, TotalLoss
    from jolideco_amd.priors import Priors, UniformPrior

    class FakePoisson:
        names_all = ["a", "b"]
        counts_all = [None, None]

    priors = Priors()
    priors["flux"] = UniformPrior()
    total = TotalLoss(FakePoisson(), PriorLoss(priors), poisson_loss_validation=None, beta=0.5)
    assert total.trace_names == ["total", "datasets-total", "priors-total", "prior-flux", "dataset-a", "dataset-b", "filename"]
    assert total.prior_weight == 2
    row = total.make_row([1.0, 2.0], [4.0])
    assert row["datasets-total"] == 3.0 and row["priors-total"] == -2.0 and row["total"] == 1.0
    assert row["prior-flux"] == -2.0 and row["dataset-a"] == 1.0 and row["dataset-b"] == 2.0
    expected = cpu_ref._trace_row(["a", "b"], ["flux"], [1.0, 2.0], [4.0], 0.5)
    for key, value in expected.items():
        assert row[key] == value
    total.trace.add_row(row)
    assert len(total.trace) == 1 and total.trace[-1]["total"] == 1.0


def test_synthetic_data_known_answers():
    """jolideco/data/tests/test_core.py:16-43 known answers for the toy datasets."""
    from jolideco_amd.data import disk_source_gauss_psf, gauss_and_point_sources_gauss_psf, point_source_gauss_psf

    rs = np.random.RandomState(642020)
    data = point_source_gauss_psf(random_state=rs)
    assert data["counts"].shape == (32, 32) and data["psf"].shape == (17, 17)
    np.testing.assert_allclose(data["psf"][7][7], 0.015965, rtol=1e-3)
    np.testing.assert_allclose(data["flux"].sum(), 1000)
    np.testing.assert_allclose(data["psf"].sum(), 1.0, rtol=1e-6)
    data = disk_source_gauss_psf(random_state=np.random.RandomState(0))
    np.testing.assert_allclose(data["flux"].sum(), 1000, rtol=1e-5)
    np.testing.assert_allclose(data["exposure"][0, 0], 0.5)
    data = gauss_and_point_sources_gauss_psf(random_state=np.random.RandomState(0))
    assert data["flux"][26, 16] >= 1000 and data["exposure"][0, 0] == 0.5 and data["exposure"][-1, 0] == 1.5


def test_synthetic_observations_shapes():
    from jolideco_amd.data import synthetic_gmm, synthetic_observations

    datasets, truth, flux_init = synthetic_observations(shape=(64, 96), n_obs=3, seed=1, n_points=4)
    assert list(datasets) == ["obs-0", "obs-1", "obs-2"]
    for d in datasets.values():
        assert d["counts"].shape == (64, 96) and d["counts"].dtype == np.float32
        assert d["psf"].shape == (17, 17) and abs(d["psf"].sum() - 1) < 1e-5
        assert (d["counts"] >= 0).all() and (d["exposure"] > 0).all()
    assert truth.shape == flux_init.shape == (64, 96)
    means, covs, weights = synthetic_gmm(4, 64, seed=0)
    assert covs.shape == (4, 64, 64) and abs(weights.sum() - 1) < 1e-12 and not means.any()
    assert all(np.linalg.eigvalsh(c).min() > 0 for c in covs)


def test_api_surface_matches_the_reference():
    """Names and constructor arguments a user of jolideco finds."""
    import inspect

    import jolideco_amd as jd
    from jolideco_amd.priors import PRIOR_REGISTRY

    for name in ("MAPDeconvolver", "MAPDeconvolverResult", "FluxComponents", "SpatialFluxComponent", "NPredModel",
                 "NPredModels", "PoissonLoss", "PriorLoss", "TotalLoss", "GMMPatchPrior", "GaussianMixtureModel",
                 "UniformPrior", "InverseGammaPrior", "ExponentialPrior"):
        assert hasattr(jd, name), name
    params = inspect.signature(jd.MAPDeconvolver.__init__).parameters
    for arg in ("n_epochs", "beta", "learning_rate", "compute_error", "stop_early", "stop_early_n_average", "device",
                "display_progress", "optimizer_type", "optimizer_kwargs", "checkpoint_path"):
        assert arg in params, arg
    assert params["n_epochs"].default == 1000 and params["learning_rate"].default == 0.1 and params["beta"].default == 1
    run = inspect.signature(jd.MAPDeconvolver.run).parameters
    assert list(run)[1:] == ["datasets", "datasets_validation", "components", "calibrations"]
    gp = inspect.signature(jd.GMMPatchPrior.__init__).parameters
    for arg in ("gmm", "stride", "cycle_spin", "cycle_spin_subpix", "generator", "norm", "patch_norm", "jitter",
                "marginalize", "device"):
        assert arg in gp, arg
    assert {"uniform", "inverse-gamma", "exponential", "gmm-patches"} <= set(PRIOR_REGISTRY)


def test_unsupported_options_raise_instead_of_silently_differing():
    import jolideco_amd as jd
    from jolideco_amd.priors.patches import GaussianMixtureModel, GaussianMixtureModelMeta

    means, covs, weights = cpu_ref.synthetic_gmm(2, 64, seed=0)
    gmm = GaussianMixtureModel.from_numpy(means, covs, weights, meta=GaussianMixtureModelMeta(stride=4))
    with pytest.raises(NotImplementedError):
        jd.GMMPatchPrior(gmm=gmm, jitter=True)
    with pytest.raises(NotImplementedError):
        jd.GMMPatchPrior(gmm=gmm, cycle_spin_subpix=True)
    lin = jd.SpatialFluxComponent.from_numpy(np.full((8, 8), 3.0), use_log_flux=False)
    assert not lin.use_log_flux and torch.allclose(lin._flux_upsampled, torch.full((1, 1, 8, 8), 3.0))
    with pytest.raises(ValueError):
        jd.MAPDeconvolver(optimizer_type="lbfgs", device="cuda")
    with pytest.raises(ValueError):
        jd.MAPDeconvolver(fit_mode="nope", device="cuda")
    comp = jd.SpatialFluxComponent.from_numpy(np.full((8, 8), 2.0))
    assert torch.allclose(comp.flux_upsampled, torch.full((1, 1, 8, 8), 2.0))
    assert len(jd.FluxComponents({"flux": comp}).parameters()) == 1
    comp.frozen = True
    assert jd.FluxComponents({"flux": comp}).parameters() == []


def test_result_write_read_roundtrip(tmp_path):
    """MAPDeconvolverResult.write / read (numpy .npz; the reference's FITS / ASDF writers need astropy / asdf)."""
    import jolideco_amd as jd
    from jolideco_amd.utils.table import TraceTable

    comps = jd.FluxComponents()
    comps["a"] = jd.SpatialFluxComponent.from_numpy(np.arange(1, 65, dtype=float).reshape(8, 8), upsampling_factor=2)
    comps["b"] = jd.SpatialFluxComponent.from_numpy(np.full((16, 16), 2.5), use_log_flux=False, frozen=True)
    trace = TraceTable(names=["total", "dataset-x", "filename"])
    trace.add_row({"total": 1.5, "dataset-x": 2.5, "filename": ""})
    trace.add_row({"total": 1.25, "dataset-x": 2.0, "filename": ""})
    cals = jd.NPredCalibrations()
    cals["x"] = jd.NPredCalibration(shift_x=0.25, shift_y=-0.5, background_norm=1.5, frozen=True)
    result = jd.MAPDeconvolverResult(config={}, components=comps, trace_loss=trace, calibrations=cals)
    path = tmp_path / "result.npz"
    result.write(path)
    with pytest.raises(OSError):
        result.write(path)
    back = jd.MAPDeconvolverResult.read(path)
    for name in ("a", "b"):
        np.testing.assert_allclose(back.components[name].flux_upsampled_numpy, comps[name].flux_upsampled_numpy, rtol=1e-6)
    assert back.components["a"].upsampling_factor == 2 and back.components["a"].flux_numpy.shape == (8, 8)
    assert not back.components["b"].use_log_flux and back.components["b"].frozen
    np.testing.assert_allclose(back.trace_loss["total"], [1.5, 1.25])
    d = back.calibrations["x"].to_dict()
    assert d["shift_x"] == 0.25 and d["shift_y"] == -0.5 and abs(d["background_norm"] - 1.5) < 1e-6 and d["frozen"]


def test_prior_hessian_times_ones_matches_autograd():
    """`Prior.hessian_ones` (the prior's share of jolideco/loss.py:263-279) against torch's double backward of
    the same log-prior written with torch ops."""
    import torch

    from jolideco_amd.loss import PriorLoss, TotalLoss
    from jolideco_amd.priors import ExponentialPrior, InverseGammaPrior, Priors, UniformPrior

    flux = torch.tensor(np.random.RandomState(3).gamma(2.0, size=(1, 1, 6, 7)).astype(np.float32))
    prior = InverseGammaPrior(alpha=0.1)

    def log_prior(x):
        return torch.sum(-prior.beta / x + (-prior.alpha - 1) * torch.log(x)) / x.numel() + prior.log_constant_term

    expected = torch.autograd.functional.vhp(log_prior, flux, v=torch.ones_like(flux))[1]
    np.testing.assert_allclose(prior.hessian_ones(flux).numpy(), expected.numpy(), rtol=2e-6, atol=1e-9)
    assert torch.all(ExponentialPrior(alpha=2).hessian_ones(flux) == 0)
    assert torch.all(UniformPrior().hessian_ones(flux) == 0)

    priors = Priors({"a": prior, "b": UniformPrior()})
    total = TotalLoss(poisson_loss=None, prior_loss=PriorLoss(priors), beta=2.0)
    errors = total.fluxes_error(fluxes=(flux, flux))
    assert list(errors) == ["a", "b"]
    with np.errstate(invalid="ignore", divide="ignore"):
        ref = np.sqrt(1.0 / (-2.0 * expected.numpy().astype(np.float64)))
    got = errors["a"].numpy()
    assert np.array_equal(np.isnan(got), np.isnan(ref))
    np.testing.assert_allclose(got[np.isfinite(ref)], ref[np.isfinite(ref)], rtol=1e-5)
    assert torch.all(torch.isinf(errors["b"]))
