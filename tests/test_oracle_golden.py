"""CPU: pin oracle/cpu_ref.py against the golden fixtures generated from the LIVE reference
(oracle/refload/make_golden.py), including the reference's own known-answer numbers
(jolideco/tests/test_core.py:72-79,144-153,181-188 and SURVEY.md Appendix A).

The fixtures were produced on the same torch build, where the oracle reproduces the reference
bit for bit (asserted at generation time).  Here a small tolerance absorbs a different thread
count / BLAS blocking on another host.
"""
import numpy as np
import pytest
import torch

from conftest import rel_linf, unpack_datasets
from oracle import cpu_ref

TOL = 2e-6


def _trace_close(rows, arrays, prefix="trace/", rtol=1e-5):
    for key, ref in arrays.items():
        if key.startswith(prefix):
            name = key[len(prefix):]
            mine = np.array([r[name] for r in rows])
            np.testing.assert_allclose(mine, ref, rtol=rtol, atol=1e-7, err_msg=name)


def test_rng_draw_order(golden):
    """cycle_spin draw order with torch's default CPU generator seed (utils/torch.py:108-116)."""
    gen = torch.Generator(device="cpu")
    assert gen.initial_seed() == cpu_ref.TORCH_DEFAULT_GENERATOR_SEED
    mine = [cpu_ref.draw_cycle_spin_shifts(gen, (8, 8)) for _ in range(32)]
    ref = golden("rng_draws")["shifts"]
    assert np.array_equal(np.array(mine), ref)
    # first eight randint(-2, 3) draws
    assert [v for pair in mine[:4] for v in pair] == [-2, 2, 0, -2, 2, 2, 2, 2]


def test_anchor_a(golden):
    """Config 1: 128^2 point source, uniform prior, 50 epochs."""
    a = golden("anchor_a")
    final, trace = cpu_ref.map_fit_sequential(
        unpack_datasets(a), {"flux": a["flux_init"]}, {"flux": cpu_ref.UniformPriorRef()}, n_epochs=50
    )
    assert rel_linf(final["flux"], a["flux_final"]) < TOL
    np.testing.assert_allclose(final["flux"][64, 64], 38.512722, rtol=1e-6)
    np.testing.assert_allclose(final["flux"][0, 0], 2.005425, rtol=1e-6)
    np.testing.assert_allclose(final["flux"].sum(), 32797.7734, rtol=1e-6)
    np.testing.assert_allclose(trace[-1]["total"], 2.311005, rtol=1e-6)
    _trace_close(trace, a)


def test_anchor_b(golden):
    """64^2, 3 observations, GMM K=8, 10 epochs sequential: step order, beta/N, RNG order, stale trace."""
    b = golden("anchor_b")
    gmm = cpu_ref.GMM.from_numpy(b["gmm_means"], b["gmm_covariances"], b["gmm_weights"], stride=4)
    final, trace, steps = cpu_ref.map_fit_sequential(
        unpack_datasets(b), {"flux": b["flux_init"]}, {"flux": cpu_ref.GMMPatchPriorRef(gmm)}, n_epochs=10,
        record_steps=True,
    )
    assert rel_linf(final["flux"], b["flux_final"]) < TOL
    assert rel_linf(steps[0]["grads"][0], b["grad_step0"]) < TOL
    np.testing.assert_allclose(final["flux"][32, 32], 16.988111, rtol=1e-6)
    np.testing.assert_allclose(final["flux"].sum(), 60076.6172, rtol=1e-6)
    np.testing.assert_allclose(trace[-1]["total"], 37.910606, rtol=1e-6)
    np.testing.assert_allclose(trace[-1]["prior-flux"], 1.404399, rtol=1e-5)
    _trace_close(trace, b)


def test_reference_known_answers(golden):
    """The reference's own golden pixels (jolideco/tests/test_core.py) at its own rtol 1e-3, and
    full-image parity with the stored reference outputs."""
    r = golden("reference_tests")
    flux_init = r["flux_init"]
    gauss = unpack_datasets(r, "gauss/data/")
    disk = unpack_datasets(r, "disk/data/")

    final, trace = cpu_ref.map_fit_sequential(
        gauss, {"flux-1": flux_init}, {"flux-1": cpu_ref.UniformPriorRef()}, n_epochs=100
    )
    np.testing.assert_allclose(final["flux-1"][12, 12], 1.542659, rtol=1e-3)  # test_core.py:72-79
    assert rel_linf(final["flux-1"], r["uniform/flux_final"]) < TOL
    _trace_close(trace, r, prefix="uniform/trace/")

    final, trace = cpu_ref.map_fit_sequential(
        disk, {"flux-1": flux_init}, {"flux-1": cpu_ref.InverseGammaPriorRef(alpha=10)}, n_epochs=100
    )
    np.testing.assert_allclose(final["flux-1"][12, 12], 0.136798, rtol=1e-3)  # test_core.py:144-153
    assert rel_linf(final["flux-1"], r["inverse_gamma/flux_final"]) < TOL
    _trace_close(trace, r, prefix="inverse_gamma/trace/")

    train = {n: disk[n] for n in ["0", "1"]}
    val = {n: disk[n] for n in ["2"]}
    final, trace = cpu_ref.map_fit_sequential(
        train, {"flux-1": flux_init}, {"flux-1": cpu_ref.ExponentialPriorRef(alpha=1)}, n_epochs=100,
        datasets_validation=val,
    )
    np.testing.assert_allclose(final["flux-1"][12, 12], 1.382768, rtol=1e-3)  # test_core.py:181-188
    assert rel_linf(final["flux-1"], r["exponential/flux_final"]) < TOL
    _trace_close(trace, r, prefix="exponential/trace/")


STAGE_CASES = ["sq96_psf17", "rect80x112_psf12x16", "rect97x110_psf9x5", "sq256_psf33"]


@pytest.mark.parametrize("name", STAGE_CASES)
def test_stage_forward_model_and_poisson(golden, name):
    """npred, Poisson NLL and d loss / d theta for non-square images, even-sized and asymmetric PSFs."""
    s = golden("stages")
    data = unpack_datasets({k[len(name) + 1:]: v for k, v in s.items() if k.startswith(name + "/data/")})["d"]
    loss, npred, grad = cpu_ref.poisson_loss_and_grad(s[f"{name}/theta"], data)
    assert rel_linf(npred, s[f"{name}/npred"]) < TOL
    np.testing.assert_allclose(loss, float(s[f"{name}/loss"]), rtol=1e-6)
    assert rel_linf(grad, s[f"{name}/grad_theta"]) < TOL
    # independent float64 statement of the loss formula
    np.testing.assert_allclose(cpu_ref.poisson_nll_numpy(npred, data["counts"]), loss, rtol=3e-6)
    # edge-corrected exposure (models/npred.py:108-113)
    d = cpu_ref.DatasetRef.from_numpy(data, ["flux"])
    assert rel_linf(d.exposures[0].numpy()[0, 0], s[f"{name}/exposure_corrected"]) < TOL


@pytest.mark.parametrize("gname", ["k16", "k5m"])
@pytest.mark.parametrize("name", STAGE_CASES[:3])
def test_stage_gmm_prior(golden, name, gname):
    """GMM prior value / gradient / arg-max for every stored shift, max and logsumexp."""
    s = golden("stages")
    gmm = cpu_ref.GMM.from_numpy(
        s[f"gmm/{gname}/means"], s[f"gmm/{gname}/covariances"], s[f"gmm/{gname}/weights"], stride=4
    )
    flux = np.exp(s[f"{name}/theta"].astype(np.float32))
    prefix = f"{name}/prior/{gname}/"
    shifts = sorted({k[len(prefix):].split("/")[0] for k in s if k.startswith(prefix)})
    assert shifts
    for tag in shifts:
        sy, sx = (int(v) for v in tag[1:].split("_"))
        for mode in ("max", "lse"):
            key = f"{prefix}{tag}/{mode}"
            value, grad, arg = cpu_ref.gmm_prior_value_and_grad(flux, gmm, 4, (sy, sx), mode == "lse")
            np.testing.assert_allclose(value, float(s[f"{key}/value"]), rtol=2e-6)
            same_choice = True
            if mode == "max":
                # the fixture's flux is float32(gamma*3); exp(log(.)) differs from it by <= 1 ulp,
                # which may flip the arg-max of a near-tie patch (and with it that patch's gradient)
                ref_arg, margin = s[f"{key}/argmax"], s[f"{key}/margin"]
                clear = margin > 1e-3
                assert np.array_equal(arg[clear], ref_arg[clear])
                assert (arg != ref_arg).sum() <= 2
                same_choice = bool((arg == ref_arg).all())
            if same_choice:
                # logsumexp responsibilities amplify the 1-ulp input difference (l ~ 1e2 with an
                # fp32 absolute error ~1e-5 moves a near-tie responsibility by ~1e-4)
                assert rel_linf(grad, s[f"{key}/grad_flux"]) < (1e-5 if mode == "max" else 5e-4)


def test_gmm_log_prob_matches_sklearn():
    """estimate_log_prob against scikit-learn (jolideco/priors/patches/tests/test_gmm.py:10-35:
    meta.stride None => unit pixel weights)."""
    from sklearn.mixture import GaussianMixture
    from sklearn.mixture._gaussian_mixture import _compute_precision_cholesky

    rs = np.random.RandomState(0)
    K, D = 3, 64
    means, covs, weights = cpu_ref.synthetic_gmm(K, D, seed=9, zero_means=False)
    sk = GaussianMixture(n_components=K, covariance_type="full")
    sk.means_, sk.covariances_, sk.weights_ = means, covs, weights
    sk.precisions_cholesky_ = _compute_precision_cholesky(covs, "full")
    x = rs.normal(size=(50, D)) * 0.3
    expected = sk._estimate_weighted_log_prob(x)
    gmm = cpu_ref.GMM.from_numpy(means, covs, weights, stride=None)
    got = cpu_ref.gmm_log_prob(torch.from_numpy(x.astype(np.float32)), gmm).numpy()
    np.testing.assert_allclose(got, expected, rtol=2e-4, atol=2e-3)


def test_pixel_weights_known_values():
    """outer product of [1/8,3/8,5/8,7/8,...] rescaled to sum stride^2."""
    w = cpu_ref.pixel_weights((8, 8), 4)
    np.testing.assert_allclose(w.sum(), 16.0, rtol=1e-12)
    np.testing.assert_allclose(w[0, 0], 0.015625, rtol=1e-12)
    np.testing.assert_allclose(w[3, 3], 0.765625, rtol=1e-12)
    np.testing.assert_allclose(w, w.T)


def test_joint_harness(golden):
    """Joint mode: one Adam step per epoch on sum_d L_d - beta * logprior."""
    j = golden("joint_multi")
    gmm = cpu_ref.GMM.from_numpy(j["gmm/means"], j["gmm/covariances"], j["gmm/weights"], stride=4)
    datasets = unpack_datasets(j, "joint/data/")
    final, trace = cpu_ref.map_fit_joint(
        datasets, {"flux": j["joint/flux_init"]}, {"flux": cpu_ref.GMMPatchPriorRef(gmm)}, n_epochs=12
    )
    assert rel_linf(final["flux"], j["joint/flux_final"]) < TOL
    _trace_close(trace, j, prefix="joint/trace/")


def test_two_components_per_component_psf(golden):
    """Config 5 shape: two components with their own PSFs, GMM + inverse-gamma priors, beta 0.7."""
    j = golden("joint_multi")
    gmm = cpu_ref.GMM.from_numpy(j["gmm/means"], j["gmm/covariances"], j["gmm/weights"], stride=4)
    datasets = unpack_datasets(j, "multi/data/")
    final, trace = cpu_ref.map_fit_sequential(
        datasets,
        {"extended": j["multi/init/extended"], "points": j["multi/init/points"]},
        {"extended": cpu_ref.GMMPatchPriorRef(gmm), "points": cpu_ref.InverseGammaPriorRef(10, 1.5)},
        n_epochs=6, beta=0.7,
    )
    assert rel_linf(final["extended"], j["multi/final/extended"]) < TOL
    assert rel_linf(final["points"], j["multi/final/points"]) < TOL
    _trace_close(trace, j, prefix="multi/trace/")


def test_upsampling_known_answers(golden):
    """upsampling_factor=2 (the reference's own test, jolideco/tests/test_core.py:99-124) and an odd
    factor 3 with a GMM prior on the up-sampled flux."""
    u = golden("upsampling")
    disk = unpack_datasets(u, "disk/data/")
    final, trace = cpu_ref.map_fit_sequential(
        disk, {"flux-1": u["flux_init"]}, {"flux-1": cpu_ref.UniformPriorRef()}, n_epochs=100,
        upsampling_factors={"flux-1": 2},
    )
    assert final["flux-1"].shape == (64, 64)
    assert rel_linf(final["flux-1"], u["u2/flux_upsampled_final"]) < TOL
    flux = cpu_ref.downsampled_flux(torch.from_numpy(final["flux-1"])[None, None], 2).numpy()[0, 0]
    assert rel_linf(flux, u["u2/flux_final"]) < TOL
    np.testing.assert_allclose(flux[12, 12], 3.565998, rtol=1e-3)  # test_core.py:117-124
    np.testing.assert_allclose(flux[0, 0], 1.605782, rtol=1e-3)
    np.testing.assert_allclose(trace[-1]["total"], 5.844786, rtol=1e-3)
    np.testing.assert_allclose(trace[-1]["dataset-0"], 1.946759, rtol=1e-3)
    _trace_close(trace, u, prefix="u2/trace/")

    gmm = cpu_ref.GMM.from_numpy(u["u3/gmm_means"], u["u3/gmm_covariances"], u["u3/gmm_weights"], stride=4)
    final, trace = cpu_ref.map_fit_sequential(
        unpack_datasets(u, "u3/data/"), {"flux": u["u3/flux_init"]}, {"flux": cpu_ref.GMMPatchPriorRef(gmm)},
        n_epochs=5, upsampling_factors={"flux": 3},
    )
    assert rel_linf(final["flux"], u["u3/flux_upsampled_final"]) < TOL
    _trace_close(trace, u, prefix="u3/trace/")


@pytest.mark.parametrize("tag,u,n_epochs", [("u1", 1, 8), ("u2", 2, 5)])
def test_calibrations(golden, tag, u, n_epochs):
    """NPredCalibrations (jolideco/models/npred.py:298-510): trainable shift + background norm, fixed PSF
    scale, a shift that starts at exactly 0 (never moves: shift_image_torch bypass) and a frozen one."""
    c = golden("calibration")
    datasets = unpack_datasets(c, f"{tag}/data/")
    gmm = cpu_ref.GMM.from_numpy(c[f"{tag}/gmm_means"], c[f"{tag}/gmm_covariances"], c[f"{tag}/gmm_weights"], stride=4)
    cals = {}
    for name in datasets:
        sx, sy, norm, psf_scale, frozen = c[f"{tag}/cal_init/{name}"]
        cals[name] = cpu_ref.CalibrationRef.create(sx, sy, norm, psf_scale, bool(frozen))
    final, trace = cpu_ref.map_fit_sequential(
        datasets, {"flux": c[f"{tag}/flux_init"]}, {"flux": cpu_ref.GMMPatchPriorRef(gmm)}, n_epochs=n_epochs,
        upsampling_factors={"flux": u}, calibrations=cals,
    )
    assert rel_linf(final["flux"], c[f"{tag}/flux_upsampled_final"]) < TOL
    _trace_close(trace, c, prefix=f"{tag}/trace/")
    for name in datasets:
        d = cals[name].to_dict()
        got = np.array([d["shift_x"], d["shift_y"], d["background_norm"], d["psf_scale"]])
        np.testing.assert_allclose(got, c[f"{tag}/cal_final/{name}"], rtol=1e-5, atol=1e-7)
    assert cals["o1"].to_dict()["shift_x"] == 0.0  # bypassed shift gets no gradient


@pytest.mark.parametrize("tag", ["nomask", "mask"])
def test_linear_flux_parameter(golden, tag):
    """use_log_flux=False incl. the reference's trace quirk (post-step flux without a mask, stale flux with one)."""
    g = golden("linear_flux")
    gmm = cpu_ref.GMM.from_numpy(g[f"{tag}/gmm_means"], g[f"{tag}/gmm_covariances"], g[f"{tag}/gmm_weights"], stride=4)
    masks = {"flux": g[f"{tag}/mask"]} if tag == "mask" else None
    final, trace = cpu_ref.map_fit_sequential(
        unpack_datasets(g, f"{tag}/data/"), {"flux": g[f"{tag}/flux_init"]}, {"flux": cpu_ref.GMMPatchPriorRef(gmm)},
        n_epochs=6, masks=masks, use_log_flux=False,
    )
    assert rel_linf(final["flux"], g[f"{tag}/flux_final"]) < TOL
    _trace_close(trace, g, prefix=f"{tag}/trace/")


def test_upsampling_with_per_component_psfs_of_different_shapes(golden):
    """upsampling_factor=2, two components, a 9x9 and a 5x5 PSF with strong edges (live-reference fixture): every PSF
    is up-sampled as given (models/npred.py:96-106)."""
    m = golden("upsampling_mixed_psf")
    datasets = unpack_datasets(m)
    gmm = cpu_ref.GMM.from_numpy(m["gmm/means"], m["gmm/covariances"], m["gmm/weights"], stride=4)
    final, trace = cpu_ref.map_fit_sequential(
        datasets, {"extended": m["init/extended"], "points": m["init/points"]},
        {"extended": cpu_ref.GMMPatchPriorRef(gmm), "points": cpu_ref.InverseGammaPriorRef(10, 1.5)},
        n_epochs=5, upsampling_factors={"extended": 2, "points": 2},
    )
    for name in ("extended", "points"):
        assert rel_linf(final[name], m[f"final_upsampled/{name}"]) < TOL
    _trace_close(trace, m)
