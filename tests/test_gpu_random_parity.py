"""Randomised (seeded) end-to-end parity of the HIP path against the CPU oracle: image shapes, PSF shapes (odd, even,
non-square), numbers of observations and components, GMM sizes, strides, arg-max / logsumexp, both fit modes.  The
fixed cases of test_gpu_fit.py pin the reference's own numbers; these widen the net.  Tolerance: see the comment
at the assertion (the bulk of the pixels to the north-star's 1e-5, isolated ill-conditioned pixels bounded), 2e-5 on the
trace."""
import numpy as np
import pytest

from oracle import cpu_ref

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _psf(rs, shape):
    from jolideco_amd.data import gaussian_kernel

    k = gaussian_kernel(rs.uniform(0.8, 2.5), shape)
    if rs.rand() < 0.4:  # core + wing: rank 2
        k = 0.8 * k + 0.2 * gaussian_kernel(rs.uniform(2.5, 4.0), shape)
    if rs.rand() < 0.25:  # not separable at all: the general (direct) kernel
        k = k * (1.0 + 0.3 * rs.rand(*shape))
    return (k / k.sum()).astype(np.float32)


_DISTANCES = {}  # seed -> [(|gpu - f64|, |oracle32 - f64|) per component], filled by the per-seed test


@pytest.mark.parametrize("seed", range(8))
def test_random_fit_matches_the_oracle(seed):
    from jolideco_amd import FluxComponents, GMMPatchPrior, InverseGammaPrior, MAPDeconvolver, SpatialFluxComponent
    from jolideco_amd.data import synthetic_gmm
    from jolideco_amd.priors.patches import GaussianMixtureModel, GaussianMixtureModelMeta

    rs = np.random.RandomState(1000 + seed)
    H, W = int(rs.randint(24, 90)), int(rs.randint(24, 110))
    kh, kw = int(rs.randint(3, 14)), int(rs.randint(3, 14))
    n_obs, n_comp = int(rs.randint(1, 4)), int(rs.randint(1, 3))
    joint = bool(seed % 2)
    marginalize = seed == 5
    stride = int(rs.choice([2, 4, 4, 5]))
    K = int(rs.choice([3, 8, 20]))
    names = ["extended", "points"][:n_comp]
    truth = 2.0 + 20.0 * np.exp(-0.5 * (((np.mgrid[0:H, 0:W][0] - H / 2) / (H / 5)) ** 2 + ((np.mgrid[0:H, 0:W][1] - W / 3) / (W / 6)) ** 2))
    datasets = {}
    for i in range(n_obs):
        exposure = (1.0 + 0.2 * i) * (1.0 + 0.4 * np.linspace(-1, 1, H)[:, None] * np.ones((H, W)))
        background = np.full((H, W), 0.3 + 0.2 * i)
        psfs = {name: _psf(rs, (kh, kw)) for name in names}
        datasets[f"obs-{i}"] = {
            "counts": rs.poisson(truth * exposure * 0.2 + background).astype(np.float32),
            "psf": psfs if (n_comp > 1 or rs.rand() < 0.5) else psfs["extended"],
            "exposure": exposure.astype(np.float32),
            "background": background.astype(np.float32),
        }
    flux_init = rs.gamma(20.0, size=(H, W)) * 0.2
    means, covs, weights = synthetic_gmm(K, 64, seed=seed)
    inits = {"extended": flux_init, "points": 0.1 * flux_init}

    # oracle: in the reference's precision (fp32) and, the same restatement, in float64 (how far fp32 is from exact)
    n_epochs = 4

    def run_oracle():
        gmm_o = cpu_ref.GMM.from_numpy(means, covs, weights, stride=4)
        priors_o = {"extended": cpu_ref.GMMPatchPriorRef(gmm_o, stride=stride, marginalize=marginalize)}
        if n_comp > 1:
            priors_o["points"] = cpu_ref.InverseGammaPriorRef(alpha=10.0, beta=1.5)
        fit_o = cpu_ref.map_fit_joint if joint else cpu_ref.map_fit_sequential
        return fit_o(datasets, {n: inits[n] for n in names}, {n: priors_o[n] for n in names}, n_epochs=n_epochs)

    final_o, trace_o = run_oracle()
    with cpu_ref.precision(np.float64):
        final_64, _ = run_oracle()

    # HIP path
    gmm = GaussianMixtureModel.from_numpy(means, covs, weights, meta=GaussianMixtureModelMeta(stride=4))
    comps = FluxComponents()
    comps["extended"] = SpatialFluxComponent.from_numpy(
        flux=inits["extended"], prior=GMMPatchPrior(gmm=gmm, stride=stride, marginalize=marginalize)
    )
    if n_comp > 1:
        second = InverseGammaPrior(alpha=10.0, beta=1.5)
        comps["points"] = SpatialFluxComponent.from_numpy(flux=inits["points"], prior=second)
    deco = MAPDeconvolver(n_epochs=n_epochs, display_progress=False, device=DEV, fit_mode="joint" if joint else "sequential")
    res = deco.run(datasets, components=comps)

    for name in names:
        got, ref, exact = res.components[name].flux_upsampled_numpy, final_o[name], final_64[name]
        norm = np.abs(ref).max()
        err = np.abs(got - ref) / norm
        d_gpu, d_o = np.abs(got - exact) / norm, np.abs(ref - exact) / norm  # distance of either fp32 path to float64
        q50, q99 = np.quantile(err, [0.5, 0.99])
        print(f"seed {seed}: {H}x{W} psf {kh}x{kw} obs {n_obs} comps {n_comp} K {K} stride {stride} joint {joint} lse {marginalize}: "
              f"{name} |gpu-oracle| median {q50:.1e} q99 {q99:.1e} max {err.max():.1e} n>1e-5 {int((err > 1e-5).sum())}; "
              f"|gpu-f64| max {d_gpu.max():.1e} q99 {np.quantile(d_gpu, 0.99):.1e}; |oracle-f64| max {d_o.max():.1e} q99 {np.quantile(d_o, 0.99):.1e}")
        # The bulk must agree to the north-star tolerance; the isolated ill-conditioned pixels (see the ensemble test
        # above for the float64 arbitration) are bounded at 2.5 x the largest deviation measured over the seeds (2.1e-4).
        assert q50 < 2e-6 and q99 < 5e-5, name
        assert err.max() < 5e-4, name
        _DISTANCES.setdefault(seed, []).append((d_gpu, d_o))
    for column in ("total", "datasets-total", "priors-total"):
        ref = np.array([row[column] for row in trace_o])
        np.testing.assert_allclose(np.asarray(res.trace_loss[column]), ref, rtol=2e-5, atol=1e-6, err_msg=column)


def test_random_fits_are_as_close_to_float64_as_the_fp32_oracle():
    """The part of |gpu - oracle32| above 1e-5 (isolated pixels where Adam's g / (|g| + 1e-8) is ill-conditioned) is
    arbitrated by the SAME oracle run in float64, over the ensemble of seeds: which of the two fp32 paths trips over
    such a pixel in a given scene is chance (seed 0: the oracle, 2.6e-4 vs 4.4e-5; seed 3: the HIP path, 1.9e-5 vs
    2.0e-6), so the comparison is made on the pooled distances."""
    for seed in range(8):
        if seed not in _DISTANCES:
            test_random_fit_matches_the_oracle(seed)
    d_gpu = np.concatenate([d[0].ravel() for seed in range(8) for d in _DISTANCES[seed]])
    d_o = np.concatenate([d[1].ravel() for seed in range(8) for d in _DISTANCES[seed]])
    qs = [0.5, 0.99, 0.999, 0.9999]
    q_gpu, q_o = np.quantile(d_gpu, qs), np.quantile(d_o, qs)
    print("pooled |gpu-f64|  quantiles", dict(zip(qs, q_gpu)), "max", d_gpu.max(), "n>1e-5", int((d_gpu > 1e-5).sum()))
    print("pooled |o32-f64| quantiles", dict(zip(qs, q_o)), "max", d_o.max(), "n>1e-5", int((d_o > 1e-5).sum()))
    assert np.all(q_gpu <= 2.0 * q_o + 1e-6)
    assert d_gpu.max() <= 2.0 * d_o.max()
    assert (d_gpu > 1e-5).sum() <= 2 * (d_o > 1e-5).sum() + 10
