"""GPU edge cases against the CPU oracle on identical inputs: tiny / ragged images, a single patch,
K not a multiple of the 4-way component split, strides other than 4, PSFs larger than the image,
1x1 PSFs, images whose width defeats the 16-byte fast paths."""
import numpy as np
import pytest
import torch

from conftest import assert_prior_grad_matches, rel_linf
from oracle import cpu_ref

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _gmm_pair(K, seed, stride, zero_means=False):
    from jolideco_amd.priors.patches import GaussianMixtureModel, GaussianMixtureModelMeta

    means, covs, weights = cpu_ref.synthetic_gmm(K, 64, seed=seed, zero_means=zero_means)
    gmm = GaussianMixtureModel.from_numpy(means, covs, weights, meta=GaussianMixtureModelMeta(stride=stride))
    return gmm, cpu_ref.GMM.from_numpy(means, covs, weights, stride=stride)


@pytest.mark.parametrize(
    "shape,K,stride,shifts",
    [((8, 8), 1, 4, (0, 0)), ((8, 8), 3, 4, (2, -2)), ((9, 13), 2, 4, (-1, 1)), ((31, 17), 5, 2, (1, 2)),
     ((40, 33), 7, 3, (-2, 0)), ((16, 200), 6, 4, (0, 1)), ((130, 12), 9, 1, (2, 2)), ((64, 64), 128, 4, (-2, -2)),
     # component counts of the reference's trained mixtures (zoran-weiss: 200, gmm.py:358-367) and one past a power of 2
     ((96, 120), 200, 4, (1, -3)), ((72, 72), 129, 4, (0, 2))],
)
def test_gmm_prior_edge_shapes(shape, K, stride, shifts):
    gmm, gmm_o = _gmm_pair(K, seed=shape[0] + K, stride=stride)
    rs = np.random.RandomState(shape[1])
    flux_np = (rs.gamma(2.5, size=shape) * 2).astype(np.float32)
    value_o, grad_o, arg_o = cpu_ref.gmm_prior_value_and_grad(flux_np, gmm_o, stride, shifts)
    flux = torch.from_numpy(flux_np).to(DEV)
    scale = (stride**2 / 64) / flux.numel()
    n_patches = ((shape[0] - 8) // stride + 1) * ((shape[1] - 8) // stride + 1)
    value, grad = torch.zeros(1, device=DEV), torch.zeros_like(flux)
    argmax = torch.full((n_patches,), -7, dtype=torch.int32, device=DEV)
    gmm.handle(DEV).prior_fwd_bwd(flux, stride, shifts, value, scale, grad=grad, grad_coef=scale, argmax_out=argmax)
    np.testing.assert_allclose(float(value), value_o, rtol=5e-6)
    got = argmax.cpu().numpy()
    assert got.min() >= 0 and got.max() < K
    # always compared: pixels under a flipped near-tie patch are masked, every other pixel held to 1e-5
    assert_prior_grad_matches(grad.cpu().numpy(), grad_o, got, arg_o, shape, stride, shifts)
    # logsumexp mode on the same input
    lse_o, glse_o, _ = cpu_ref.gmm_prior_value_and_grad(flux_np, gmm_o, stride, shifts, marginalize=True)
    v2, g2 = torch.zeros(1, device=DEV), torch.zeros_like(flux)
    gmm.handle(DEV).prior_fwd_bwd(flux, stride, shifts, v2, scale, grad=g2, grad_coef=scale, marginalize=True)
    np.testing.assert_allclose(float(v2), lse_o, rtol=5e-6)
    assert rel_linf(g2.cpu().numpy(), glse_o) < 1e-4


@pytest.mark.parametrize(
    "shape,kshape",
    [((12, 12), (17, 17)), ((5, 9), (3, 3)), ((33, 35), (1, 1)), ((64, 61), (6, 3)), ((10, 300), (33, 2)), ((257, 66), (2, 33))],
)
def test_fused_npred_poisson_edge_shapes(shape, kshape, conv_method):
    """PSF larger than the image, 1x1 PSF, widths that are not multiples of 4, even / thin PSFs."""
    from jolideco_amd import FluxComponents, NPredModels, SpatialFluxComponent
    from jolideco_amd.ops import stirling_mean

    rs = np.random.RandomState(shape[0] * 7 + kshape[1])
    psf = rs.uniform(0.1, 1.0, size=kshape).astype(np.float32)
    psf /= psf.sum()
    data = {
        "counts": rs.poisson(3.0, size=shape).astype(np.float32),
        "psf": psf,
        "exposure": rs.uniform(0.5, 1.5, size=shape).astype(np.float32),
        "background": rs.uniform(0.2, 1.0, size=shape).astype(np.float32),
    }
    theta = rs.normal(size=shape).astype(np.float32)
    loss_o, npred_o, grad_o = cpu_ref.poisson_loss_and_grad(theta, data)
    comps = FluxComponents()
    comps["flux"] = SpatialFluxComponent.from_numpy(flux=np.exp(theta))
    models = NPredModels.from_dataset_numpy(dataset=data, components=comps, device=DEV)
    from conftest import expected_plan_method
    from jolideco_amd.ops import psf_separable_rank

    assert models.plan.method == expected_plan_method(conv_method, psf_separable_rank(np.asarray(data["psf"])),
                                                      np.asarray(data["psf"]).shape)
    flux = torch.exp(torch.from_numpy(theta)).to(DEV)
    loss, grad, npred = torch.zeros(1, device=DEV), torch.zeros_like(flux), torch.empty_like(flux)
    models.fwd_bwd([flux], torch.from_numpy(data["counts"]).to(DEV), stirling_mean(data["counts"]), loss, grads=[grad],
                   npred_out=npred)
    assert rel_linf(npred.cpu().numpy(), npred_o) < 1e-5
    np.testing.assert_allclose(float(loss), loss_o, rtol=3e-6)
    assert rel_linf((grad * flux).cpu().numpy(), grad_o) < 2e-5


def test_zero_counts_and_huge_dynamic_range():
    """counts == 0 everywhere (Stirling term off, log of small npred) and a 1e6 dynamic range flux."""
    from jolideco_amd import MAPDeconvolver, SpatialFluxComponent
    from jolideco_amd.data import gaussian_kernel

    rs = np.random.RandomState(0)
    shape = (24, 40)
    data = {
        "counts": np.zeros(shape, np.float32),
        "psf": gaussian_kernel(1.5, (9, 9)).astype(np.float32),
        "exposure": np.ones(shape, np.float32),
        "background": np.full(shape, 1e-3, np.float32),
    }
    flux_init = np.exp(rs.uniform(-7, 7, size=shape))
    comp = SpatialFluxComponent.from_numpy(flux=flux_init)
    res = MAPDeconvolver(n_epochs=3, display_progress=False, device=DEV).run({"d": data}, components=comp)
    final, trace = cpu_ref.map_fit_sequential({"d": data}, {"flux": flux_init}, {"flux": cpu_ref.UniformPriorRef()}, n_epochs=3)
    assert np.isfinite(res.flux_total).all()
    assert rel_linf(res.flux_total, final["flux"]) < 1e-5
    np.testing.assert_allclose(res.trace_loss[-1]["total"], trace[-1]["total"], rtol=1e-5)


def test_shape_and_argument_errors():
    """Same error behaviour as the reference where it has one, loud errors otherwise."""
    from jolideco_amd import GMMPatchPrior, MAPDeconvolver, SpatialFluxComponent
    from jolideco_amd.ops import ConvPlan

    gmm, _ = _gmm_pair(2, 0, 4)
    with pytest.raises(RuntimeError, match="smaller than a patch"):
        flux = torch.ones((6, 20), device=DEV)
        gmm.handle(DEV).prior_fwd_bwd(flux, 4, (0, 0), torch.zeros(1, device=DEV), 1.0)
    with pytest.raises(ValueError, match="Early stopping requires"):
        MAPDeconvolver(n_epochs=1, stop_early=True, device=DEV, display_progress=False).run({}, components=None)
    with pytest.raises(ValueError, match="four dimensional"):
        SpatialFluxComponent(flux_upsampled=torch.ones(4, 4))
    plan = ConvPlan(16, 16, 3, 3, DEV)
    with pytest.raises(ValueError, match="does not match the plan"):
        plan.conv_same(torch.ones((8, 8), device=DEV), None, plan.psf_spectrum(torch.ones((3, 3), device=DEV)))
    with pytest.raises(RuntimeError, match="float32"):
        plan.psf_spectrum(torch.ones((3, 3), device=DEV, dtype=torch.float64))
    assert isinstance(GMMPatchPrior(gmm=gmm).to_dict(), dict)
