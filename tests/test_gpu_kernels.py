"""GPU parity of the individual HIP stages against the golden vectors generated from the reference
(tests/golden/stages.npz) -- every call goes through the C-ABI of libjolideco_hip.so.

Tolerances: fp32 arithmetic, different (FFT grid / summation) order than the reference CPU path:
relative L-inf <= 1e-5 on images, rtol 1e-6..1e-5 on scalars (BASELINE.json: 1e-5 rel. L-inf).
"""
import numpy as np
import pytest
import torch

from conftest import assert_prior_grad_matches, rel_linf, unpack_datasets

pytestmark = pytest.mark.gpu

CASES = ["sq96_psf17", "rect80x112_psf12x16", "rect97x110_psf9x5", "sq256_psf33"]
DEV = "cuda:0"


def _case(stages, name):
    sub = {k[len(name) + 1:]: v for k, v in stages.items() if k.startswith(name + "/")}
    data = unpack_datasets(sub)["d"]
    return sub, data


@pytest.mark.parametrize("name", CASES)
def test_fused_npred_poisson_fwd_bwd(golden, name, conv_method):
    """jd_npred_poisson_fwd_bwd == NPredModels.evaluate + PoissonNLLLoss + autograd of the reference."""
    from jolideco_amd import FluxComponents, NPredModels, SpatialFluxComponent
    from jolideco_amd.ops import stirling_mean

    sub, data = _case(golden("stages"), name)
    theta = torch.from_numpy(sub["theta"]).to(DEV)
    flux = torch.exp(theta)
    comps = FluxComponents()
    comps["flux"] = SpatialFluxComponent.from_numpy(flux=np.exp(sub["theta"]))
    models = NPredModels.from_dataset_numpy(dataset=data, components=comps, device=DEV)
    from conftest import expected_plan_method
    from jolideco_amd.ops import psf_separable_rank

    assert models.plan.method == expected_plan_method(conv_method, psf_separable_rank(np.asarray(data["psf"])),
                                                      np.asarray(data["psf"]).shape)
    # edge corrected exposure (models/npred.py:108-113)
    assert rel_linf(models["flux"].exposure.cpu().numpy()[0, 0], sub["exposure_corrected"]) < 2e-6

    counts = torch.from_numpy(data["counts"]).to(DEV)
    loss = torch.zeros(1, device=DEV)
    grad = torch.full_like(flux, 7.0)  # must be overwritten (accumulate=False)
    npred = torch.empty_like(flux)
    models.fwd_bwd([flux], counts, stirling_mean(data["counts"]), loss, grads=[grad], npred_out=npred)
    torch.cuda.synchronize()
    assert rel_linf(npred.cpu().numpy(), sub["npred"]) < 1e-5
    np.testing.assert_allclose(float(loss), float(sub["loss"]), rtol=2e-6)
    grad_theta = (grad * flux).cpu().numpy()  # chain rule of exp, models/core.py:588-589
    assert rel_linf(grad_theta, sub["grad_theta"]) < 1e-5

    # accumulate=True adds a second copy; forward-only leaves grads alone
    models.fwd_bwd([flux], counts, stirling_mean(data["counts"]), loss, grads=[grad], accumulate=True, grad_scale=0.5)
    assert rel_linf((grad * flux).cpu().numpy(), 1.5 * sub["grad_theta"]) < 1e-5
    before = grad.clone()
    loss2 = torch.zeros(1, device=DEV)
    models.fwd_bwd([flux], counts, stirling_mean(data["counts"]), loss2)
    assert torch.equal(before, grad)
    np.testing.assert_allclose(float(loss2), float(sub["loss"]), rtol=2e-6)


@pytest.mark.parametrize("name", CASES)
def test_autograd_seams_match_fused_path(golden, name, conv_method):
    """NPredModels.evaluate + loss_function + backward (the reference's own loop structure)."""
    from jolideco_amd import FluxComponents, PoissonLoss, SpatialFluxComponent

    sub, data = _case(golden("stages"), name)
    comps = FluxComponents()
    comps["flux"] = SpatialFluxComponent.from_numpy(flux=np.exp(sub["theta"]))
    comps = comps.to(DEV)
    with torch.no_grad():
        comps["flux"]._flux_upsampled.copy_(torch.from_numpy(sub["theta"])[None, None])
    ploss = PoissonLoss.from_datasets({"d": data}, components=comps, device=DEV)
    fluxes = comps.to_flux_tuple()
    counts, model = next(ploss.iter_by_dataset)
    npred = model.evaluate(fluxes=fluxes)
    loss = ploss.loss_function(npred, counts)
    loss.backward()
    assert rel_linf(npred.detach().cpu().numpy()[0, 0], sub["npred"]) < 1e-5
    np.testing.assert_allclose(float(loss), float(sub["loss"]), rtol=2e-6)
    assert rel_linf(comps["flux"]._flux_upsampled.grad.cpu().numpy()[0, 0], sub["grad_theta"]) < 1e-5


def _gmm(stages, gname, stride=4):
    from jolideco_amd.priors.patches import GaussianMixtureModel, GaussianMixtureModelMeta

    return GaussianMixtureModel.from_numpy(
        means=stages[f"gmm/{gname}/means"], covariances=stages[f"gmm/{gname}/covariances"],
        weights=stages[f"gmm/{gname}/weights"], meta=GaussianMixtureModelMeta(stride=stride),
    )


def _prior_keys(stages, name, gname, mode):
    prefix = f"{name}/prior/{gname}/"
    out = []
    for key in stages:
        if key.startswith(prefix) and key.endswith(f"/{mode}/value"):
            shift = key[len(prefix):].split("/")[0]
            sy, sx = (int(v) for v in shift[1:].split("_"))
            out.append((prefix + shift + f"/{mode}", (sy, sx)))
    return out


@pytest.mark.parametrize("gname", ["k16", "k5m"])
@pytest.mark.parametrize("name", CASES)
def test_gmm_prior_max_value_grad_argmax(golden, name, gname):
    """jd_gmm_prior_fwd_bwd (max mode) == GMMPatchPrior.__call__ + autograd of the reference."""
    stages = golden("stages")
    gmm = _gmm(stages, gname)
    handle = gmm.handle(DEV)
    flux = torch.exp(torch.from_numpy(stages[f"{name}/theta"])).to(DEV)
    H, W = flux.shape
    scale = (4 * 4 / 64) / (H * W)
    keys = _prior_keys(stages, name, gname, "max")
    assert keys
    for key, shifts in keys:
        value = torch.zeros(1, device=DEV)
        grad = torch.zeros_like(flux)
        n_patches = ((H - 8) // 4 + 1) * ((W - 8) // 4 + 1)
        argmax = torch.full((n_patches,), -5, dtype=torch.int32, device=DEV)
        handle.prior_fwd_bwd(flux, 4, shifts, value, scale, grad=grad, grad_coef=scale, argmax_out=argmax)
        torch.cuda.synchronize()
        np.testing.assert_allclose(float(value), float(stages[f"{key}/value"]), rtol=3e-6)
        ref_arg, margin = stages[f"{key}/argmax"], stages[f"{key}/margin"]
        got_arg = argmax.cpu().numpy()
        clear = margin > 1e-3  # near-ties may flip under re-ordered fp32 sums
        assert np.array_equal(got_arg[clear], ref_arg[clear])
        assert (got_arg != ref_arg).sum() <= 2
        # the gradient is ALWAYS compared: pixels under a flipped near-tie patch are masked, the rest held to 1e-5
        assert_prior_grad_matches(grad.cpu().numpy(), stages[f"{key}/grad_flux"], got_arg, ref_arg, (H, W), 4, shifts,
                                  margin=margin)


@pytest.mark.parametrize("name", CASES[:3])
def test_gmm_prior_marginalized_value(golden, name):
    """logsumexp mode (priors/patches/core.py:242-243): forward value and gradient
    (sum_k responsibility_k * gamma_k).  The fixture's flux is float32(gamma * 3) while the kernels see
    exp(log(.)), a <= 1 ulp difference that the responsibilities of near-tie components amplify to
    ~1e-4 (the CPU oracle shows the same deviation, tests/test_oracle_golden.py), hence 5e-4."""
    stages = golden("stages")
    for gname in ("k16", "k5m"):
        handle = _gmm(stages, gname).handle(DEV)
        flux = torch.exp(torch.from_numpy(stages[f"{name}/theta"])).to(DEV)
        H, W = flux.shape
        scale = (4 * 4 / 64) / (H * W)
        for key, shifts in _prior_keys(stages, name, gname, "lse"):
            value = torch.zeros(1, device=DEV)
            handle.prior_fwd_bwd(flux, 4, shifts, value, scale, marginalize=True)
            np.testing.assert_allclose(float(value), float(stages[f"{key}/value"]), rtol=3e-6)
            value2, grad = torch.zeros(1, device=DEV), torch.zeros_like(flux)
            handle.prior_fwd_bwd(flux, 4, shifts, value2, scale, grad=grad, grad_coef=scale, marginalize=True)
            assert torch.equal(value, value2)
            assert rel_linf(grad.cpu().numpy(), stages[f"{key}/grad_flux"]) < 5e-4


def test_gmm_prior_patch_row_shards_sum_to_whole(golden):
    """Sharding the prior by patch rows (multi-GPU split) reproduces the unsharded value/gradient."""
    stages = golden("stages")
    name, gname = "rect97x110_psf9x5", "k16"
    handle = _gmm(stages, gname).handle(DEV)
    flux = torch.exp(torch.from_numpy(stages[f"{name}/theta"])).to(DEV)
    H, W = flux.shape
    n_rows = (H - 8) // 4 + 1
    scale = (16 / 64) / (H * W)
    whole_v, whole_g = torch.zeros(1, device=DEV), torch.zeros_like(flux)
    handle.prior_fwd_bwd(flux, 4, (-2, 1), whole_v, scale, grad=whole_g, grad_coef=-0.3 * scale)
    part_v, part_g = torch.zeros(1, device=DEV), torch.zeros_like(flux)
    bounds = [0, 5, 5, 13, n_rows]  # includes an empty shard
    for lo, hi in zip(bounds[:-1], bounds[1:]):
        handle.prior_fwd_bwd(
            flux, 4, (-2, 1), part_v, scale, grad=part_g, grad_coef=-0.3 * scale, patch_rows=(lo, hi),
            accumulate_value=True,
        )
    np.testing.assert_allclose(float(part_v), float(whole_v), rtol=1e-6)
    assert rel_linf(part_g.cpu().numpy(), whole_g.cpu().numpy()) < 1e-6


def test_gmm_estimate_log_prob_matches_sklearn_formula():
    """GaussianMixtureModel.estimate_log_prob vs an fp64 evaluation of the same formula; with
    meta.stride=None the pixel weights are 1 and this is sklearn's _estimate_weighted_log_prob
    (reference test priors/patches/tests/test_gmm.py:10-35)."""
    from jolideco_amd.data import synthetic_gmm
    from jolideco_amd.priors.patches import GaussianMixtureModel, GaussianMixtureModelMeta

    means, covs, weights = synthetic_gmm(7, 64, seed=3)
    means = 0.05 * np.random.RandomState(1).normal(size=means.shape)
    for stride in (None, 4):
        gmm = GaussianMixtureModel.from_numpy(means, covs, weights, meta=GaussianMixtureModelMeta(stride=stride))
        x = np.random.RandomState(0).normal(size=(133, 64)).astype(np.float32)
        got = gmm.estimate_log_prob(torch.from_numpy(x).to(DEV)).cpu().numpy()
        pc = gmm.precisions_cholesky_numpy.astype(np.float64)
        w = gmm.pixel_weights_numpy.astype(np.float64)
        ref = np.empty((133, 7))
        for k in range(7):
            y = x.astype(np.float64) @ pc[k] - means[k].astype(np.float32).astype(np.float64) @ pc[k]
            ref[:, k] = -0.5 * (64 * np.log(2 * np.pi) + (y * y * w).sum(1)) + np.log(np.diag(pc[k])).sum() + np.log(
                weights[k]
            )
        np.testing.assert_allclose(got, ref, rtol=2e-5, atol=2e-4)
        if stride is None:
            from sklearn.mixture import GaussianMixture
            from sklearn.mixture._gaussian_mixture import _compute_precision_cholesky

            sk = GaussianMixture()
            sk.weights_, sk.covariances_, sk.means_ = weights, covs, means
            sk.precisions_cholesky_ = _compute_precision_cholesky(covs, "full")
            np.testing.assert_allclose(got, sk._estimate_weighted_log_prob(X=x), rtol=2e-5, atol=2e-4)


def test_convolve_fft_torch_known_answers(conv_method):
    """Reference test utils/tests/test_torch.py:24-41: box image * normalised box kernel."""
    from scipy.signal import convolve2d

    from jolideco_amd.utils.torch import convolve_fft_torch

    image = np.zeros((9, 9), dtype=np.float32)
    image[3:6, 3:6] = 1
    kernel = np.ones((3, 3), dtype=np.float32) / 9
    ref = convolve2d(image, kernel, mode="same")
    out = convolve_fft_torch(torch.from_numpy(image[None, None]).to(DEV), torch.from_numpy(kernel[None, None]).to(DEV))
    np.testing.assert_allclose(out.cpu().numpy()[0, 0], ref, atol=2e-7)
    # even-sized, asymmetric kernel on a non-square image: same crop as scipy's "same"? no --
    # the reference crops at (k-1)//2, check against an explicit full convolution
    rs = np.random.RandomState(0)
    image = rs.uniform(size=(21, 30)).astype(np.float32)
    kernel = rs.uniform(size=(4, 6)).astype(np.float32)
    full = convolve2d(image.astype(np.float64), kernel.astype(np.float64), mode="full")
    ref = full[1 : 1 + 21, 2 : 2 + 30]
    out = convolve_fft_torch(torch.from_numpy(image[None, None]).to(DEV), torch.from_numpy(kernel[None, None]).to(DEV))
    np.testing.assert_allclose(out.cpu().numpy()[0, 0], ref, rtol=1e-5, atol=1e-5)


def test_adam_step_matches_torch_optim():
    """jd_adam_step == torch.optim.Adam on theta with grad = grad_flux * exp(theta), several steps."""
    from jolideco_amd import _hip
    from jolideco_amd._hip import check, ptr, stream_ptr
    from jolideco_amd.ops import adam_bias_terms

    rs = np.random.RandomState(5)
    n = 1000 * 4 + 3  # exercises the scalar tail path
    theta0 = rs.normal(size=n).astype(np.float32)
    ref_p = torch.nn.Parameter(torch.from_numpy(theta0.copy()))
    opt = torch.optim.Adam([ref_p], lr=0.1)
    theta = torch.from_numpy(theta0.copy()).to(DEV)
    flux = torch.exp(theta)
    flux2 = torch.empty_like(flux)
    m, v = torch.zeros_like(theta), torch.zeros_like(theta)
    for step in range(1, 6):
        g = rs.normal(size=n).astype(np.float32)
        ref_p.grad = torch.from_numpy(g) * torch.exp(ref_p.detach())
        opt.step()
        grad = torch.from_numpy(g).to(DEV)
        ss, b2 = adam_bias_terms(step, 0.1, 0.9, 0.999)
        check(_hip.lib().jd_adam_step(ptr(theta), ptr(flux), ptr(flux2), ptr(grad), ptr(m), ptr(v), None, n, ss, 0.9,
                                      0.999, 1 - 0.9, 1 - 0.999, b2, 1e-8, 1, 1, None, stream_ptr()))
        flux, flux2 = flux2, flux
        assert float(grad.abs().max()) == 0.0
        np.testing.assert_allclose(theta.cpu().numpy(), ref_p.detach().numpy(), rtol=2e-6, atol=2e-6)
        np.testing.assert_allclose(flux.cpu().numpy(), np.exp(theta.cpu().numpy()), rtol=2e-6)


def test_product_path_fails_loudly_without_gpu_tensors():
    from jolideco_amd.utils.torch import convolve_fft_torch

    with pytest.raises(RuntimeError):
        convolve_fft_torch(torch.zeros(1, 1, 8, 8), torch.ones(1, 1, 3, 3))


@pytest.mark.parametrize(
    "shape,kshape",
    [((64, 64), (1, 1)), ((70, 130), (17, 17)), ((129, 67), (33, 33)), ((50, 200), (2, 31)), ((203, 61), (32, 3)),
     ((256, 256), (5, 16))],
)
@pytest.mark.parametrize("direct_kernel", ["split_fp16", "fp32"])
def test_direct_conv_matches_fft_and_float64(shape, kshape, direct_kernel, jd_option):
    """The MFMA Toeplitz kernels (split-fp16 x 3, the default where its window planes and fragment table fit in LDS --
    not at 33 x 33 -- and the fp32 one, JD_DIRECT_FP32=1), rocFFT on the fast grid and rocFFT on the reference's exact
    grid all compute the same 'same' convolution and its adjoint (ragged tiles, even / 1-pixel PSFs)."""
    if direct_kernel == "fp32":
        jd_option("JD_DIRECT_FP32", "1")
    from scipy.signal import convolve2d

    from jolideco_amd.ops import ConvPlan

    rs = np.random.RandomState(sum(shape) + sum(kshape))
    image = rs.gamma(2.0, size=shape).astype(np.float32)
    scale = rs.uniform(0.5, 1.5, size=shape).astype(np.float32)
    psf = rs.uniform(size=kshape).astype(np.float32)
    psf /= psf.sum()
    grad_out = rs.normal(size=shape).astype(np.float32)
    kh, kw = kshape
    full = convolve2d((image * scale).astype(np.float64), psf.astype(np.float64), mode="full")
    oy, ox = (kh - 1) // 2, (kw - 1) // 2
    ref = full[oy : oy + shape[0], ox : ox + shape[1]]
    t = lambda a: torch.from_numpy(a).to(DEV)  # noqa: E731
    results = {}
    for method in ("direct", "fft", "fft-exact"):
        plan = ConvPlan(shape[0], shape[1], kh, kw, DEV, method=method)
        assert plan.method == ("direct" if method == "direct" else "fft")
        khat = plan.psf_spectrum(t(psf))
        out = plan.conv_same(t(image), t(scale), khat)
        adj = plan.conv_same_adjoint(t(grad_out), t(scale), khat)
        acc = torch.full(shape, 2.0, device=DEV)
        plan.conv_same_adjoint(t(grad_out), t(scale), khat, grad_image=acc, accumulate=True)
        torch.cuda.synchronize()
        assert rel_linf(out.cpu().numpy(), ref) < (2e-6 if method == "direct" else 1e-5), method
        # <conv(u), g> == <u, adj(g)>  (adjoint identity in float64 on the host)
        lhs = float((out.double().cpu() * torch.from_numpy(grad_out).double()).sum())
        rhs = float((torch.from_numpy(image).double() * adj.double().cpu()).sum())
        assert abs(lhs - rhs) < 1e-4 * max(1.0, abs(lhs)), method
        assert rel_linf((acc - 2.0).cpu().numpy(), adj.cpu().numpy()) < 1e-5
        results[method] = (out.cpu().numpy(), adj.cpu().numpy())
        plan.close()
    for method in ("fft", "fft-exact"):
        assert rel_linf(results[method][0], results["direct"][0]) < 1e-5
        assert rel_linf(results[method][1], results["direct"][1]) < 1e-5


def test_split_direct_conv_keeps_faint_pixels_next_to_bright_sources():
    """The split-fp16 direct kernel scales every 80 x 80 window by ONE power of two; its lo planes carry a factor
    2^11, so a pixel keeps its 22 bits while it is more than 3.7e-9 of the brightest pixel of its window.  A sky of
    0.01 with point sources of 1e5 (ratio 1e7, more than a deconvolved image of a bright point source on a faint
    background shows): every output pixel -- also the faint ones between the sources, where a maximum-normalised error
    would hide anything -- must agree with the float64 convolution RELATIVE TO ITS OWN VALUE."""
    from scipy.signal import convolve2d

    from jolideco_amd.ops import ConvPlan

    rs = np.random.RandomState(12)
    shape = (192, 256)
    image = (0.01 * rs.uniform(0.5, 1.5, size=shape)).astype(np.float32)
    for _ in range(12):
        image[rs.randint(0, shape[0]), rs.randint(0, shape[1])] = 1e5 * rs.uniform(0.5, 2.0)
    scale = rs.uniform(0.5, 1.5, size=shape).astype(np.float32)
    g = np.exp(-0.5 * ((np.arange(17) - 8) / 2.0) ** 2)
    psf = np.outer(g, g) * (1.0 + 0.2 * rs.uniform(-1, 1, size=(17, 17)))  # not separable
    psf = (psf / psf.sum()).astype(np.float32)
    ref = convolve2d((image * scale).astype(np.float64), psf.astype(np.float64), mode="full")[8:-8, 8:-8]
    plan = ConvPlan(shape[0], shape[1], 17, 17, DEV, method="direct")
    khat = plan.psf_spectrum(torch.from_numpy(psf).to(DEV))
    out = plan.conv_same(torch.from_numpy(image).to(DEV), torch.from_numpy(scale).to(DEV), khat).cpu().numpy()
    plan.close()
    local = np.abs(out - ref) / ref
    print("split direct conv, dynamic range 1e7: worst error relative to the pixel's own value", local.max(),
          "at a pixel of", ref.flat[local.argmax()], "; output range", ref.min(), ref.max())
    assert local.max() < 3e-6


def _gauss(n, sigma, offset=0.0):
    x = np.arange(n) - (n - 1) / 2 - offset
    return np.exp(-0.5 * (x / sigma) ** 2)


@pytest.mark.parametrize(
    "shape,kshape,rank",
    [((70, 130), (17, 17), 1), ((129, 67), (33, 33), 2), ((50, 200), (2, 31), 1), ((203, 61), (32, 3), 3),
     ((256, 256), (13, 16), 1), ((64, 64), (1, 1), 1), ((96, 260), (34, 34), 1), ((40, 40), (68, 5), 2),
     ((31, 33), (25, 25), 3)],
)
def test_separable_conv_matches_direct_and_float64(shape, kshape, rank):
    """Low-rank PSFs (sums of `rank` outer products, asymmetric and off-centre on purpose): the separable kernel
    computes the same 'same' convolution and adjoint as float64 scipy and as the MFMA / rocFFT paths -- ragged
    tiles, even sizes, widths that need the alignment shift, images narrower than one tile, W % 4 != 0."""
    from scipy.signal import convolve2d

    from jolideco_amd.ops import ConvPlan, psf_separable_rank

    kh, kw = kshape
    rs = np.random.RandomState(sum(shape) + sum(kshape) + rank)
    psf = np.zeros(kshape)
    for r in range(rank):
        u = _gauss(kh, 1.0 + 2.0 * r, 0.4 * r) * (1 + 0.3 * np.tanh(np.arange(kh) - kh / 2))
        v = _gauss(kw, 1.5 + 1.5 * r, -0.7 * r) * (1 + 0.2 * np.sin(np.arange(kw)))
        psf += rs.uniform(0.3, 1.0) * np.outer(u, v)
    psf = (psf / psf.sum()).astype(np.float32)
    assert psf_separable_rank(psf) == min(rank, kh, kw)
    image = rs.gamma(2.0, size=shape).astype(np.float32)
    scale = rs.uniform(0.5, 1.5, size=shape).astype(np.float32)
    grad_out = rs.normal(size=shape).astype(np.float32)
    oy, ox = (kh - 1) // 2, (kw - 1) // 2
    ref = convolve2d((image * scale).astype(np.float64), psf.astype(np.float64), mode="full")[oy : oy + shape[0], ox : ox + shape[1]]
    # adjoint in float64: correlation, then the exposure
    pad = np.zeros((shape[0] + kh - 1, shape[1] + kw - 1))
    pad[oy : oy + shape[0], ox : ox + shape[1]] = grad_out
    ref_adj = convolve2d(pad, psf[::-1, ::-1].astype(np.float64), mode="valid") * scale
    t = lambda a: torch.from_numpy(a).to(DEV)  # noqa: E731

    plan = ConvPlan(shape[0], shape[1], kh, kw, DEV, method="separable")
    assert plan.method == "separable" and (plan.Hp, plan.Wp) == shape
    khat = plan.psf_spectrum(t(psf))
    out = plan.conv_same(t(image), t(scale), khat)
    adj = plan.conv_same_adjoint(t(grad_out), t(scale), khat)
    acc = torch.full(shape, 2.0, device=DEV)
    plan.conv_same_adjoint(t(grad_out), t(scale), khat, grad_image=acc, accumulate=True)
    no_scale = plan.conv_same(t(image), None, khat)
    torch.cuda.synchronize()
    assert rel_linf(out.cpu().numpy(), ref) < 2e-6
    assert rel_linf(adj.cpu().numpy(), ref_adj) < 2e-6
    assert rel_linf((acc - 2.0).cpu().numpy(), ref_adj) < 2e-6
    ref_ns = convolve2d(image.astype(np.float64), psf.astype(np.float64), mode="full")[oy : oy + shape[0], ox : ox + shape[1]]
    assert rel_linf(no_scale.cpu().numpy(), ref_ns) < 2e-6
    plan.close()
    other = ConvPlan(shape[0], shape[1], kh, kw, DEV, method="auto")  # direct up to 33x33, rocFFT beyond
    khat_o = other.psf_spectrum(t(psf))
    assert rel_linf(other.conv_same(t(image), t(scale), khat_o).cpu().numpy(), out.cpu().numpy()) < 1e-5
    assert rel_linf(other.conv_same_adjoint(t(grad_out), t(scale), khat_o).cpu().numpy(), adj.cpu().numpy()) < 1e-5
    other.close()


def test_separable_plan_refuses_a_general_psf():
    from jolideco_amd.ops import ConvPlan, psf_separable_rank

    rs = np.random.RandomState(0)
    psf = rs.uniform(size=(9, 9)).astype(np.float32)
    assert psf_separable_rank(psf) == 0
    plan = ConvPlan(32, 32, 9, 9, DEV, method="separable")
    with pytest.raises(RuntimeError, match="outer products"):
        plan.psf_spectrum(torch.from_numpy(psf).to(DEV))
    plan.close()
    with pytest.raises(RuntimeError, match="68x68"):
        ConvPlan(96, 96, 69, 3, DEV, method="separable")


def test_large_psf_falls_back_to_fft():
    from jolideco_amd.ops import ConvPlan

    plan = ConvPlan(96, 96, 41, 41, DEV)
    assert plan.method == "fft"
    plan.close()
    with pytest.raises(RuntimeError, match="33x33"):
        ConvPlan(96, 96, 41, 41, DEV, method="direct")


def test_gmm_triangular_skip_is_bit_identical_to_dense(golden, jd_option):
    """The block-skipping (upper triangular P) and the dense variants of the GMM kernels give the
    same bits: the skipped terms are exact zeros at the end of every fmaf chain."""
    from jolideco_amd import _hip

    stages = golden("stages")
    name, gname = "rect80x112_psf12x16", "k5m"
    gmm = _gmm(stages, gname)
    handle = gmm.handle(DEV)
    assert _hip.lib().jd_gmm_is_triangular(handle._handle) == 1
    flux = torch.exp(torch.from_numpy(stages[f"{name}/theta"])).to(DEV)
    H, W = flux.shape
    scale = (16 / 64) / (H * W)
    n_patches = ((H - 8) // 4 + 1) * ((W - 8) // 4 + 1)
    out = {}
    for variant in ("tri", "dense"):
        if variant == "dense":
            jd_option("JD_GMM_DENSE", "1")
        value, grad = torch.zeros(1, device=DEV), torch.zeros_like(flux)
        argmax = torch.zeros(n_patches, dtype=torch.int32, device=DEV)
        handle.prior_fwd_bwd(flux, 4, (1, -2), value, scale, grad=grad, grad_coef=scale, argmax_out=argmax)
        lse = torch.zeros(1, device=DEV)
        handle.prior_fwd_bwd(flux, 4, (1, -2), lse, scale, marginalize=True)
        x = torch.from_numpy(np.random.RandomState(3).normal(size=(70, 64)).astype(np.float32)).to(DEV)
        out[variant] = (value.cpu(), grad.cpu(), argmax.cpu(), lse.cpu(), handle.estimate_log_prob(x).cpu())
    for a, b in zip(out["tri"], out["dense"]):
        assert torch.equal(a, b)


def test_gmm_non_triangular_precisions_use_the_dense_variant():
    """A user supplied, non-triangular precisions_cholesky (the constructor accepts any matrix,
    patches/gmm.py:64-117) must not take the block-skipping path."""
    from jolideco_amd import _hip
    from jolideco_amd.priors.patches import GaussianMixtureModel, GaussianMixtureModelMeta

    rs = np.random.RandomState(8)
    K = 6
    pc = rs.normal(size=(K, 64, 64)).astype(np.float32) * 0.3 + 2 * np.eye(64, dtype=np.float32)
    means = (0.1 * rs.normal(size=(K, 64))).astype(np.float32)
    weights = rs.dirichlet(np.ones(K)).astype(np.float32)
    gmm = GaussianMixtureModel(means, np.zeros((K, 64, 64), np.float32), weights, pc, meta=GaussianMixtureModelMeta(stride=4))
    assert _hip.lib().jd_gmm_is_triangular(gmm.handle(DEV)._handle) == 0
    x = rs.normal(size=(100, 64)).astype(np.float32)
    got = gmm.estimate_log_prob(torch.from_numpy(x).to(DEV)).cpu().numpy()
    w = gmm.pixel_weights_numpy.astype(np.float64)
    ref = np.empty((100, K))
    for k in range(K):
        y = (x.astype(np.float64) - means[k]) @ pc[k].astype(np.float64)
        ref[:, k] = -0.5 * (64 * np.log(2 * np.pi) + (y * y * w).sum(1)) + np.log(np.diag(pc[k])).sum() + np.log(weights[k])
    np.testing.assert_allclose(got, ref, rtol=2e-5, atol=5e-4)


@pytest.mark.parametrize("path", ["screened", "dense"])
def test_gmm_prior_marginalized_gradient_exact_inputs(jd_option, path):
    """logsumexp gradient against the CPU oracle on bit-identical inputs (no exp(log(.)) round trip) -- through the
    screen (default: the components within 25 of the bound, their records combined per patch) and through the dense
    kernels (JD_GMM_LSE_SCREEN=0).
    Tolerance 5e-5 instead of the 1e-5 of the max mode: the responsibilities exponentiate the ABSOLUTE
    fp32 error of the log-likelihoods (|l| ~ 1e2..1e3, so ~1e-5 per component whatever the summation
    order), which no fp32 implementation can avoid; measured 1.9e-5 here."""
    from jolideco_amd.data import synthetic_gmm
    from jolideco_amd.priors.patches import GaussianMixtureModel, GaussianMixtureModelMeta
    from oracle import cpu_ref

    jd_option("JD_GMM_LSE_SCREEN", "1" if path == "screened" else "0")
    rs = np.random.RandomState(12)
    means, covs, weights = synthetic_gmm(9, 64, seed=6)
    means = 0.05 * rs.normal(size=means.shape)
    flux_np = (rs.gamma(3, size=(70, 91)) * 2).astype(np.float32)
    gmm_o = cpu_ref.GMM.from_numpy(means, covs, weights, stride=4)
    value_o, grad_o, _ = cpu_ref.gmm_prior_value_and_grad(flux_np, gmm_o, 4, (-1, 2), marginalize=True)
    gmm = GaussianMixtureModel.from_numpy(means, covs, weights, meta=GaussianMixtureModelMeta(stride=4))
    handle = gmm.handle(DEV)
    flux = torch.from_numpy(flux_np).to(DEV)
    scale = (16 / 64) / flux.numel()
    value, grad = torch.zeros(1, device=DEV), torch.zeros_like(flux)
    handle.prior_fwd_bwd(flux, 4, (-1, 2), value, scale, grad=grad, grad_coef=scale, marginalize=True)
    np.testing.assert_allclose(float(value), value_o, rtol=3e-6)
    assert rel_linf(grad.cpu().numpy(), grad_o) < 5e-5
    # the float64 oracle arbitrates the part above 1e-5: the kernel must be as close to it as the fp32 oracle is
    with cpu_ref.precision(np.float64):
        gmm_64 = cpu_ref.GMM.from_numpy(means, covs, weights, stride=4)
        _, grad_64, _ = cpu_ref.gmm_prior_value_and_grad(flux_np, gmm_64, 4, (-1, 2), marginalize=True)
    d_gpu, d_o = rel_linf(grad.cpu().numpy(), grad_64), rel_linf(grad_o, grad_64)
    print(f"lse gradient: |gpu-oracle32| {rel_linf(grad.cpu().numpy(), grad_o):.2e} |gpu-f64| {d_gpu:.2e} |oracle32-f64| {d_o:.2e}")
    assert d_gpu <= 2.0 * d_o + 2e-6
    # patch-row shards accumulate to the same gradient
    n_rows = (70 - 8) // 4 + 1
    pv, pg = torch.zeros(1, device=DEV), torch.zeros_like(flux)
    for lo, hi in ((0, 6), (6, 6), (6, n_rows)):
        handle.prior_fwd_bwd(flux, 4, (-1, 2), pv, scale, grad=pg, grad_coef=scale, marginalize=True,
                             patch_rows=(lo, hi), accumulate_value=True)
    np.testing.assert_allclose(float(pv), float(value), rtol=1e-6)
    assert rel_linf(pg.cpu().numpy(), grad.cpu().numpy()) < 1e-6


def test_gmm_logsumexp_screen_equals_the_dense_kernels(jd_option):
    """marginalize=True through the screen against the dense logsumexp kernel (both are gated against the oracle and
    its float64 run above): value to 2e-6, gradient to 5e-5 of its largest entry (two fp32 evaluations of the
    responsibilities: see the tolerance note of the oracle test) -- on a noisy image, a smooth one (patches with more
    candidates than a patch may keep are marked and evaluated by the dense kernel), one that is half of each, a
    patch-row shard, with filtered patches; the dense kernel's own bits where it takes every patch: 40 equal components
    (every patch marked) and an infinite pixel (the pass falls back on the device)."""
    from jolideco_amd.data import synthetic_gmm
    from jolideco_amd.priors.patches import GaussianMixtureModel, GaussianMixtureModelMeta

    shape = (136, 172)
    rs = np.random.RandomState(8)
    yy, xx = np.mgrid[: shape[0], : shape[1]]
    smooth = (50.0 + 30.0 * np.exp(-((yy - 60.0) ** 2 + (xx - 90.0) ** 2) / 800.0)).astype(np.float32)
    noisy = rs.gamma(20, size=shape).astype(np.float32)
    noisy[60:64, 80:90] = -2e5  # filtered patches: no value, no gradient

    def run(handle, image, screen, rows=(0, -1)):
        jd_option("JD_GMM_LSE_SCREEN", "2" if screen else "0")  # (2: every pass, whatever the earlier ones did)
        flux = torch.from_numpy(image).to(DEV)
        value, grad = torch.zeros(1, device=DEV), torch.zeros_like(flux)
        handle.prior_fwd_bwd(flux, 4, (3, -5), value, 0.25, grad=grad, grad_coef=-0.7, patch_rows=rows, marginalize=True)
        torch.cuda.synchronize()
        return float(value), grad.cpu().numpy()

    means, covs, weights = synthetic_gmm(24, 64, seed=9)
    handle = GaussianMixtureModel.from_numpy(means, covs, weights, meta=GaussianMixtureModelMeta(stride=4)).handle(DEV)
    half = np.where(xx < shape[1] // 2, smooth, noisy).astype(np.float32)
    for image in (noisy, smooth, half):
        for rows in ((0, -1), (5, 19)):
            for _ in range(3):  # (the record buffer of a handle grows over the first passes: all of them must be right)
                a, b = run(handle, image, True, rows), run(handle, image, False, rows)
                np.testing.assert_allclose(a[0], b[0], rtol=2e-6)
                assert rel_linf(a[1], b[1]) < 5e-5 and np.abs(b[1]).max() > 0
    # fallbacks on the device: the gated dense kernels take the pass -> the same bits as the dense path
    means, covs, weights = synthetic_gmm(1, 64, seed=3)
    gmm40 = GaussianMixtureModel.from_numpy(np.repeat(means, 40, axis=0), np.repeat(covs, 40, axis=0), np.full(40, 1 / 40),
                                            meta=GaussianMixtureModelMeta(stride=4))
    a, b = run(gmm40.handle(DEV), noisy, True), run(gmm40.handle(DEV), noisy, False)
    np.testing.assert_allclose(a[0], b[0], rtol=2e-6)  # (the per-patch values are summed in another order)
    assert np.array_equal(a[1], b[1])
    bad = noisy.copy()
    bad[20, 30] = np.inf
    a, b = run(handle, bad, True), run(handle, bad, False)
    assert np.array_equal(np.isnan(a[1]), np.isnan(b[1]))
    finite = np.isfinite(b[1])
    assert np.array_equal(a[1][finite], b[1][finite])
    a, b = run(handle, noisy, True), run(handle, noisy, False)  # ... and the next pass is back on the screen
    np.testing.assert_allclose(a[0], b[0], rtol=2e-6)
    assert rel_linf(a[1], b[1]) < 5e-5


def test_marginalized_prior_fit_matches_oracle():
    """A short sequential fit with GMMPatchPrior(marginalize=True) against oracle/cpu_ref.py."""
    from jolideco_amd import GMMPatchPrior, MAPDeconvolver, SpatialFluxComponent
    from jolideco_amd.data import point_source_gauss_psf, synthetic_gmm
    from jolideco_amd.priors.patches import GaussianMixtureModel, GaussianMixtureModelMeta
    from oracle import cpu_ref

    rs = np.random.RandomState(5)
    datasets = {f"o{i}": point_source_gauss_psf(shape=(40, 48), sigma_psf=2 + i, random_state=rs) for i in range(2)}
    for d in datasets.values():
        d.pop("flux")
    flux_init = rs.gamma(30, size=(40, 48))
    means, covs, weights = synthetic_gmm(6, 64, seed=3)
    gmm = GaussianMixtureModel.from_numpy(means, covs, weights, meta=GaussianMixtureModelMeta(stride=4))
    comp = SpatialFluxComponent.from_numpy(flux=flux_init, prior=GMMPatchPrior(gmm=gmm, marginalize=True))
    res = MAPDeconvolver(n_epochs=4, display_progress=False, device=DEV).run(datasets, components=comp)
    gmm_o = cpu_ref.GMM.from_numpy(means, covs, weights, stride=4)
    final, trace = cpu_ref.map_fit_sequential(
        datasets, {"flux": flux_init}, {"flux": cpu_ref.GMMPatchPriorRef(gmm_o, marginalize=True)}, n_epochs=4
    )
    with cpu_ref.precision(np.float64):
        gmm_64 = cpu_ref.GMM.from_numpy(means, covs, weights, stride=4)
        final_64, _ = cpu_ref.map_fit_sequential(
            datasets, {"flux": flux_init}, {"flux": cpu_ref.GMMPatchPriorRef(gmm_64, marginalize=True)}, n_epochs=4
        )
    err, d_gpu, d_o = rel_linf(res.flux_total, final["flux"]), rel_linf(res.flux_total, final_64["flux"]), rel_linf(final["flux"], final_64["flux"])
    print(f"lse fit: |gpu-oracle32| {err:.2e} |gpu-f64| {d_gpu:.2e} |oracle32-f64| {d_o:.2e}")
    assert err < 5e-5  # see the tolerance note above
    assert d_gpu <= 2.0 * d_o + 2e-6  # as close to the float64 fit as the fp32 oracle
    np.testing.assert_allclose(res.trace_loss[-1]["total"], trace[-1]["total"], rtol=2e-5)


@pytest.mark.parametrize(
    "shape,K,seed,mean_scale",
    [((96, 128), 8, 0, 0.0), ((257, 131), 37, 1, 0.0), ((512, 512), 128, 2, 0.0), ((64, 64), 1, 3, 0.0),
     ((200, 168), 64, 4, 0.02), ((120, 136), 16, 5, 1.0)],
)
def test_gmm_screened_argmax_is_bit_identical_to_dense(shape, K, seed, mean_scale, jd_option):
    """Max mode through the bf16 screen + exact fp32 re-evaluation of the survivors (csrc/gmm.hip, gmm_screen_kernel)
    returns exactly the dense fp32 kernel's numbers: same arg-max for every patch, same value bits, same gradient
    bits -- on noise, on smooth structure with bright points, with filtered patches and with cycle-spin shifts; also
    for mixtures with small (trained-GMM like) and large component means, which only widen the screen's bounds."""
    from jolideco_amd.data import synthetic_gmm
    from jolideco_amd.priors.patches import GaussianMixtureModel, GaussianMixtureModelMeta

    rs = np.random.RandomState(seed)
    means, covs, weights = synthetic_gmm(K, 64, seed=seed)
    means = means + mean_scale * rs.normal(size=means.shape)
    gmm = GaussianMixtureModel.from_numpy(means, covs, weights, meta=GaussianMixtureModelMeta(stride=4))
    y, x = np.mgrid[0 : shape[0], 0 : shape[1]]
    smooth = 1.0 + 20 * np.exp(-0.5 * (((y - shape[0] / 3) / 9.0) ** 2 + ((x - shape[1] / 2) / 14.0) ** 2))
    images = {
        "noise": rs.gamma(30, size=shape),
        "smooth+points": smooth * rs.gamma(50, size=shape) / 50,
        "flat": np.full(shape, 3.0),
    }
    images["smooth+points"][rs.randint(0, shape[0], 5), rs.randint(0, shape[1], 5)] += 500.0
    images["noise"][5:9, 7:11] = -2e5  # filtered patches (patches/core.py:215)
    handle = gmm.handle(DEV)
    n_patches = ((shape[0] - 8) // 4 + 1) * ((shape[1] - 8) // 4 + 1)
    for name, image in images.items():
        flux = torch.from_numpy(image.astype(np.float32)).to(DEV)
        for shifts in [(0, 0), (-2, 1)]:
            out = {}
            for mode in ("1", "0"):
                jd_option("JD_GMM_SCREEN", mode)
                value, grad = torch.zeros(1, device=DEV), torch.zeros_like(flux)
                argmax = torch.full((n_patches,), -7, dtype=torch.int32, device=DEV)
                handle.prior_fwd_bwd(flux, 4, shifts, value, 0.25, grad=grad, grad_coef=0.5, argmax_out=argmax)
                torch.cuda.synchronize()
                out[mode] = (float(value), grad.cpu().numpy(), argmax.cpu().numpy())
            assert np.array_equal(out["1"][2], out["0"][2]), (name, shifts)
            assert out["1"][0] == out["0"][0] or abs(out["1"][0] - out["0"][0]) <= 2e-7 * abs(out["0"][0]), (name, shifts)
            assert np.array_equal(out["1"][1], out["0"][1]), (name, shifts)
    # a patch-row shard (multi-GPU partition) goes through the same path
    flux = torch.from_numpy(images["noise"].astype(np.float32)).to(DEV)
    rows = (shape[0] - 8) // 4 + 1
    vals = {}
    for mode in ("1", "0"):
        jd_option("JD_GMM_SCREEN", mode)
        parts = []
        for r0, r1 in ((0, rows // 3), (rows // 3, rows)):
            v = torch.zeros(1, device=DEV)
            handle.prior_fwd_bwd(flux, 4, (1, -1), v, 1.0, patch_rows=(r0, r1))
            parts.append(float(v))
        vals[mode] = parts
    assert vals["1"] == pytest.approx(vals["0"], rel=2e-7)


@pytest.mark.parametrize("method,shape", [("separable", (97, 150)), ("direct", (97, 150)), ("direct", (200, 192))])
def test_poisson_epilogue_of_the_convolution_matches_the_two_kernel_path(monkeypatch, method, shape, jd_option):
    """One component without up-sampling: the Poisson pass runs as the epilogue of the forward convolution
    (sep_conv_kernel<.., POISSON> / direct_conv_kernel<.., POISSON>).  Same loss, predicted counts and gradient as
    convolution + poisson_fused_kernel (JD_SEP_NO_FUSION=1), on a ragged image (scalar epilogue) and on one with
    W % 4 == 0 (float4 epilogue, partial last tile row), with / without the gradient."""
    from jolideco_amd import FluxComponents, NPredModels, SpatialFluxComponent
    from jolideco_amd.ops import stirling_mean

    monkeypatch.setenv("JOLIDECO_CONV_METHOD", method)
    rs = np.random.RandomState(5)
    g = np.exp(-0.5 * ((np.arange(17) - 8) / 2.0) ** 2)
    psf = np.outer(g, g)
    if method == "direct":  # not separable, with negative lobes: conv < 0 somewhere, the clamp's backward pass matters
        psf = psf * (1.0 + 0.3 * rs.uniform(-1, 1, size=psf.shape)) - 0.02
    data = {
        "counts": rs.poisson(3.0, size=shape).astype(np.float32),
        "psf": (psf / psf.sum()).astype(np.float32),
        "exposure": rs.uniform(0.5, 1.5, size=shape).astype(np.float32),
        "background": rs.uniform(0.2, 1.0, size=shape).astype(np.float32),
    }
    flux_np = rs.gamma(3.0, size=shape).astype(np.float32)
    if method == "direct":  # sparse sources: the negative lobes win between them
        flux_np *= rs.uniform(size=shape) < 0.03
    flux = torch.from_numpy(flux_np).to(DEV)
    comps = FluxComponents()
    comps["flux"] = SpatialFluxComponent.from_numpy(flux=flux.cpu().numpy())
    models = NPredModels.from_dataset_numpy(dataset=data, components=comps, device=DEV)
    assert models.plan.method == method
    counts = torch.from_numpy(data["counts"]).to(DEV)
    results = {}
    for fusion in ("fused", "split"):
        if fusion == "split":
            jd_option("JD_SEP_NO_FUSION", "1")
        loss, grad, npred = torch.zeros(1, device=DEV), torch.full_like(flux, 3.0), torch.empty_like(flux)
        models.fwd_bwd([flux], counts, stirling_mean(data["counts"]), loss, grads=[grad], npred_out=npred, accumulate=True,
                       grad_scale=0.5)
        loss_fwd = torch.zeros(1, device=DEV)
        models.fwd_bwd([flux], counts, stirling_mean(data["counts"]), loss_fwd)
        torch.cuda.synchronize()
        results[fusion] = (float(loss), float(loss_fwd), grad.cpu().numpy(), npred.cpu().numpy())
    a, b = results["fused"], results["split"]
    assert a[0] == pytest.approx(b[0], rel=1e-6) and a[1] == pytest.approx(b[1], rel=1e-6) and a[0] == pytest.approx(a[1], rel=1e-6)
    assert np.array_equal(a[3], b[3])       # predicted counts: same arithmetic, same bits
    if method == "direct":
        clipped = a[3] == data["background"]
        assert 0.05 < clipped.mean() < 0.95  # the clamp is active on part of the image
    assert rel_linf(a[2] - 3.0, b[2] - 3.0) < 1e-6  # gradient: same g, same adjoint kernel


def _prior_both_ways(handle, flux, n_patches, jd_option, shifts=(1, -2)):
    out = {}
    for mode in ("1", "0"):
        jd_option("JD_GMM_SCREEN", mode)
        value, grad = torch.zeros(1, device=DEV), torch.zeros_like(flux)
        argmax = torch.full((n_patches,), -7, dtype=torch.int32, device=DEV)
        handle.prior_fwd_bwd(flux, 4, shifts, value, 1.0, grad=grad, grad_coef=1.0, argmax_out=argmax)
        torch.cuda.synchronize()
        out[mode] = (float(value), grad.cpu().numpy(), argmax.cpu().numpy())
    return out["1"], out["0"]


def test_gmm_record_buffer_grows_after_a_pass_that_ran_out_of_it():
    """16 distinct components, each present 8 times: every patch has at least 8 exactly tied candidates, twice what the
    record-gradient buffer has room for at first (4 rows per patch).  The first passes fall back to the dense kernel
    (same bits, four times the work); the statistics the last block of a pass leaves in host-mapped memory make the
    following passes double the room until the screened path holds: no synchronisation, no wrong number in between."""
    from jolideco_amd.data import synthetic_gmm
    from jolideco_amd.priors.patches import GaussianMixtureModel, GaussianMixtureModelMeta

    shape = (160, 192)
    n_patches = ((shape[0] - 8) // 4 + 1) * ((shape[1] - 8) // 4 + 1)
    means, covs, weights = synthetic_gmm(16, 64, seed=5)
    gmm = GaussianMixtureModel.from_numpy(np.tile(means, (8, 1)), np.tile(covs, (8, 1, 1)), np.tile(weights, 8) / 8,
                                          meta=GaussianMixtureModelMeta(stride=4))
    handle = gmm.handle(DEV)
    flux = torch.from_numpy(np.random.RandomState(1).gamma(20, size=shape).astype(np.float32)).to(DEV)
    results, stats = [], []
    for _ in range(6):
        value, grad = torch.zeros(1, device=DEV), torch.zeros_like(flux)
        argmax = torch.full((n_patches,), -7, dtype=torch.int32, device=DEV)
        handle.prior_fwd_bwd(flux, 4, (2, -1), value, 1.0, grad=grad, grad_coef=1.0, argmax_out=argmax)
        torch.cuda.synchronize()  # (the test wants to see each pass's statistics; the library never waits for them)
        results.append((float(value), grad.cpu().numpy(), argmax.cpu().numpy()))
        stats.append(handle.screen_stats())
    print("screen statistics per pass (generation, fell back, slots, patches, rows per patch):", stats)
    assert stats[0][1] == 1 and stats[0][4] == 4           # ran out of room: the dense kernel took the first pass
    assert stats[-1][1] == 0 and stats[-1][4] >= 16         # room doubled until the screened path holds
    assert all(s[3] == n_patches for s in stats)
    for value, grad, argmax in results[1:]:                 # the same bits whoever computed them
        assert value == results[0][0] and np.array_equal(grad, results[0][1]) and np.array_equal(argmax, results[0][2])
    assert np.all(results[0][2] < 16)                       # ties go to the lowest component, like torch.max


def test_gmm_screen_falls_back_to_the_dense_kernel(jd_option):
    """The two situations in which the screen gives up (device flag -> the always-enqueued dense kernel overwrites
    the result, no host sync): more candidates than a wave's record list holds, and non-finite screening values."""
    from jolideco_amd.data import synthetic_gmm
    from jolideco_amd.priors.patches import GaussianMixtureModel, GaussianMixtureModelMeta

    shape = (160, 192)
    n_patches = ((shape[0] - 8) // 4 + 1) * ((shape[1] - 8) // 4 + 1)
    rs = np.random.RandomState(0)
    # (a) 128 IDENTICAL components: every component is a candidate for every patch (128 > 32 records per patch)
    means, covs, weights = synthetic_gmm(1, 64, seed=3)
    K = 128
    gmm = GaussianMixtureModel.from_numpy(
        np.repeat(means, K, axis=0), np.repeat(covs, K, axis=0), np.full(K, 1.0 / K), meta=GaussianMixtureModelMeta(stride=4)
    )
    flux = torch.from_numpy(rs.gamma(20, size=shape).astype(np.float32)).to(DEV)
    screened, dense = _prior_both_ways(gmm.handle(DEV), flux, n_patches, jd_option)
    assert np.array_equal(screened[2], dense[2]) and np.all(dense[2] == 0)  # ties -> the lowest component, like torch.max
    assert screened[0] == pytest.approx(dense[0], rel=2e-7) and np.array_equal(screened[1], dense[1])
    # (b) an infinite pixel: the patches that contain it have no finite log-likelihood
    means, covs, weights = synthetic_gmm(16, 64, seed=4)
    gmm = GaussianMixtureModel.from_numpy(means, covs, weights, meta=GaussianMixtureModelMeta(stride=4))
    image = rs.gamma(20, size=shape).astype(np.float32)
    image[40, 50] = np.inf
    flux = torch.from_numpy(image).to(DEV)
    screened, dense = _prior_both_ways(gmm.handle(DEV), flux, n_patches, jd_option)
    assert np.array_equal(screened[2], dense[2])
    assert np.array_equal(np.isnan(screened[1]), np.isnan(dense[1]))
    finite = np.isfinite(dense[1])
    assert np.array_equal(screened[1][finite], dense[1][finite])
    assert (np.isnan(screened[0]) and np.isnan(dense[0])) or screened[0] == dense[0]


def test_gmm_fused_backward_equals_the_bucketed_backward(jd_option):
    """Screened arg-max with a gradient: by default the exact kernel also writes the gradient row of every surviving
    record and the gather kernel reads the winner's row (no second sort, no backward kernel); JD_GMM_FUSED_BWD=0 keeps
    the bucketed backward pass.  Same bits -- whole image, a patch-row shard, with and without the arg-max asked for,
    and when the record-gradient buffer would overflow (4 identical components: 4 records per patch > 2) or a pixel is
    infinite (both fall back on the device: dense forward kernel + unsorted backward kernel)."""
    from jolideco_amd.data import synthetic_gmm
    from jolideco_amd.priors.patches import GaussianMixtureModel, GaussianMixtureModelMeta

    shape = (136, 172)
    n_py, n_px = (shape[0] - 8) // 4 + 1, (shape[1] - 8) // 4 + 1
    rs = np.random.RandomState(5)

    def run(handle, flux, fused, rows=(0, -1), want_argmax=True):
        jd_option("JD_GMM_FUSED_BWD", "1" if fused else "0")
        value, grad = torch.zeros(1, device=DEV), torch.zeros_like(flux)
        argmax = torch.full((n_py * n_px,), -7, dtype=torch.int32, device=DEV) if want_argmax else None
        handle.prior_fwd_bwd(flux, 4, (3, -5), value, 0.25, grad=grad, grad_coef=-0.7, patch_rows=rows, argmax_out=argmax)
        torch.cuda.synchronize()
        return float(value), grad.cpu().numpy(), None if argmax is None else argmax.cpu().numpy()

    means, covs, weights = synthetic_gmm(24, 64, seed=9)
    gmm = GaussianMixtureModel.from_numpy(means, covs, weights, meta=GaussianMixtureModelMeta(stride=4))
    image = rs.gamma(20, size=shape).astype(np.float32)
    image[60:64, 80:90] = -2e5  # filtered patches (patches/core.py:215-216): no gradient
    flux = torch.from_numpy(image).to(DEV)
    handle = gmm.handle(DEV)
    for rows in ((0, -1), (5, 19)):
        a, b = run(handle, flux, True, rows), run(handle, flux, False, rows)
        assert a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
        assert np.abs(a[1]).max() > 0 and (a[2] == -1).sum() > 0
        c = run(handle, flux, True, rows, want_argmax=False)
        assert c[0] == a[0] and np.array_equal(c[1], a[1])

    # fallbacks: the record-gradient buffer holds 2 rows per patch; an infinite pixel
    means, covs, weights = synthetic_gmm(1, 64, seed=3)
    gmm4 = GaussianMixtureModel.from_numpy(
        np.repeat(means, 4, axis=0), np.repeat(covs, 4, axis=0), np.full(4, 0.25), meta=GaussianMixtureModelMeta(stride=4)
    )
    a, b = run(gmm4.handle(DEV), flux, True), run(gmm4.handle(DEV), flux, False)
    assert a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    assert set(np.unique(a[2])) == {-1, 0}  # ties -> the lowest component
    image[20, 30] = np.inf
    flux = torch.from_numpy(image).to(DEV)
    a, b = run(handle, flux, True), run(handle, flux, False)
    assert np.array_equal(a[2], b[2]) and np.array_equal(np.isnan(a[1]), np.isnan(b[1]))
    finite = np.isfinite(b[1])
    assert np.array_equal(a[1][finite], b[1][finite])
    # ... and the next pass on the same handle is back on the fast path (generation-stamped flag, nothing to clear)
    image[20, 30] = 1.0
    flux = torch.from_numpy(image).to(DEV)
    a, b = run(handle, flux, True), run(handle, flux, False)
    assert a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and np.isfinite(a[0])


def test_gmm_screen_large_k_and_huge_dynamic_range(jd_option):
    """K above the popularity-order limit (natural order is kept) and fluxes / precisions far outside the fp16 range
    (the power-of-two operand scales keep the screen exact): still the dense kernel's bits."""
    from jolideco_amd.data import synthetic_gmm
    from jolideco_amd.priors.patches import GaussianMixtureModel, GaussianMixtureModelMeta

    shape = (72, 88)
    n_patches = ((shape[0] - 8) // 4 + 1) * ((shape[1] - 8) // 4 + 1)
    rs = np.random.RandomState(1)
    means, covs, weights = synthetic_gmm(1100, 64, seed=6)
    covs = covs * np.logspace(-9, 7, covs.shape[0])[:, None, None]  # precisions from 1e-4 to 3e4: beyond fp16 without scaling
    gmm = GaussianMixtureModel.from_numpy(means, covs, weights, meta=GaussianMixtureModelMeta(stride=4))
    handle = gmm.handle(DEV)
    for scale in (1e-12, 1.0, 3e7):
        flux = torch.from_numpy((scale * rs.gamma(20, size=shape)).astype(np.float32)).to(DEV)
        for _ in range(2):  # the second call uses the component order learnt in the first
            screened, dense = _prior_both_ways(handle, flux, n_patches, jd_option)
            assert np.array_equal(screened[2], dense[2]), scale
            assert screened[0] == pytest.approx(dense[0], rel=2e-7), scale
            assert np.array_equal(screened[1], dense[1]), scale


@pytest.mark.parametrize("stride", [4, 5, 8])
@pytest.mark.parametrize("marginalize", [False, True])
def test_gmm_tiled_gather_equals_the_per_pixel_gather(jd_option, stride, marginalize):
    """The overlap-add of the patch gradients runs tile-wise through LDS for strides >= 4 (`gmm_gather_tile_kernel`);
    JD_GMM_GATHER_TILED=0 selects the per-pixel kernel.  Same additions in the same order: same bits -- whole image,
    patch-row shards (tile boundaries inside the shard), filtered patches, image sizes that are no multiple of the
    tile, all three sources of rows (record rows of the fused arg-max backward pass, bucketed backward, logsumexp)."""
    from jolideco_amd.data import synthetic_gmm
    from jolideco_amd.priors.patches import GaussianMixtureModel, GaussianMixtureModelMeta

    shape = (150, 203)
    rs = np.random.RandomState(stride)
    means, covs, weights = synthetic_gmm(12, 64, seed=4)
    gmm = GaussianMixtureModel.from_numpy(means, covs, weights, meta=GaussianMixtureModelMeta(stride=4))
    image = rs.gamma(20, size=shape).astype(np.float32)
    image[40:44, 100:110] = -2e5  # filtered patches: no gradient
    flux = torch.from_numpy(image).to(DEV)
    handle = gmm.handle(DEV)
    n_rows = (shape[0] - 8) // stride + 1

    def run(tiled, rows, fused=True):
        jd_option("JD_GMM_GATHER_TILED", "1" if tiled else "0")
        jd_option("JD_GMM_FUSED_BWD", "1" if fused else "0")
        value, grad = torch.zeros(1, device=DEV), torch.full_like(flux, 0.5)  # accumulates into a non-zero image
        handle.prior_fwd_bwd(flux, stride, (3, -5), value, 0.25, grad=grad, grad_coef=-0.7, patch_rows=rows,
                             marginalize=marginalize)
        torch.cuda.synchronize()
        return grad.cpu().numpy()

    for rows in ((0, -1), (3, n_rows - 2), (7, 8)):
        for fused in ((True, False) if not marginalize else (True,)):
            a, b = run(True, rows, fused), run(False, rows, fused)
            assert np.array_equal(a, b), (rows, fused)
            assert np.abs(a - 0.5).max() > 0


@pytest.mark.parametrize("marginalize", [False, True])
def test_prior_bands_of_a_sharded_prior_add_up_to_the_whole(marginalize):
    """jd_gmm_prior_band_fwd_bwd + jd_add_rolled_bands (what the ranks of a sharded joint fit exchange with one
    all-gather): the bands of three shards of patch rows -- one of them empty -- laid out like an all-gather buffer and
    added back give the gradient of the accumulating calls (gradient image zero before), shard values sum to the whole
    value."""
    from jolideco_amd.data import synthetic_gmm
    from jolideco_amd.ops import add_rolled_bands, band_rows
    from jolideco_amd.priors.patches import GaussianMixtureModel, GaussianMixtureModelMeta

    shape, stride, shifts = (118, 164), 4, (-2, 3)
    rs = np.random.RandomState(2)
    means, covs, weights = synthetic_gmm(10, 64, seed=8)
    handle = GaussianMixtureModel.from_numpy(means, covs, weights, meta=GaussianMixtureModelMeta(stride=4)).handle(DEV)
    flux = torch.from_numpy(rs.gamma(20, size=shape).astype(np.float32)).to(DEV)
    n_rows = (shape[0] - 8) // stride + 1
    shards = [(0, 9), (9, 9), (9, n_rows)]
    y_ranges = [band_rows(r, stride, shape[0]) for r in shards]
    assert y_ranges[1][0] == y_ranges[1][1] and y_ranges[0][1] > y_ranges[2][0]  # empty shard; the bands overlap
    chunk = max(y1 - y0 for y0, y1 in y_ranges) * shape[1] + 4
    pieces = torch.full((len(shards) * chunk,), 7.0, device=DEV)  # garbage behind the bands must be ignored
    ref_v, ref_g = torch.zeros(1, device=DEV), torch.zeros_like(flux)
    values = []
    for i, rows in enumerate(shards):
        handle.prior_fwd_bwd(flux, stride, shifts, ref_v, 0.25, grad=ref_g, grad_coef=-0.7, patch_rows=rows,
                             accumulate_value=True, marginalize=marginalize)
        v = torch.zeros(1, device=DEV)
        handle.prior_fwd_bwd(flux, stride, shifts, v, 0.25, grad_coef=-0.7, patch_rows=rows, marginalize=marginalize,
                             band_out=pieces[i * chunk : (i + 1) * chunk])
        values.append(float(v))
    grad = torch.zeros_like(flux)
    add_rolled_bands(grad, shifts, pieces, chunk, y_ranges)
    torch.cuda.synchronize()
    # bit for bit outside the rows two bands share; there grad + coef * sum is one fused multiply-add in the accumulating
    # calls and a rounded product plus an addition through the bands: equal to an ulp
    got, ref = grad.cpu().numpy(), ref_g.cpu().numpy()
    shared = np.zeros(shape[0], dtype=bool)
    shared[(np.arange(y_ranges[2][0], y_ranges[0][1]) - shifts[0]) % shape[0]] = True
    assert np.array_equal(got[~shared], ref[~shared]) and np.abs(ref).max() > 0
    assert rel_linf(got, ref) < 1e-6
    np.testing.assert_allclose(sum(values), float(ref_v), rtol=1e-6)
    assert values[1] == 0.0
    # into a non-zero gradient image: (g + a) + b vs g + (a + b) on the overlap rows -- equal to rounding
    g1, g2 = torch.full_like(flux, 0.3), torch.full_like(flux, 0.3)
    for rows in shards:
        handle.prior_fwd_bwd(flux, stride, shifts, ref_v, 0.25, grad=g1, grad_coef=-0.7, patch_rows=rows, marginalize=marginalize)
    add_rolled_bands(g2, shifts, pieces, chunk, y_ranges)
    assert rel_linf(g2.cpu().numpy(), g1.cpu().numpy()) < 1e-6


def _separable_step_outputs(shape, psf, seed=3):
    """loss, gradient, predicted counts, gradient of the call without npred, plain convolution and adjoint of one
    dataset on the separable plan, as numpy arrays"""
    from jolideco_amd import FluxComponents, NPredModels, SpatialFluxComponent
    from jolideco_amd.ops import stirling_mean

    rs = np.random.RandomState(seed)
    data = {"counts": rs.poisson(3.0, size=shape).astype(np.float32), "psf": psf.astype(np.float32),
            "exposure": rs.uniform(0.5, 1.5, size=shape).astype(np.float32),
            "background": rs.uniform(0.2, 1.0, size=shape).astype(np.float32)}
    flux = torch.from_numpy(rs.gamma(3.0, size=shape).astype(np.float32)).to(DEV)
    comps = FluxComponents()
    comps["flux"] = SpatialFluxComponent.from_numpy(flux=flux.cpu().numpy())
    models = NPredModels.from_dataset_numpy(dataset=data, components=comps, device=DEV)
    assert models.plan.method == "separable"
    counts = torch.from_numpy(data["counts"]).to(DEV)
    loss, grad, npred = torch.zeros(1, device=DEV), torch.zeros_like(flux), torch.empty_like(flux)
    models.fwd_bwd([flux], counts, stirling_mean(data["counts"]), loss, grads=[grad], npred_out=npred)
    grad2 = torch.zeros_like(flux)
    models.fwd_bwd([flux], counts, stirling_mean(data["counts"]), loss, grads=[grad2])  # fused epilogue, no npred
    conv = models.plan.conv_same(flux, models["flux"].exposure[0, 0], models["flux"].khat)
    adj = models.plan.conv_same_adjoint(grad, models["flux"].exposure[0, 0], models["flux"].khat)
    torch.cuda.synchronize()
    # (the model's exposure is the edge-corrected one, models/npred.py:108-113)
    return [t.cpu().numpy() for t in (loss, grad, npred, grad2, conv, adj)], (
        data, flux.cpu().numpy(), models["flux"].exposure[0, 0].cpu().numpy())


def test_separable_convolution_lds_aliasing_changes_no_bit(jd_option):
    """For a rank-1 operator the row-pass image of `sep_conv_kernel` shares the LDS of the input window (a sixth block
    per CU for the fused forward + Poisson launch); option JD_SEP_NO_ALIAS keeps them apart.  Same arithmetic, same
    bits: loss, gradient, predicted counts, plain convolution, adjoint.  (The rank is the one the library registered for
    the operator buffer when it built it, checked again on the device.)"""
    from jolideco_amd.data import gaussian_kernel

    jd_option("JD_SEP_WALK", 0)  # the tile kernel is the one under test
    out = {}
    for mode in ("alias", "plain"):
        jd_option("JD_SEP_NO_ALIAS", 1 if mode == "plain" else None)
        out[mode], _ = _separable_step_outputs((200, 328), gaussian_kernel(2.0, (17, 17)))
    for a, b in zip(out["alias"], out["plain"]):
        assert np.array_equal(a, b)
    assert np.abs(out["alias"][1]).max() > 0


@pytest.mark.parametrize("shape,kshape", [((200, 328), (17, 17)), ((75, 260), (9, 13)), ((130, 516), (16, 17)),
                                          ((64, 132), (5, 7))], ids=["17x17", "9x13", "16x17", "5x7"])
def test_strip_walk_kernels_match_the_tile_kernel_and_float64(jd_option, shape, kshape):
    """csrc/walkconv.hip (strip-walk form of the separable convolution: forward model + Poisson pass, plain
    convolution, adjoint) against csrc/sepconv.hip's tile kernel on the same inputs, and the plain convolution against
    float64 (scipy): different summation orders of the same 'same' convolution (utils/torch.py:347-370)."""
    from scipy.signal import fftconvolve

    from jolideco_amd.data import gaussian_kernel

    psf = gaussian_kernel(1.7, kshape)
    out = {}
    for walk in (0, 1):
        jd_option("JD_SEP_WALK", walk)
        out[walk], (data, flux, exposure) = _separable_step_outputs(shape, psf)
    names = ("loss", "gradient", "npred", "gradient (no npred)", "convolution", "adjoint")
    for name, a, b in zip(names, out[1], out[0]):
        assert rel_linf(a, b) < 2e-6, name
    assert np.array_equal(out[1][1], out[1][3])
    ref = fftconvolve(flux.astype(np.float64) * exposure, data["psf"].astype(np.float64), mode="full")
    oy, ox = (kshape[0] - 1) // 2, (kshape[1] - 1) // 2
    assert rel_linf(out[1][4], ref[oy:oy + shape[0], ox:ox + shape[1]]) < 1e-6


def test_an_operator_buffer_overwritten_behind_the_library_is_reported(jd_option):
    """The kernels that assume a rank-1 operator (walk kernels, LDS aliasing) are launched on the rank the library
    registered when it built the buffer; they re-check op[0] on the device and the next call fails loudly if the
    buffer was overwritten with an operator of another rank (never a silently wrong convolution)."""
    from jolideco_amd.data import gaussian_kernel
    from jolideco_amd.ops import ConvPlan

    jd_option("JD_SEP_WALK", 1)
    H, W = 64, 128
    plan = ConvPlan(H, W, 17, 17, DEV, method="separable")
    g1 = torch.from_numpy(gaussian_kernel(2.0, (17, 17)).astype(np.float32)).to(DEV)
    g2 = torch.from_numpy((0.6 * gaussian_kernel(1.5, (17, 17)) + 0.4 * gaussian_kernel(4.0, (17, 17))).astype(np.float32)).to(DEV)
    k1, k2 = plan.psf_spectrum(g1), plan.psf_spectrum(g2)  # rank 1 and rank 2
    image = torch.rand(H, W, device=DEV)
    ok = plan.conv_same(image, None, k1)
    k1.copy_(k2)  # the caller overwrites the rank-1 buffer with a rank-2 operator
    plan.conv_same(image, None, k1)
    torch.cuda.synchronize()
    with pytest.raises(RuntimeError, match="rank"):
        plan.conv_same(image, None, k1)
    # a copy the library has never seen takes the general path and is right
    k3 = k2.clone()
    got = plan.conv_same(image, None, k3).cpu().numpy()
    ref = plan.conv_same(image, None, k2).cpu().numpy()
    assert np.array_equal(got, ref) and np.isfinite(ok.cpu().numpy()).all()
    plan.close()


@pytest.mark.parametrize("shape,kshape,n_obs", [((100, 260), (17, 17), 8), ((130, 300), (17, 8), 3), ((64, 128), (9, 13), 2),
                                                ((96, 256), (5, 5), 1), ((200, 512), (16, 17), 11)],
                         ids=["8obs", "3obs", "2obs", "1obs", "11obs"])
def test_fused_likelihood_step_equals_the_two_launch_path_bit_for_bit(jd_option, shape, kshape, n_obs):
    """Option JD_SEP_JOINT = 1: forward model, Poisson pass, adjoint convolution and the sum over the datasets in ONE
    launch (walk_joint_kernel: the g images are never written).  Same arithmetic in the same order as the strip-walk
    forward + adjoint launches: the gradient of the joint step is the same bit for bit (batched and per dataset), the
    losses are summed over other tiles and agree to rounding."""
    from jolideco_amd.data import gaussian_kernel
    from jolideco_amd.ops import ConvPlan, stirling_mean

    H, W = shape
    rs = np.random.RandomState(H + W + n_obs)
    flux = torch.from_numpy(rs.gamma(5.0, size=shape).astype(np.float32)).to(DEV)
    plan = ConvPlan(H, W, kshape[0], kshape[1], DEV, method="separable")
    data = []
    for i in range(n_obs):
        psf = torch.from_numpy(gaussian_kernel(1.2 + 0.3 * i, kshape).astype(np.float32)).to(DEV)
        exposure = (1.0 + 0.1 * i) * (1.0 + 0.4 * np.linspace(-1, 1, H)[:, None] * np.ones(shape))
        counts = rs.poisson(5.0, size=shape).astype(np.float32)
        data.append((plan.psf_spectrum(psf), torch.from_numpy(exposure.astype(np.float32)).to(DEV),
                     torch.full(shape, 0.5 + 0.1 * i, device=DEV), torch.from_numpy(counts).to(DEV), stirling_mean(counts)))

    def step(batch):
        losses = [torch.zeros(1, device=DEV) for _ in range(n_obs)]
        grad = torch.full(shape, 0.25, device=DEV)  # accumulate into a non-zero image
        if batch:
            plan.npred_poisson_batch_fwd_bwd(flux, [d[1] for d in data], [d[0] for d in data], [d[2] for d in data],
                                             [d[3] for d in data], [d[4] for d in data], losses, grad=grad, accumulate=True)
        else:
            for i, d in enumerate(data):
                plan.npred_poisson_fwd_bwd([flux], [d[1]], [d[0]], d[2], d[3], d[4], losses[i], grads=[grad], accumulate=True)
        torch.cuda.synchronize()
        return grad.cpu().numpy(), np.array([float(v) for v in losses])

    jd_option("JD_SEP_WALK", 1)
    jd_option("JD_SEP_JOINT", 0)
    ref_grad, ref_loss = step(batch=True)
    jd_option("JD_SEP_JOINT", 1)
    for batch in (True, False):
        grad, loss = step(batch)
        assert np.array_equal(grad, ref_grad), f"batch={batch}"
        np.testing.assert_allclose(loss, ref_loss, rtol=1e-6)
    assert np.abs(ref_grad - 0.25).max() > 0
    plan.close()


def test_adam_step_multi_equals_the_single_tensor_steps_bit_for_bit():
    """jd_adam_step_multi (many small parameter vectors, one launch, each with its own step count) == one jd_adam_step
    with use_log_flux = 0 per tensor: parameters and both moments, bit for bit, over several steps."""
    import ctypes

    from jolideco_amd import _hip
    from jolideco_amd._hip import check, ptr, ptr_array, stream_ptr
    from jolideco_amd.ops import adam_bias_terms

    rs = np.random.RandomState(4)
    sizes = [2, 1, 2, 1, 7, 64, 130]
    lr, beta1, beta2, eps = 0.1, 0.9, 0.999, 1e-8

    def fresh():
        rs2 = np.random.RandomState(5)
        return [[torch.from_numpy(rs2.normal(size=n).astype(np.float32)).to(DEV) for n in sizes],
                [torch.zeros(n, device=DEV) for n in sizes], [torch.zeros(n, device=DEV) for n in sizes]]

    (ta, ma, va), (tb, mb, vb) = fresh(), fresh()
    steps = [0] * len(sizes)
    for it in range(4):
        grads = [torch.from_numpy(rs.normal(size=n).astype(np.float32)).to(DEV) for n in sizes]
        active = [i for i in range(len(sizes)) if not (it == 1 and i == 2)]  # tensor 2 sits one step out: own step count
        terms = {}
        for i in active:
            steps[i] += 1
            terms[i] = adam_bias_terms(steps[i], lr, beta1, beta2)
        for i in active:
            check(_hip.lib().jd_adam_step(ptr(ta[i]), ptr(ta[i]), ptr(ta[i]), ptr(grads[i]), ptr(ma[i]), ptr(va[i]), None, sizes[i],
                                          terms[i][0], beta1, beta2, 1 - beta1, 1 - beta2, terms[i][1], eps, 0, 0, None, stream_ptr(ta[i].device)))
        n = len(active)
        check(_hip.lib().jd_adam_step_multi(
            n, ptr_array([tb[i] for i in active]), ptr_array([grads[i] for i in active]), ptr_array([mb[i] for i in active]),
            ptr_array([vb[i] for i in active]), (ctypes.c_int * n)(*[sizes[i] for i in active]),
            (ctypes.c_float * n)(*[terms[i][0] for i in active]), (ctypes.c_float * n)(*[terms[i][1] for i in active]),
            beta1, beta2, 1 - beta1, 1 - beta2, eps, None, stream_ptr(tb[0].device)))
    torch.cuda.synchronize()
    for i in range(len(sizes)):
        np.testing.assert_array_equal(ta[i].cpu().numpy(), tb[i].cpu().numpy())
        np.testing.assert_array_equal(ma[i].cpu().numpy(), mb[i].cpu().numpy())
        np.testing.assert_array_equal(va[i].cpu().numpy(), vb[i].cpu().numpy())
