"""Datasets whose PSF arrays differ in size (config 3 has 17x17 PSFs and, for sigma >= 3, 33x33
ones) in ONE batched joint step, and the 33-tap frame of the strip-walk kernels (csrc/walkconv.hip).

The reference convolves any kernel size through one call (jolideco/utils/torch.py:347-370); here PSFs of different
sizes are embedded in zeros up to a common array shape (same 'same' convolution), share one separable plan, and the
kernels work on the NON-ZERO taps of each operator: a 17x17 PSF in a 33x33 array walks in the 17-tap frame.
"""
import numpy as np
import pytest
import torch

from conftest import rel_linf

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _embed(psf, shape):
    from jolideco_amd.models.npred import embed_kernel

    return embed_kernel(np.asarray(psf, dtype=np.float32), shape)


@pytest.mark.parametrize("shape,kshape", [((200, 328), (33, 33)), ((75, 260), (25, 33)), ((130, 516), (33, 17)),
                                          ((90, 132), (20, 31)), ((64, 256), (32, 32))],
                         ids=["33x33", "25x33", "33x17", "20x31", "32x32"])
def test_33_tap_frame_matches_the_tile_kernel_and_float64(jd_option, shape, kshape):
    """PSFs wider than 17 taps walk in the 33-tap frame (two columns per lane, 36 accumulator rows): forward model +
    Poisson pass, plain convolution and adjoint against the tile kernel of csrc/sepconv.hip on the same inputs, the plain
    convolution against float64."""
    from scipy.signal import fftconvolve

    from jolideco_amd.data import gaussian_kernel
    from test_gpu_kernels import _separable_step_outputs

    psf = gaussian_kernel(3.1, kshape)
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


def test_the_frame_follows_the_nonzero_taps_not_the_array_size(jd_option):
    """jd_conv_operator_walk_frame: a 17x17 (or 9x13) PSF embedded in a 33x33 array of zeros walks in the 17-tap frame --
    and gives the bits of the same PSF on its own 17x17 plan; a full 33x33 PSF takes the 33-tap frame; a rank-2 PSF or
    a buffer the library did not build none.  jd_conv_operator_forget: a freed operator is unknown again."""
    from jolideco_amd import _hip
    from jolideco_amd.data import gaussian_kernel
    from jolideco_amd.ops import ConvPlan

    jd_option("JD_SEP_WALK", 1)
    H, W = 96, 260
    big = ConvPlan(H, W, 33, 33, DEV, method="separable")
    small = ConvPlan(H, W, 17, 17, DEV, method="separable")
    to_dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV)  # noqa: E731
    g17 = gaussian_kernel(2.0, (17, 17)).astype(np.float32)
    k_embedded = big.psf_spectrum(to_dev(_embed(g17, (33, 33))))
    k_own = small.psf_spectrum(to_dev(g17))
    k_913 = big.psf_spectrum(to_dev(_embed(gaussian_kernel(1.5, (9, 13)), (33, 33))))
    k_full = big.psf_spectrum(to_dev(gaussian_kernel(3.2, (33, 33))))
    k_off = big.psf_spectrum(to_dev(np.roll(_embed(g17, (33, 33)), 5, axis=0)))  # 17 taps, but off centre: 33-tap frame
    rank2 = 0.6 * gaussian_kernel(1.5, (33, 33)) + 0.4 * gaussian_kernel(4.0, (33, 33))
    k_rank2 = big.psf_spectrum(to_dev(rank2))
    assert big.walk_frame(k_embedded) == 17 and small.walk_frame(k_own) == 17 and big.walk_frame(k_913) == 17
    assert big.walk_frame(k_full) == 33 and big.walk_frame(k_off) == 33
    assert big.walk_frame(k_rank2) == 0 and big.walk_frame(k_full.clone()) == 0
    image, scale = torch.rand(H, W, device=DEV) + 0.5, torch.rand(H, W, device=DEV) + 0.5
    assert torch.equal(big.conv_same(image, scale, k_embedded), small.conv_same(image, scale, k_own))
    assert torch.equal(big.conv_same_adjoint(image, scale, k_embedded), small.conv_same_adjoint(image, scale, k_own))
    address = k_full.data_ptr()
    del k_full
    assert _hip.lib().jd_conv_operator_walk_frame(big._handle, address) == 0
    big.close(), small.close()


@pytest.mark.parametrize("walk", [1, 0])
def test_an_operator_overwritten_in_place_by_a_wider_one_is_reported_not_trimmed(jd_option, walk):
    """Operator buffers are immutable (round-4 advice): the library remembers, by device address, the support of the
    operator it built there, and launches on the 17-tap frame / the trimmed tile window from that record.  A registered
    buffer overwritten IN PLACE by an operator with wider support (`khat.copy_(other)`) carries its own record in its
    header; the kernels compare, and the next library call raises instead of silently dropping the outer taps."""
    from jolideco_amd.data import gaussian_kernel
    from jolideco_amd.ops import ConvPlan

    jd_option("JD_SEP_WALK", walk)
    H, W = 96, 260
    plan = ConvPlan(H, W, 33, 33, DEV, method="separable")
    to_dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV)  # noqa: E731
    k_narrow = plan.psf_spectrum(to_dev(_embed(gaussian_kernel(2.0, (17, 17)), (33, 33))))
    k_wide = plan.psf_spectrum(to_dev(gaussian_kernel(3.2, (33, 33))))
    image, scale = torch.rand(H, W, device=DEV) + 0.5, torch.rand(H, W, device=DEV) + 0.5
    ok = plan.conv_same(image, scale, k_narrow)
    torch.cuda.synchronize()
    k_narrow.copy_(k_wide)  # the registry still holds the 17-tap support for this address
    plan.conv_same(image, scale, k_narrow)  # launched on the narrow frame / window: the guard trips on the device
    torch.cuda.synchronize()
    with pytest.raises(RuntimeError, match="immutable"):
        plan.conv_same(image, scale, k_wide)
    # the flag is consumed by the report; a correct operator works again, and rebuilding INTO the address re-registers it
    wide = plan.conv_same(image, scale, k_wide)
    rebuilt = plan.psf_spectrum(to_dev(gaussian_kernel(3.2, (33, 33))), out=k_narrow)
    assert rebuilt is k_narrow and plan.walk_frame(k_narrow) == 33
    assert torch.equal(plan.conv_same(image, scale, k_narrow), wide)
    assert not torch.equal(ok, wide)
    plan.close()


def _mixed_batch(shape, frames, seed):
    """Operators, exposures, backgrounds, counts of len(frames) observations on ONE 33x33 separable plan; frames[i] = 17:
    a 17x17 (or smaller) Gaussian embedded in zeros, 33: a 33x33 Gaussian."""
    from jolideco_amd.data import gaussian_kernel
    from jolideco_amd.ops import ConvPlan, stirling_mean

    H, W = shape
    rs = np.random.RandomState(seed)
    plan = ConvPlan(H, W, 33, 33, DEV, method="separable")
    data = []
    for i, frame in enumerate(frames):
        psf = gaussian_kernel(3.0 + 0.2 * i, (33, 33)) if frame == 33 else _embed(gaussian_kernel(1.3 + 0.2 * i, (17, 17 - 2 * (i % 2))), (33, 33))
        khat = plan.psf_spectrum(torch.from_numpy(psf.astype(np.float32)).to(DEV))
        assert plan.walk_frame(khat) == frame
        exposure = (1.0 + 0.1 * i) * (1.0 + 0.4 * np.linspace(-1, 1, H)[:, None] * np.ones(shape))
        counts = rs.poisson(5.0, size=shape).astype(np.float32)
        data.append((khat, torch.from_numpy(exposure.astype(np.float32)).to(DEV), torch.full(shape, 0.5 + 0.1 * i, device=DEV),
                     torch.from_numpy(counts).to(DEV), stirling_mean(counts)))
    flux = torch.from_numpy(rs.gamma(5.0, size=shape).astype(np.float32)).to(DEV)
    return plan, data, flux


@pytest.mark.parametrize("shape,frames", [((100, 260), (17, 17, 17, 17, 17, 17, 33, 33)), ((130, 300), (33, 17, 33)),
                                          ((64, 128), (17, 33)), ((200, 516), (33, 33, 33, 33, 33, 33, 33, 17, 17, 33, 17)),
                                          ((96, 256), (33, 33, 33, 33, 33, 33, 33, 33, 33))],
                         ids=["6+2", "33-17-33", "17-33", "11obs", "9x33"])
@pytest.mark.parametrize("kernels", ["walk", "tile"])
def test_batched_step_over_both_frames_equals_the_per_dataset_loop_bit_for_bit(jd_option, shape, frames, kernels):
    """One batched joint step over operators of both frames -- forward launch: the 17-tap datasets at four columns per
    lane and the 33-tap datasets at two in ONE grid; adjoint: one launch per run of consecutive datasets of one frame, so
    that the contributions are added in dataset order -- gives the gradient of the per-dataset calls bit for bit and
    their losses to rounding (the tile kernel on the same plan likewise), and agrees with the tile kernel to rounding."""
    plan, data, flux = _mixed_batch(shape, frames, seed=sum(shape) + len(frames))
    n_obs = len(frames)

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

    jd_option("JD_SEP_WALK", 1 if kernels == "walk" else 0)
    grad_b, loss_b = step(True)
    grad_l, loss_l = step(False)
    assert np.array_equal(grad_b, grad_l)
    np.testing.assert_allclose(loss_b, loss_l, rtol=1e-6)
    assert np.abs(grad_b - 0.25).max() > 0
    if kernels == "walk":
        jd_option("JD_SEP_WALK", 0)
        grad_t, loss_t = step(True)
        assert np.abs(grad_b - grad_t).max() < 1e-7  # (a few ulps of the 0.25 the gradient image started from)
        np.testing.assert_allclose(loss_b, loss_t, rtol=1e-6)
    # forward only (the trace evaluation): no gradient image is touched
    losses = [torch.zeros(1, device=DEV) for _ in range(n_obs)]
    plan.npred_poisson_batch_fwd_bwd(flux, [d[1] for d in data], [d[0] for d in data], [d[2] for d in data],
                                     [d[3] for d in data], [d[4] for d in data], losses)
    np.testing.assert_allclose(np.array([float(v) for v in losses]), loss_l, rtol=1e-6)
    plan.close()


@pytest.mark.parametrize("kernels", ["walk", "tile"])
def test_datasets_with_different_psf_sizes_share_the_batched_joint_step(monkeypatch, jd_option, kernels):
    """MAPDeconvolver(fit_mode="joint") on observations with 17x17, 9x9 and 33x33 PSFs: the PSFs are embedded up to
    33x33 (`common_kernel_shape`), every dataset has the same separable plan, the step is batched -- and the fit is the
    per-dataset loop's bit for bit and the oracle's (which convolves every PSF at its own size) to 1e-5."""
    from jolideco_amd import GMMPatchPrior, MAPDeconvolver, SpatialFluxComponent
    from jolideco_amd.data import gaussian_kernel, synthetic_gmm, synthetic_observations
    from jolideco_amd.priors.patches import GaussianMixtureModel, GaussianMixtureModelMeta
    from oracle import cpu_ref

    jd_option("JD_SEP_WALK", 1 if kernels == "walk" else 0)
    datasets, _, flux_init = synthetic_observations(shape=(96, 132), n_obs=8, seed=5)
    datasets["obs-2"]["psf"] = gaussian_kernel(1.2, (9, 9)).astype(np.float32)
    assert {d["psf"].shape for d in datasets.values()} == {(17, 17), (9, 9), (33, 33)}
    means, covs, weights = synthetic_gmm(8, 64, seed=3)
    results = {}
    for mode in ("batch", "loop"):
        if mode == "loop":
            monkeypatch.setenv("JOLIDECO_NO_BATCH", "1")
        gmm = GaussianMixtureModel.from_numpy(means, covs, weights, meta=GaussianMixtureModelMeta(stride=4))
        comp = SpatialFluxComponent.from_numpy(flux=flux_init, prior=GMMPatchPrior(gmm=gmm))
        deconvolver = MAPDeconvolver(n_epochs=4, display_progress=False, device=DEV, fit_mode="joint")
        session = deconvolver.session(datasets, components=comp)
        models = session.total_loss.poisson_loss.npred_models_all
        assert len({id(m.plan) for m in models}) == 1 and models[0].plan.method == "separable"
        assert (models[0].plan.kh, models[0].plan.kw) == (33, 33)
        assert session.batch_joint == (mode == "batch")
        if kernels == "walk":
            assert [m.plan.walk_frame(m["flux"].khat) for m in models] == [17] * 6 + [33] * 2
        res = deconvolver.run(datasets, components=comp)
        results[mode] = res.flux_total
    assert np.array_equal(results["batch"], results["loop"])
    gmm_o = cpu_ref.GMM.from_numpy(means, covs, weights, stride=4)
    final, _ = cpu_ref.map_fit_joint(datasets, {"flux": flux_init}, {"flux": cpu_ref.GMMPatchPriorRef(gmm_o)}, n_epochs=4)
    assert rel_linf(results["batch"], final["flux"]) < 1e-5


@pytest.mark.parametrize("kernels", ["walk", "tile"])
def test_two_components_with_psfs_of_both_frames_match_the_loop_and_the_oracle(monkeypatch, jd_option, kernels):
    """BASELINE config 5 in small with the section-8(d) PSF sizes: "extended" sees 17x17 PSFs and, from observation 6 on,
    33x33 ones; "points" 17x17 throughout.  The batched multi-component step (walk_multi_kernel: one wave per component,
    each in the frame ITS operator needs; adjoint per component, one launch per run of datasets of one frame) is the
    per-dataset loop bit for bit and the oracle's joint fit to 1e-5."""
    from jolideco_amd import FluxComponents, InverseGammaPrior, MAPDeconvolver, SpatialFluxComponent, UniformPrior
    from jolideco_amd.data import gaussian_kernel, synthetic_observations
    from oracle import cpu_ref

    jd_option("JD_SEP_WALK", 1 if kernels == "walk" else 0)
    datasets, _, flux_init = synthetic_observations(shape=(72, 136), n_obs=9, seed=11)
    for i, d in enumerate(datasets.values()):
        d["psf"] = {"extended": d["psf"], "points": gaussian_kernel(1.0 + 0.1 * i, (17, 17)).astype(np.float32)}
    results = {}
    for mode in ("batch", "loop"):
        if mode == "loop":
            monkeypatch.setenv("JOLIDECO_NO_BATCH", "1")
        comps = FluxComponents()
        comps["extended"] = SpatialFluxComponent.from_numpy(flux=flux_init, prior=UniformPrior())
        comps["points"] = SpatialFluxComponent.from_numpy(flux=0.1 * flux_init, prior=InverseGammaPrior(alpha=10))
        deconvolver = MAPDeconvolver(n_epochs=4, display_progress=False, device=DEV, fit_mode="joint")
        session = deconvolver.session(datasets, components=comps)
        models = session.total_loss.poisson_loss.npred_models_all
        assert session.batch_joint == (mode == "batch") and len({id(m.plan) for m in models}) == 1
        if kernels == "walk":
            assert [models[i].plan.walk_frame(models[i]["extended"].khat) for i in (0, 5, 6, 8)] == [17, 17, 33, 33]
            assert all(m.plan.walk_frame(m["points"].khat) == 17 for m in models)
        res = deconvolver.run(datasets, components=comps)
        results[mode] = {name: res.components[name].flux_upsampled_numpy for name in ("extended", "points")}
    for name in ("extended", "points"):
        assert np.array_equal(results["batch"][name], results["loop"][name]), name
    final, _ = cpu_ref.map_fit_joint(
        datasets, {"extended": flux_init, "points": 0.1 * flux_init},
        {"extended": cpu_ref.UniformPriorRef(), "points": cpu_ref.InverseGammaPriorRef(alpha=10)}, n_epochs=4)
    for name in ("extended", "points"):
        assert rel_linf(results["batch"][name], final[name]) < 1e-5, name
