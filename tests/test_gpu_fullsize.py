"""GPU, BASELINE.json full sizes (1024^2 / 2048^2 / 4096^2, K = 128): size-independent properties of
the HIP path -- the oracle cannot run at these sizes in seconds."""
import numpy as np
import pytest
import torch

from conftest import rel_linf

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _gmm128():
    from jolideco_amd.data import synthetic_gmm
    from jolideco_amd.priors.patches import GaussianMixtureModel, GaussianMixtureModelMeta

    means, covs, weights = synthetic_gmm(128, 64, seed=0)
    return GaussianMixtureModel.from_numpy(means, covs, weights, meta=GaussianMixtureModelMeta(stride=4))


@pytest.mark.parametrize("edge", [1024, 2048])
def test_convolution_adjoint_identity_and_linearity(edge):
    """<conv(u), g> == <u, adj(g)> and conv(a u + b v) == a conv(u) + b conv(v) at full size, both methods."""
    from jolideco_amd.data import gaussian_kernel
    from jolideco_amd.ops import ConvPlan

    g = torch.Generator(device="cpu").manual_seed(edge)
    u = torch.rand((edge, edge), generator=g).to(DEV)
    v = torch.rand((edge, edge), generator=g).to(DEV)
    go = torch.randn((edge, edge), generator=g).to(DEV)
    scale = (0.5 + torch.rand((edge, edge), generator=g)).to(DEV)
    psf = torch.from_numpy(gaussian_kernel(2.0, (17, 17)).astype(np.float32)).to(DEV)
    outs = {}
    for method in ("direct", "fft"):
        plan = ConvPlan(edge, edge, 17, 17, DEV, method=method)
        khat = plan.psf_spectrum(psf)
        cu, cv = plan.conv_same(u, scale, khat), plan.conv_same(v, scale, khat)
        mix = plan.conv_same(2.0 * u - 0.5 * v, scale, khat)
        assert rel_linf((2.0 * cu - 0.5 * cv).cpu().numpy(), mix.cpu().numpy()) < 2e-6
        adj = plan.conv_same_adjoint(go, scale, khat)
        lhs = float((cu.double() * go.double()).sum())
        rhs = float((u.double() * adj.double()).sum())
        # <A u, g> is a cancelling sum (g ~ N(0, 1): |lhs| ~ 1e2 .. 1e3 from terms that add up to ~1e5 .. 1e6 in
        # magnitude): the bound is relative to the sum of the |terms|.  The fp32 kernels are exact transposes of each
        # other (~1e-10); the split-fp16 direct kernel rounds u and g independently (measured 2e-9).
        assert abs(lhs - rhs) < 1e-7 * float((cu.abs().double() * go.abs().double()).sum())
        # flux conservation of a unit-sum PSF away from the edges
        ones = plan.conv_same(torch.ones_like(u), None, khat)
        assert float((ones[16:-16, 16:-16] - 1).abs().max()) < 1e-5
        outs[method] = cu.cpu().numpy()
        plan.close()
    assert rel_linf(outs["fft"], outs["direct"]) < 1e-5


@pytest.mark.parametrize("edge", [1024, 2048, 4096])
def test_gmm_prior_full_size_properties(edge):
    """Config 2/3/4 prior (K = 128): patch-row shards sum to the whole (value and gradient, 8-way like the
    8-GPU split), run-to-run bit identical, roll equivariance, gradient sums to zero per construction
    (mean-subtracted patches => sum of d logprior / d flux == 0)."""
    handle = _gmm128().handle(DEV)
    g = torch.Generator(device="cpu").manual_seed(7)
    flux = (torch.rand((edge, edge), generator=g) * 3 + 0.5).to(DEV)
    scale = (16 / 64) / (edge * edge)
    n_rows = (edge - 8) // 4 + 1

    def run(shifts, rows=(0, -1), fl=flux, value=None, grad=None, acc=False):
        value = torch.zeros(1, device=DEV) if value is None else value
        grad = torch.zeros_like(fl) if grad is None else grad
        handle.prior_fwd_bwd(fl, 4, shifts, value, scale, grad=grad, grad_coef=scale, patch_rows=rows, accumulate_value=acc)
        return value, grad

    v1, g1 = run((1, -2))
    v2, g2 = run((1, -2))
    assert torch.equal(v1, v2) and torch.equal(g1, g2)  # deterministic reductions / overlap-add
    assert np.isfinite(float(v1)) and bool(torch.isfinite(g1).all())
    # mean-subtracted patches: the gradient of every patch sums to zero
    assert abs(float(g1.double().sum())) < 1e-6 * float(g1.double().abs().sum())
    # 8-way patch-row shards (the multi-GPU split) accumulate to the whole
    pv, pg = torch.zeros(1, device=DEV), torch.zeros_like(flux)
    bounds = [n_rows * r // 8 for r in range(9)]
    for lo, hi in zip(bounds[:-1], bounds[1:]):
        run((1, -2), rows=(lo, hi), value=pv, grad=pg, acc=True)
    np.testing.assert_allclose(float(pv), float(v1), rtol=2e-6)
    assert rel_linf(pg.cpu().numpy(), g1.cpu().numpy()) < 2e-6
    # cycle spin == evaluating the rolled image without a shift, gradient rolled back
    rolled = torch.roll(flux, shifts=(1, -2), dims=(0, 1)).contiguous()
    v3, g3 = run((0, 0), fl=rolled)
    assert torch.equal(v3, v1)
    assert torch.equal(torch.roll(g3, shifts=(-1, 2), dims=(0, 1)), g1)


def test_joint_step_2048_8obs_runs_and_decreases_the_loss():
    """BASELINE config 3 end to end (the bench workload): finite, decreasing total loss, deterministic."""
    import bench

    totals = []
    for _ in range(2):
        session = bench.build_session("c3", torch.device(DEV))
        rows = []
        for _ in range(4):
            session.epoch()
            rows.append(session.scalars.clone())
        torch.cuda.synchronize()
        vals = torch.stack(rows).cpu().numpy()
        assert np.isfinite(vals).all()
        total = vals[:, :8].sum(1) - vals[:, 8]  # sum_d L_d - beta * logprior
        assert np.all(np.diff(total) < 0)
        totals.append(vals)
    assert np.array_equal(totals[0], totals[1])


def test_config5_two_components_16_observations():
    """BASELINE config 5 at full size: 2 flux components with their own PSFs and priors, 16
    observations, joint step.  Finite, decreasing objective; sharding 2 ways (observations + prior rows,
    element-wise prior on rank 0 only) reproduces the unsharded gradient buffer."""
    import bench
    from jolideco_amd.distributed import DistContext

    session = bench.build_session("c5", torch.device(DEV))
    assert session.n_c == 2 and session.n_d == 16
    rows = []
    for _ in range(3):
        session.epoch()
        rows.append(session.scalars.clone())
    torch.cuda.synchronize()
    vals = torch.stack(rows).cpu().numpy()
    assert np.isfinite(vals).all()
    total = vals[:, :16].sum(1) - vals[:, 16:18].sum(1)
    assert np.all(np.diff(total) < 0)
    del session

    # gradient of one joint step: whole == sum of the 2 rank shares (dry-run contexts, same theta)
    grads = []
    for ctx in (None, DistContext(0, 2, dry_run=True), DistContext(1, 2, dry_run=True)):
        sess = bench.build_session("c5", torch.device(DEV), dist=ctx)
        sess.cfg._optimizer_step = lambda states, step: None  # keep the gradient buffer, do not update
        sess.epoch()
        torch.cuda.synchronize()
        grads.append(sess.comm.clone())
        del sess
    whole, part = grads[0], grads[1] + grads[2]
    assert rel_linf(part.cpu().numpy(), whole.cpu().numpy()) < 2e-6
