"""Flux components that share ONE forward operator (the reference's default: `psf` is one array, every component's model
of a dataset is built from the same exposure and PSF, models/npred.py:279-295).  With non-negative PSF, exposure and
fluxes no clip of models/npred.py:194 ever acts, so the batched joint step evaluates the SUM of the component fluxes
through one forward model and one adjoint per dataset (`PoissonLoss.fwd_bwd_batch`, jd_sum_images / jd_copy_image_to).
Checked against autograd of the oracle (which convolves and clips every component, as the reference does) and against
the per-component launches (JOLIDECO_MERGE_COMPONENTS=0)."""
import numpy as np
import pytest
import torch

from conftest import rel_linf
from oracle import cpu_ref

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _session(datasets, flux_init, n_comp=2, linear_last=False):
    from jolideco_amd import FluxComponents, InverseGammaPrior, MAPDeconvolver, SpatialFluxComponent, UniformPrior

    comps = FluxComponents()
    names = ["extended", "points", "third"][:n_comp]
    for j, name in enumerate(names):
        prior = InverseGammaPrior(alpha=10, beta=1.5) if name == "points" else UniformPrior()
        comps[name] = SpatialFluxComponent.from_numpy(
            flux=(0.5 ** j) * flux_init, prior=prior, use_log_flux=not (linear_last and j == n_comp - 1)
        )
    deco = MAPDeconvolver(n_epochs=1, display_progress=False, device=DEV, fit_mode="joint")
    session = deco.session(datasets, components=comps)
    session.cfg._optimizer_step = lambda states, step: None  # keep the gradient buffers, no update
    return session, names


def _step(session, shape, n_comp):
    session.epoch()
    torch.cuda.synchronize()
    n = shape[0] * shape[1]
    comm = session.comm.cpu().numpy()
    return [comm[c * n : (c + 1) * n].reshape(shape).copy() for c in range(n_comp)], comm[n_comp * n :].copy()


@pytest.mark.parametrize("shape,n_obs,n_comp", [((320, 256), 3, 2), ((2048, 2048), 8, 2), ((192, 260), 2, 3)])
def test_components_sharing_the_operator_take_one_forward_model(shape, n_obs, n_comp, monkeypatch):
    from jolideco_amd.data import synthetic_observations

    datasets, _, flux_init = synthetic_observations(shape=shape, n_obs=n_obs, seed=3)
    monkeypatch.delenv("JOLIDECO_MERGE_COMPONENTS", raising=False)
    merged, names = _session(datasets, flux_init, n_comp)
    assert merged.batch_joint and merged.flux_nonneg
    loss = merged.total_loss.poisson_loss
    assert loss.mergeable([li for _, li in merged.local_idx])
    grads, scalars = _step(merged, shape, n_comp)

    monkeypatch.setenv("JOLIDECO_MERGE_COMPONENTS", "0")
    plain, _ = _session(datasets, flux_init, n_comp)
    assert not plain.total_loss.poisson_loss.mergeable([li for _, li in plain.local_idx])
    grads_p, scalars_p = _step(plain, shape, n_comp)
    np.testing.assert_allclose(scalars, scalars_p, rtol=2e-6)
    for name, g, gp in zip(names, grads, grads_p):
        err = rel_linf(g, gp)
        assert err < 2e-6, (name, err)

    if shape[0] * shape[1] * n_obs > 1 << 22:  # (the oracle below at the small sizes only)
        return
    seen = [st.flux_cur.cpu().numpy() for st in merged.states]
    fl = [torch.from_numpy(np.ascontiguousarray(v))[None, None].requires_grad_(True) for v in seen]
    losses = []
    for d in datasets.values():
        value = cpu_ref.DatasetRef.from_numpy(d, names).loss(tuple(fl))
        value.backward()
        losses.append(float(value))
    ig = cpu_ref.InverseGammaPriorRef(alpha=10, beta=1.5)
    value_ig = ig(fl[1])
    (-1.0 * value_ig).backward()
    np.testing.assert_allclose(scalars[:n_obs], np.array(losses), rtol=5e-6)
    for name, g, f in zip(names, grads, fl):
        err = rel_linf(g, f.grad.numpy()[0, 0])
        assert err < 1e-5, (name, err)


def test_the_merge_needs_every_condition(monkeypatch):
    """Per-component PSFs, a PSF with a negative tap, a flux that is its own parameter (it may turn negative): the
    per-component launches, as before."""
    from jolideco_amd.data import gaussian_kernel, synthetic_observations

    monkeypatch.delenv("JOLIDECO_MERGE_COMPONENTS", raising=False)
    shape = (160, 192)
    datasets, _, flux_init = synthetic_observations(shape=shape, n_obs=2, seed=1)
    indices = [0, 1]

    session, _ = _session(datasets, flux_init, 2, linear_last=True)
    assert session.total_loss.poisson_loss.mergeable(indices) and not session.flux_nonneg

    per_component = {k: dict(v) for k, v in datasets.items()}
    for i, d in enumerate(per_component.values()):
        d["psf"] = {"extended": d["psf"], "points": gaussian_kernel(1.0 + 0.1 * i, (17, 17)).astype(np.float32)}
    session, _ = _session(per_component, flux_init, 2)
    assert not session.total_loss.poisson_loss.mergeable(indices)

    negative = {k: dict(v) for k, v in datasets.items()}
    for d in negative.values():
        col = np.array(d["psf"][:, d["psf"].shape[1] // 2], dtype=np.float64)
        col[0] = -1e-3 * col.max()  # rank 1, one negative tap
        psf = np.outer(col, col)
        d["psf"] = (psf / psf.sum()).astype(np.float32)
    session, _ = _session(negative, flux_init, 2)
    assert not session.total_loss.poisson_loss.mergeable(indices)


def test_sum_and_copy_helpers():
    from jolideco_amd import ops

    for n in (1 << 16, 1000 * 7 + 3):
        srcs = [torch.rand(n, device=DEV) for _ in range(4)]
        out = torch.empty(n, device=DEV)
        for k in (1, 2, 3, 4):
            ops.sum_images(out, srcs[:k])
            want = srcs[0].clone()
            for s in srcs[1:k]:
                want = want + s
            assert torch.equal(out, want)
        dsts = [torch.zeros(n, device=DEV) for _ in range(3)]
        ops.copy_image_to(srcs[0], dsts)
        assert all(torch.equal(d, srcs[0]) for d in dsts)
        odd = torch.rand(n + 1, device=DEV)[1:]  # (4-byte aligned only)
        ops.sum_images(out, [odd.contiguous(), srcs[1]])
        assert torch.equal(out, odd + srcs[1])


def test_sequential_fit_of_two_components_sharing_the_operator_matches_the_oracle(monkeypatch):
    """The reference's own loop (fit_mode="sequential", core.py:209-247) with two components on one PSF per dataset: the
    per-dataset call evaluates their sum (`NPredModels.fwd_bwd`); final fluxes and trace against the oracle, which
    convolves and clips every component."""
    from jolideco_amd import FluxComponents, InverseGammaPrior, MAPDeconvolver, SpatialFluxComponent, UniformPrior
    from jolideco_amd.data import synthetic_observations

    monkeypatch.delenv("JOLIDECO_MERGE_COMPONENTS", raising=False)
    shape = (96, 80)
    datasets, _, flux_init = synthetic_observations(shape=shape, n_obs=3, seed=5)
    inits = {"extended": flux_init, "points": 0.25 * flux_init}

    def run():
        comps = FluxComponents()
        comps["extended"] = SpatialFluxComponent.from_numpy(flux=inits["extended"], prior=UniformPrior())
        comps["points"] = SpatialFluxComponent.from_numpy(flux=inits["points"], prior=InverseGammaPrior(alpha=10, beta=1.5))
        return MAPDeconvolver(n_epochs=3, display_progress=False, device=DEV).run(datasets, components=comps)

    res = run()
    final, trace = cpu_ref.map_fit_sequential(
        datasets, inits, {"extended": cpu_ref.UniformPriorRef(), "points": cpu_ref.InverseGammaPriorRef(alpha=10, beta=1.5)},
        n_epochs=3,
    )
    for name in inits:
        err = rel_linf(res.components[name].flux_numpy, final[name])
        assert err < 1e-5, (name, err)
    assert abs(res.trace_loss[-1]["total"] - trace[-1]["total"]) < 2e-5 * abs(trace[-1]["total"])
    monkeypatch.setenv("JOLIDECO_MERGE_COMPONENTS", "0")
    plain = run()
    for name in inits:
        err = rel_linf(res.components[name].flux_numpy, plain.components[name].flux_numpy)
        assert err < 1e-5, (name, err)
