"""Generate tests/golden/*.npz from the LIVE reference (build container only).

Run:  python oracle/refload/make_golden.py
Imports /root/reference/jolideco through load_ref.py, runs the reference's own hot path on
seeded inputs and stores inputs + expected outputs.  The fixtures are data only; no reference
source travels.  While generating, every fixture is also evaluated with oracle/cpu_ref.py and
the agreement is asserted, which pins the oracle against the reference.
"""
import sys
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
REPO = HERE.parent.parent
sys.path.insert(0, str(HERE))
sys.path.insert(0, str(REPO))

from load_ref import load_reference  # noqa: E402

load_reference()

from jolideco.core import MAPDeconvolver  # noqa: E402
from jolideco.data import (  # noqa: E402
    disk_source_gauss_psf,
    gauss_and_point_sources_gauss_psf,
    point_source_gauss_psf,
)
from jolideco.loss import TotalLoss  # noqa: E402
from jolideco.models import FluxComponents, NPredModels, SpatialFluxComponent  # noqa: E402
from jolideco.priors import (  # noqa: E402
    ExponentialPrior,
    GMMPatchPrior,
    InverseGammaPrior,
    UniformPrior,
)
from jolideco.priors.patches.gmm import GaussianMixtureModel, GaussianMixtureModelMeta  # noqa: E402
from jolideco.utils.norms import SubtractMeanPatchNorm  # noqa: E402

from oracle import cpu_ref  # noqa: E402

OUT = REPO / "tests" / "golden"
OUT.mkdir(parents=True, exist_ok=True)


def ref_gmm(means, covs, weights, stride=4):
    meta = GaussianMixtureModelMeta(stride=stride, patch_norm=SubtractMeanPatchNorm())
    return GaussianMixtureModel.from_numpy(means=means, covariances=covs, weights=weights, meta=meta)


def trace_to_arrays(trace):
    return {f"trace/{name}": np.asarray(trace[name], dtype=np.float64) for name in trace.colnames if name != "filename"}


def rows_to_arrays(rows):
    return {f"trace/{name}": np.array([r[name] for r in rows], dtype=np.float64) for name in rows[0]}


def pack_datasets(datasets):
    out = {}
    for name, d in datasets.items():
        for key in ("counts", "exposure", "background"):
            out[f"data/{name}/{key}"] = d[key]
        if isinstance(d["psf"], dict):
            for cname, psf in d["psf"].items():
                out[f"data/{name}/psf/{cname}"] = psf
        else:
            out[f"data/{name}/psf"] = d["psf"]
    return out


def asym_psf(shape, sigma_y, sigma_x, dy=0.6, dx=-0.9, rot=0.5):
    """Deliberately asymmetric, off-centre, rotated PSF (catches conv/corr flips)."""
    kh, kw = shape
    y, x = np.mgrid[0:kh, 0:kw].astype(float)
    y -= (kh - 1) / 2 + dy
    x -= (kw - 1) / 2 + dx
    c, s = np.cos(rot), np.sin(rot)
    u, v = c * x + s * y, -s * x + c * y
    psf = np.exp(-0.5 * ((u / sigma_x) ** 2 + (v / sigma_y) ** 2)) * (1 + 0.3 * np.tanh(u))
    return (psf / psf.sum()).astype(np.float32)


def scene(shape, psf, rs, n_points=6, bkg=1.5):
    h, w = shape
    y, x = np.mgrid[0:h, 0:w].astype(float)
    truth = 2.0 + 40 * np.exp(-0.5 * (((y - h * 0.4) / (h / 9)) ** 2 + ((x - w * 0.55) / (w / 7)) ** 2))
    for _ in range(n_points):
        truth[rs.randint(0, h), rs.randint(0, w)] += rs.uniform(100, 800)
    exposure = (1 + 0.5 * np.linspace(-1, 1, h)).reshape(-1, 1) * (1 + 0.2 * np.linspace(-1, 1, w))
    from scipy.signal import fftconvolve

    npred = fftconvolve(truth * exposure, psf, mode="same") + bkg
    counts = rs.poisson(np.clip(npred, 0, None))
    return {
        "counts": counts.astype(np.float32),
        "psf": psf.astype(np.float32),
        "exposure": exposure.astype(np.float32),
        "background": (bkg * np.ones(shape)).astype(np.float32),
    }


# ---------------------------------------------------------------------------------------
def anchor_a():
    """SURVEY App. A anchor A: 128^2 point source, uniform prior, 50 epochs (config 1)."""
    rs = np.random.RandomState(428723)
    data = point_source_gauss_psf(shape=(128, 128), random_state=rs)
    flux_init = rs.gamma(30, size=(128, 128))
    comp = SpatialFluxComponent.from_numpy(flux=flux_init)
    res = MAPDeconvolver(n_epochs=50, display_progress=False).run({"obs-1": data}, components=comp)
    flux = res.flux_total
    assert abs(flux[64, 64] - 38.512722) < 1e-4 and data["counts"].sum() == 33751

    final, trace = cpu_ref.map_fit_sequential(
        {"obs-1": data}, {"flux": flux_init}, {"flux": cpu_ref.UniformPriorRef()}, n_epochs=50
    )
    assert np.array_equal(final["flux"], flux), np.abs(final["flux"] - flux).max()
    assert trace[-1]["total"] == res.trace_loss[-1]["total"]

    np.savez_compressed(
        OUT / "anchor_a.npz",
        flux_init=flux_init,
        flux_final=flux,
        **pack_datasets({"obs-1": data}),
        **trace_to_arrays(res.trace_loss),
    )
    print("anchor_a ok", flux[64, 64], res.trace_loss[-1]["total"])


def anchor_b():
    """SURVEY App. A anchor B: 64^2, 3 obs, synthetic GMM K=8, 10 epochs, sequential."""
    rs = np.random.RandomState(7)
    K, D = 8, 64
    covs = np.empty((K, D, D))
    for k in range(K):
        a = rs.normal(size=(D, D)) / np.sqrt(D)
        covs[k] = a @ a.T * rs.uniform(0.01, 1.0) + 1e-3 * np.eye(D)
    weights = rs.dirichlet(np.ones(K))
    means = np.zeros((K, D))
    datasets = {
        f"obs-{i}": point_source_gauss_psf(shape=(64, 64), sigma_psf=2 + 0.5 * i, random_state=rs) for i in range(3)
    }
    flux_init = rs.gamma(30, size=(64, 64))
    comp = SpatialFluxComponent.from_numpy(flux=flux_init, prior=GMMPatchPrior(gmm=ref_gmm(means, covs, weights)))
    res = MAPDeconvolver(n_epochs=10, display_progress=False).run(datasets, components=comp)
    flux = res.flux_total
    assert abs(flux[32, 32] - 16.988111) < 1e-4, flux[32, 32]

    gmm = cpu_ref.GMM.from_numpy(means, covs, weights, stride=4)
    final, trace, steps = cpu_ref.map_fit_sequential(
        datasets, {"flux": flux_init}, {"flux": cpu_ref.GMMPatchPriorRef(gmm)}, n_epochs=10, record_steps=True
    )
    assert np.array_equal(final["flux"], flux), np.abs(final["flux"] - flux).max()
    assert trace[-1]["prior-flux"] == res.trace_loss[-1]["prior-flux"]

    np.savez_compressed(
        OUT / "anchor_b.npz",
        gmm_means=means,
        gmm_covariances=covs,
        gmm_weights=weights,
        flux_init=flux_init,
        flux_final=flux,
        grad_step0=steps[0]["grads"][0],
        **pack_datasets(datasets),
        **trace_to_arrays(res.trace_loss),
    )
    print("anchor_b ok", flux[32, 32], res.trace_loss[-1]["total"])


def reference_test_cases():
    """Inputs + outputs of the reference's own known-answer tests
    (jolideco/tests/test_core.py:14-63,71-79,127-153,156-188)."""
    rs = np.random.RandomState(642020)
    datasets_gauss = {f"{i}": gauss_and_point_sources_gauss_psf(random_state=rs) for i in range(3)}
    rs = np.random.RandomState(642020)
    datasets_disk = {f"{i}": disk_source_gauss_psf(random_state=rs) for i in range(3)}
    flux_init = np.random.RandomState(642020).gamma(20, size=(32, 32))

    out = {"flux_init": flux_init}
    # uniform
    comps = FluxComponents()
    comps["flux-1"] = SpatialFluxComponent.from_numpy(flux=flux_init, prior=UniformPrior())
    res = MAPDeconvolver(n_epochs=100, display_progress=False).run(datasets=datasets_gauss, components=comps)
    assert np.isclose(res.flux_total[12, 12], 1.542659, rtol=1e-3)
    out["uniform/flux_final"] = res.flux_total
    out.update({f"uniform/{k}": v for k, v in trace_to_arrays(res.trace_loss).items()})
    out.update({f"gauss/{k}": v for k, v in pack_datasets(datasets_gauss).items()})
    final, trace = cpu_ref.map_fit_sequential(
        datasets_gauss, {"flux-1": flux_init}, {"flux-1": cpu_ref.UniformPriorRef()}, n_epochs=100
    )
    assert np.array_equal(final["flux-1"], res.flux_total)

    # inverse gamma
    for d in datasets_disk.values():
        d["psf"] = {"flux-1": d["psf"]}
    comps = FluxComponents()
    comps["flux-1"] = SpatialFluxComponent.from_numpy(flux=flux_init, prior=InverseGammaPrior(alpha=10))
    res = MAPDeconvolver(n_epochs=100, display_progress=False).run(datasets=datasets_disk, components=comps)
    assert np.isclose(res.flux_total[12, 12], 0.136798, rtol=1e-3)
    out["inverse_gamma/flux_final"] = res.flux_total
    out.update({f"inverse_gamma/{k}": v for k, v in trace_to_arrays(res.trace_loss).items()})
    out.update({f"disk/{k}": v for k, v in pack_datasets(datasets_disk).items()})
    final, trace = cpu_ref.map_fit_sequential(
        datasets_disk, {"flux-1": flux_init}, {"flux-1": cpu_ref.InverseGammaPriorRef(alpha=10)}, n_epochs=100
    )
    assert np.array_equal(final["flux-1"], res.flux_total), np.abs(final["flux-1"] - res.flux_total).max()
    assert trace[-1]["prior-flux-1"] == res.trace_loss[-1]["prior-flux-1"]

    # exponential + validation
    comps = FluxComponents()
    comps["flux-1"] = SpatialFluxComponent.from_numpy(flux=flux_init, prior=ExponentialPrior(alpha=1))
    train = {n: datasets_disk[n] for n in ["0", "1"]}
    val = {n: datasets_disk[n] for n in ["2"]}
    res = MAPDeconvolver(n_epochs=100, display_progress=False).run(
        datasets=train, components=comps, datasets_validation=val
    )
    assert np.isclose(res.flux_total[12, 12], 1.382768, rtol=1e-3)
    out["exponential/flux_final"] = res.flux_total
    out.update({f"exponential/{k}": v for k, v in trace_to_arrays(res.trace_loss).items()})
    final, trace = cpu_ref.map_fit_sequential(
        train,
        {"flux-1": flux_init},
        {"flux-1": cpu_ref.ExponentialPriorRef(alpha=1)},
        n_epochs=100,
        datasets_validation=val,
    )
    assert np.array_equal(final["flux-1"], res.flux_total)
    assert trace[-1]["datasets-validation-total"] == res.trace_loss[-1]["datasets-validation-total"]

    np.savez_compressed(OUT / "reference_tests.npz", **out)
    print("reference_test_cases ok")


def upsampling_case():
    """The reference's up-sampling known-answer test (jolideco/tests/test_core.py:99-124): disk
    datasets, flux component with upsampling_factor=2, uniform prior, 100 epochs; plus a short
    fit with a GMM prior on the up-sampled flux and an odd factor (3)."""
    rs = np.random.RandomState(642020)
    datasets_disk = {f"{i}": disk_source_gauss_psf(random_state=rs) for i in range(3)}
    flux_init = np.random.RandomState(642020).gamma(20, size=(32, 32))
    comps = FluxComponents()
    comps["flux-1"] = SpatialFluxComponent.from_numpy(flux=flux_init, upsampling_factor=2, prior=UniformPrior())
    res = MAPDeconvolver(n_epochs=100, learning_rate=0.1, display_progress=False).run(
        datasets=datasets_disk, components=comps
    )
    assert res.flux_upsampled_total.shape == (64, 64)
    assert np.isclose(res.flux_total[12, 12], 3.565998, rtol=1e-3) and np.isclose(res.flux_total[0, 0], 1.605782, rtol=1e-3)
    assert np.isclose(res.trace_loss[-1]["total"], 5.844786, rtol=1e-3)
    final, trace = cpu_ref.map_fit_sequential(
        datasets_disk, {"flux-1": flux_init}, {"flux-1": cpu_ref.UniformPriorRef()}, n_epochs=100,
        upsampling_factors={"flux-1": 2},
    )
    assert np.array_equal(final["flux-1"], res.flux_upsampled_total), np.abs(final["flux-1"] - res.flux_upsampled_total).max()
    assert trace[-1]["total"] == res.trace_loss[-1]["total"]
    out = {"flux_init": flux_init, "u2/flux_upsampled_final": res.flux_upsampled_total, "u2/flux_final": res.flux_total}
    out.update({f"u2/{k}": v for k, v in trace_to_arrays(res.trace_loss).items()})
    out.update({f"disk/{k}": v for k, v in pack_datasets(datasets_disk).items()})

    # GMM prior on the up-sampled flux, factor 3, 2 observations with a small 5x5 PSF
    rs = np.random.RandomState(31)
    means, covs, weights = cpu_ref.synthetic_gmm(6, 64, seed=8)
    shape = (24, 28)
    datasets = {f"o{i}": scene(shape, asym_psf((5, 5), 1.0 + 0.2 * i, 1.3), rs, n_points=3, bkg=1.0) for i in range(2)}
    init3 = rs.gamma(30, size=shape)
    comps = FluxComponents()
    comps["flux"] = SpatialFluxComponent.from_numpy(
        flux=init3, upsampling_factor=3, prior=GMMPatchPrior(gmm=ref_gmm(means, covs, weights))
    )
    res3 = MAPDeconvolver(n_epochs=5, display_progress=False).run(datasets=datasets, components=comps)
    gmm_o = cpu_ref.GMM.from_numpy(means, covs, weights, stride=4)
    final, trace = cpu_ref.map_fit_sequential(
        datasets, {"flux": init3}, {"flux": cpu_ref.GMMPatchPriorRef(gmm_o)}, n_epochs=5, upsampling_factors={"flux": 3}
    )
    assert np.array_equal(final["flux"], res3.flux_upsampled_total)
    assert trace[-1]["total"] == res3.trace_loss[-1]["total"]
    out.update({"u3/flux_init": init3, "u3/flux_upsampled_final": res3.flux_upsampled_total, "u3/flux_final": res3.flux_total,
                "u3/gmm_means": means, "u3/gmm_covariances": covs, "u3/gmm_weights": weights})
    out.update({f"u3/{k}": v for k, v in pack_datasets(datasets).items()})
    out.update({f"u3/{k}": v for k, v in trace_to_arrays(res3.trace_loss).items()})
    np.savez_compressed(OUT / "upsampling.npz", **out)
    print("upsampling_case ok", res.flux_total[12, 12], res3.trace_loss[-1]["total"])


def upsampling_mixed_psf_case():
    """upsampling_factor=2 with two flux components whose PSFs have DIFFERENT shapes (9x9 with non-negligible edges,
    5x5 box-like): the reference up-samples each PSF as given (models/npred.py:96-106, F.interpolate clamps at the
    array edge), so an implementation that first embeds the small PSF in the shape of the large one gets another border."""
    rs = np.random.RandomState(31)
    shape = (40, 44)
    means, covs, weights = cpu_ref.synthetic_gmm(6, 64, seed=12)
    datasets = {}
    for i in range(2):
        d = scene(shape, asym_psf((9, 9), 2.5 + 0.5 * i, 3.0), rs, bkg=0.6)
        small = np.ones((5, 5)) + 0.3 * rs.uniform(size=(5, 5))  # strong edges: the up-sampled border matters
        d["psf"] = {"extended": d["psf"], "points": (small / small.sum()).astype(np.float32)}
        datasets[f"o{i}"] = d
    init_ext = rs.gamma(30, size=shape)
    init_pts = rs.gamma(2, size=shape) * 0.2
    comps = FluxComponents()
    comps["extended"] = SpatialFluxComponent.from_numpy(
        flux=init_ext, upsampling_factor=2, prior=GMMPatchPrior(gmm=ref_gmm(means, covs, weights))
    )
    comps["points"] = SpatialFluxComponent.from_numpy(flux=init_pts, upsampling_factor=2, prior=InverseGammaPrior(alpha=10, beta=1.5))
    n_epochs = 5
    res = MAPDeconvolver(n_epochs=n_epochs, display_progress=False).run(datasets=datasets, components=comps)
    gmm_o = cpu_ref.GMM.from_numpy(means, covs, weights, stride=4)
    final, trace = cpu_ref.map_fit_sequential(
        datasets, {"extended": init_ext, "points": init_pts},
        {"extended": cpu_ref.GMMPatchPriorRef(gmm_o), "points": cpu_ref.InverseGammaPriorRef(10, 1.5)},
        n_epochs=n_epochs, upsampling_factors={"extended": 2, "points": 2},
    )
    up = {name: comp.flux_upsampled.detach().numpy()[0, 0] for name, comp in res.components.items()}
    assert np.array_equal(final["extended"], up["extended"]) and np.array_equal(final["points"], up["points"])
    assert trace[-1]["total"] == res.trace_loss[-1]["total"]
    out = {"gmm/means": means, "gmm/covariances": covs, "gmm/weights": weights, "init/extended": init_ext,
           "init/points": init_pts, "final_upsampled/extended": up["extended"], "final_upsampled/points": up["points"]}
    out.update(pack_datasets(datasets))
    out.update(trace_to_arrays(res.trace_loss))
    np.savez_compressed(OUT / "upsampling_mixed_psf.npz", **out)
    print("upsampling_mixed_psf_case ok", res.trace_loss[-1]["total"])


def calibration_case():
    """Fits with NPredCalibrations (jolideco/models/npred.py:298-510): trainable sub-pixel shift and
    background norm per dataset, a fixed PSF scale, one frozen calibration; upsampling_factor 1 and 2."""
    from jolideco.models import NPredCalibration, NPredCalibrations

    out = {}
    for tag, u, n_epochs in (("u1", 1, 8), ("u2", 2, 5)):
        rs = np.random.RandomState(40 + u)
        shape = (40, 44)
        datasets = {f"o{i}": scene(shape, asym_psf((7, 7), 1.1 + 0.2 * i, 1.4), rs, n_points=4, bkg=0.8 + 0.1 * i) for i in range(3)}
        flux_init = rs.gamma(30, size=shape)
        means, covs, weights = cpu_ref.synthetic_gmm(6, 64, seed=9)
        spec = {
            "o0": dict(shift_x=0.3, shift_y=-0.2, background_norm=1.2, psf_scale=1.0),
            "o1": dict(shift_x=0.0, shift_y=0.0, background_norm=0.9, psf_scale=1.1),  # shift stays exactly 0
            "o2": dict(shift_x=-0.45, shift_y=0.6, background_norm=1.0, psf_scale=1.0, frozen=True),
        }
        cals = NPredCalibrations()
        for name, kw in spec.items():
            cals[name] = NPredCalibration(**kw)
        comps = FluxComponents()
        comps["flux"] = SpatialFluxComponent.from_numpy(
            flux=flux_init, upsampling_factor=u, prior=GMMPatchPrior(gmm=ref_gmm(means, covs, weights))
        )
        res = MAPDeconvolver(n_epochs=n_epochs, display_progress=False).run(datasets=datasets, components=comps, calibrations=cals)
        cal_final = {name: c.to_dict() for name, c in res.calibrations.items()}

        gmm_o = cpu_ref.GMM.from_numpy(means, covs, weights, stride=4)
        cals_o = {name: cpu_ref.CalibrationRef.create(**kw) for name, kw in spec.items()}
        final, trace = cpu_ref.map_fit_sequential(
            datasets, {"flux": flux_init}, {"flux": cpu_ref.GMMPatchPriorRef(gmm_o)}, n_epochs=n_epochs,
            upsampling_factors={"flux": u}, calibrations=cals_o,
        )
        assert np.array_equal(final["flux"], res.flux_upsampled_total), np.abs(final["flux"] - res.flux_upsampled_total).max()
        assert trace[-1]["total"] == res.trace_loss[-1]["total"]
        for name in spec:
            for key, value in cals_o[name].to_dict().items():
                assert value == cal_final[name][key], (name, key, value, cal_final[name][key])
        assert cal_final["o1"]["shift_x"] == 0.0 and cal_final["o0"]["shift_x"] != 0.3 and cal_final["o2"]["shift_x"] == np.float32(-0.45)

        out.update({f"{tag}/flux_init": flux_init, f"{tag}/flux_upsampled_final": res.flux_upsampled_total,
                    f"{tag}/gmm_means": means, f"{tag}/gmm_covariances": covs, f"{tag}/gmm_weights": weights})
        out.update({f"{tag}/{k}": v for k, v in pack_datasets(datasets).items()})
        out.update({f"{tag}/{k}": v for k, v in trace_to_arrays(res.trace_loss).items()})
        for name, kw in spec.items():
            out[f"{tag}/cal_init/{name}"] = np.array([kw["shift_x"], kw["shift_y"], kw["background_norm"], kw["psf_scale"],
                                                      float(kw.get("frozen", False))])
            d = cal_final[name]
            out[f"{tag}/cal_final/{name}"] = np.array([d["shift_x"], d["shift_y"], d["background_norm"], d["psf_scale"]])
        print("calibration_case", tag, "ok", cal_final["o0"], res.trace_loss[-1]["total"])
    np.savez_compressed(OUT / "calibration.npz", **out)


def linear_flux_case():
    """use_log_flux=False (models/core.py:399-402,583-594): the parameter is the flux itself; without a
    mask the trace sees the post-step flux (the parameter object is what `to_flux_tuple` returns)."""
    out = {}
    for tag, with_mask in (("nomask", False), ("mask", True)):
        rs = np.random.RandomState(51)
        shape = (36, 40)
        datasets = {f"o{i}": scene(shape, asym_psf((7, 7), 1.2 + 0.2 * i, 1.5), rs, n_points=3, bkg=1.0) for i in range(2)}
        flux_init = rs.gamma(30, size=shape)
        mask = np.ones(shape, dtype=bool)
        mask[:, :6] = False
        means, covs, weights = cpu_ref.synthetic_gmm(5, 64, seed=12)
        comps = FluxComponents()
        comps["flux"] = SpatialFluxComponent.from_numpy(
            flux=flux_init, mask=mask if with_mask else None, use_log_flux=False,
            prior=GMMPatchPrior(gmm=ref_gmm(means, covs, weights)),
        )
        res = MAPDeconvolver(n_epochs=6, display_progress=False).run(datasets=datasets, components=comps)
        gmm_o = cpu_ref.GMM.from_numpy(means, covs, weights, stride=4)
        final, trace = cpu_ref.map_fit_sequential(
            datasets, {"flux": flux_init}, {"flux": cpu_ref.GMMPatchPriorRef(gmm_o)}, n_epochs=6,
            masks={"flux": mask} if with_mask else None, use_log_flux=False,
        )
        assert np.array_equal(final["flux"], res.flux_total), np.abs(final["flux"] - res.flux_total).max()
        assert trace[-1]["total"] == res.trace_loss[-1]["total"], (trace[-1]["total"], res.trace_loss[-1]["total"])
        out.update({f"{tag}/flux_init": flux_init, f"{tag}/flux_final": res.flux_total, f"{tag}/mask": mask,
                    f"{tag}/gmm_means": means, f"{tag}/gmm_covariances": covs, f"{tag}/gmm_weights": weights})
        out.update({f"{tag}/{k}": v for k, v in pack_datasets(datasets).items()})
        out.update({f"{tag}/{k}": v for k, v in trace_to_arrays(res.trace_loss).items()})
        print("linear_flux_case", tag, "ok", res.trace_loss[-1]["total"], res.flux_total.min())
    np.savez_compressed(OUT / "linear_flux.npz", **out)


def compute_error_case():
    """The reference's flux-error known-answer test (jolideco/tests/test_core.py:249-272): disk datasets,
    InverseGammaPrior(alpha=0.1), 100 epochs, compute_error=True; plus the same with a uniform prior (no curvature
    -> inf everywhere).  Only the prior terms reach the Hessian in the reference (loss.py:71 detaches the dataset
    losses), which the analytic restatement below checks."""
    rs = np.random.RandomState(642020)
    datasets_disk = {f"{i}": disk_source_gauss_psf(random_state=rs) for i in range(3)}
    flux_init = np.random.RandomState(642020).gamma(20, size=(32, 32))
    out = {"flux_init": flux_init}
    out.update({f"disk/{k}": v for k, v in pack_datasets(datasets_disk).items()})
    for tag, prior in (("inverse_gamma", InverseGammaPrior(alpha=0.1)), ("uniform", UniformPrior())):
        comps = FluxComponents()
        comps["flux-1"] = SpatialFluxComponent.from_numpy(flux=flux_init, upsampling_factor=1, prior=prior)
        res = MAPDeconvolver(n_epochs=100, learning_rate=0.1, display_progress=False, compute_error=True).run(
            datasets=datasets_disk, components=comps
        )
        err = res.components["flux-1"].flux_upsampled_error_numpy
        out[f"{tag}/flux_final"] = res.flux_total
        out[f"{tag}/flux_error"] = err
        out.update({f"{tag}/{k}": v for k, v in trace_to_arrays(res.trace_loss).items()})
    err = out["inverse_gamma/flux_error"]
    assert np.isclose(err[3, 3], 24.106102, rtol=1e-3)
    assert np.all(np.isinf(out["uniform/flux_error"]))
    # analytic restatement; the reference evaluates it on the flux BEFORE the last step, which is not observable from
    # outside, hence the loose tolerance against the final flux here (tests/test_gpu_fit.py compares tightly)
    f = out["inverse_gamma/flux_final"].astype(np.float64)
    h = -(-2.0 * 1.5 / f**3 + 1.1 / f**2) / f.size
    with np.errstate(invalid="ignore", divide="ignore"):
        approx = np.sqrt(1.0 / h)
    both = np.isfinite(err) & np.isfinite(approx)
    assert both.sum() > 0.9 * f.size and np.allclose(err[both], approx[both], rtol=0.1)
    np.savez_compressed(OUT / "compute_error.npz", **out)
    print("compute_error_case ok", err[3, 3], int(np.isnan(err).sum()))


def stage_vectors():
    """Per-stage vectors: npred / loss / dL/dtheta and GMM prior value / grad / arg-max for
    several shapes incl. non-square images, even-sized and asymmetric PSFs, sizes that leave a
    patch remainder, every cycle-spin shift, means != 0, marginalize on/off."""
    cases = {
        "sq96_psf17": dict(shape=(96, 96), psf=asym_psf((17, 17), 2.0, 3.0), seed=11),
        "rect80x112_psf12x16": dict(shape=(80, 112), psf=asym_psf((12, 16), 1.5, 2.5), seed=12),
        "rect97x110_psf9x5": dict(shape=(97, 110), psf=asym_psf((9, 5), 1.2, 0.9), seed=13),
        "sq256_psf33": dict(shape=(256, 256), psf=asym_psf((33, 33), 4.0, 5.0), seed=14),
    }
    out = {}
    for name, cfg in cases.items():
        rs = np.random.RandomState(cfg["seed"])
        data = scene(cfg["shape"], cfg["psf"], rs)
        flux = rs.gamma(3, size=cfg["shape"]).astype(np.float32) * 3
        # make some npred pixels clip at zero?  conv of positive flux is >=0 up to rounding; keep.
        comps = FluxComponents()
        comps["flux"] = SpatialFluxComponent.from_numpy(flux=flux)
        theta = comps["flux"]._flux_upsampled.detach().numpy()[0, 0].copy()
        models = NPredModels.from_dataset_numpy(dataset=data, components=comps)
        fluxes = comps.to_flux_tuple()
        npred = models.evaluate(fluxes=fluxes)
        loss_fn = torch.nn.PoissonNLLLoss(log_input=False, reduction="mean", eps=1e-25, full=True)
        counts = torch.from_numpy(data["counts"][None, None])
        loss = loss_fn(npred, counts)
        loss.backward()
        grad = comps["flux"]._flux_upsampled.grad.numpy()[0, 0]

        o_loss, o_npred, o_grad = cpu_ref.poisson_loss_and_grad(theta, data)
        assert o_loss == float(loss) and np.array_equal(o_npred, npred.detach().numpy()[0, 0])
        assert np.array_equal(o_grad, grad)
        assert abs(cpu_ref.poisson_nll_numpy(o_npred, data["counts"]) - o_loss) < 2e-6 * abs(o_loss)

        out[f"{name}/theta"] = theta
        out[f"{name}/npred"] = npred.detach().numpy()[0, 0]
        out[f"{name}/exposure_corrected"] = models["flux"].exposure.numpy()[0, 0]
        out[f"{name}/loss"] = np.float64(loss)
        out[f"{name}/grad_theta"] = grad
        out.update({f"{name}/{k}": v for k, v in pack_datasets({"d": data}).items()})

        # GMM prior on the same flux
        for gname, K, zero_means, gseed in (("k16", 16, True, 3), ("k5m", 5, False, 4)):
            means, covs, weights = cpu_ref.synthetic_gmm(K, 64, seed=gseed, zero_means=zero_means)
            gmm_r = ref_gmm(means, covs, weights)
            gmm_o = cpu_ref.GMM.from_numpy(means, covs, weights, stride=4)
            out[f"gmm/{gname}/means"] = means
            out[f"gmm/{gname}/covariances"] = covs
            out[f"gmm/{gname}/weights"] = weights
            shift_list = [(0, 0), (-2, 2), (2, -1), (1, 0)] if name != "sq256_psf33" else [(-1, -2)]
            for sy, sx in shift_list:
                for marg in (False, True):
                    prior = GMMPatchPrior(gmm=gmm_r, cycle_spin=False, marginalize=marg)
                    f = torch.from_numpy(flux[None, None]).requires_grad_(True)
                    rolled = torch.roll(f, shifts=(sy, sx), dims=(2, 3))
                    # numel identical, so prior(rolled) == cycle-spun prior with this shift
                    value = prior(rolled)
                    value.backward()
                    g = f.grad.numpy()[0, 0]
                    o_val, o_g, o_arg = cpu_ref.gmm_prior_value_and_grad(flux, gmm_o, 4, (sy, sx), marg)
                    assert o_val == float(value), (o_val, float(value))
                    assert np.array_equal(o_g, g)
                    key = f"{name}/prior/{gname}/s{sy}_{sx}/{'lse' if marg else 'max'}"
                    out[f"{key}/value"] = np.float64(value)
                    out[f"{key}/grad_flux"] = g
                    if not marg:
                        with torch.no_grad():
                            ll = prior._evaluate_log_like(torch.roll(f.detach(), shifts=(sy, sx), dims=(2, 3)))
                        arg = torch.argmax(ll, dim=1).numpy().astype(np.int32)
                        assert np.array_equal(arg, o_arg)
                        out[f"{key}/argmax"] = arg
                        # margin between best and runner-up (lets tests skip near-ties)
                        top2 = torch.topk(ll, 2, dim=1).values
                        out[f"{key}/margin"] = (top2[:, 0] - top2[:, 1]).numpy()
    np.savez_compressed(OUT / "stages.npz", **out)
    print("stage_vectors ok", len(out))


def rng_draws():
    """Draw order of cycle_spin with torch's default CPU generator seed
    (jolideco/utils/torch.py:108-116, 393-414)."""
    from jolideco.utils.torch import cycle_spin, get_default_generator

    gen = get_default_generator("cpu")
    assert gen.initial_seed() == cpu_ref.TORCH_DEFAULT_GENERATOR_SEED
    img = torch.arange(12 * 12, dtype=torch.float32).reshape(1, 1, 12, 12)
    shifts = []
    for _ in range(32):
        rolled = cycle_spin(img, (8, 8), gen)
        # recover (sy, sx) from where pixel 0 went
        idx = int(torch.argmin(rolled))
        sy, sx = idx // 12, idx % 12
        shifts.append((sy if sy <= 6 else sy - 12, sx if sx <= 6 else sx - 12))
    gen2 = torch.Generator(device="cpu")
    mine = [cpu_ref.draw_cycle_spin_shifts(gen2, (8, 8)) for _ in range(32)]
    assert mine == shifts, (mine, shifts)
    np.savez_compressed(OUT / "rng_draws.npz", shifts=np.array(shifts, dtype=np.int64))
    print("rng_draws ok", shifts[:4])


def joint_and_multi():
    """(iv) joint-mode harness from the reference's own pieces, and a 2-component / per-component
    PSF sequential fit (config 5 shape)."""
    rs = np.random.RandomState(21)
    means, covs, weights = cpu_ref.synthetic_gmm(8, 64, seed=5)
    shape = (64, 72)
    psfs = [asym_psf((11, 11), 1.5 + 0.4 * i, 2.0, dy=0.3 * i) for i in range(3)]
    datasets = {f"obs-{i}": scene(shape, psfs[i], rs, bkg=0.5 + 0.2 * i) for i in range(3)}
    flux_init = rs.gamma(30, size=shape)

    # joint harness with reference objects
    comps = FluxComponents()
    comps["flux"] = SpatialFluxComponent.from_numpy(flux=flux_init, prior=GMMPatchPrior(gmm=ref_gmm(means, covs, weights)))
    total_loss = TotalLoss.from_datasets_and_components(datasets=datasets, components=comps, beta=1.0)
    opt = torch.optim.Adam(comps.parameters(), lr=0.1)
    rows = []
    n_epochs = 12
    for _ in range(n_epochs):
        opt.zero_grad()
        fluxes = comps.to_flux_tuple()
        losses = [
            total_loss.poisson_loss.loss_function(m.evaluate(fluxes=fluxes), c)
            for c, m in total_loss.poisson_loss.iter_by_dataset
        ]
        lp = total_loss.prior_loss.evaluate(fluxes=fluxes)
        total = sum(losses) - 1.0 * sum(lp)
        total.backward()
        opt.step()
        rows.append(
            cpu_ref._trace_row(list(datasets), ["flux"], [v.item() for v in losses], [v.item() for v in lp], 1.0)
        )
    flux_joint = comps["flux"].flux_upsampled.detach().numpy()[0, 0]

    gmm_o = cpu_ref.GMM.from_numpy(means, covs, weights, stride=4)
    final, trace = cpu_ref.map_fit_joint(
        datasets, {"flux": flux_init}, {"flux": cpu_ref.GMMPatchPriorRef(gmm_o)}, n_epochs=n_epochs
    )
    assert np.array_equal(final["flux"], flux_joint), np.abs(final["flux"] - flux_joint).max()
    assert trace[-1]["total"] == rows[-1]["total"]

    out = {
        "joint/flux_init": flux_init,
        "joint/flux_final": flux_joint,
        "gmm/means": means,
        "gmm/covariances": covs,
        "gmm/weights": weights,
    }
    out.update({f"joint/{k}": v for k, v in pack_datasets(datasets).items()})
    out.update({f"joint/{k}": v for k, v in rows_to_arrays(rows).items()})

    # two components, per-component PSFs, GMM + inverse-gamma, sequential, with a mask on "points"
    rs = np.random.RandomState(22)
    shape = (48, 56)
    datasets2 = {}
    for i in range(4):
        d = scene(shape, asym_psf((9, 9), 1.2 + 0.2 * i, 1.6), rs, bkg=0.8)
        d["psf"] = {"extended": d["psf"], "points": asym_psf((7, 7), 0.9, 1.1, dy=-0.2 * i)}
        datasets2[f"o{i}"] = d
    init_ext = rs.gamma(30, size=shape)
    init_pts = rs.gamma(2, size=shape) * 0.2
    comps = FluxComponents()
    comps["extended"] = SpatialFluxComponent.from_numpy(
        flux=init_ext, prior=GMMPatchPrior(gmm=ref_gmm(means, covs, weights))
    )
    comps["points"] = SpatialFluxComponent.from_numpy(flux=init_pts, prior=InverseGammaPrior(alpha=10, beta=1.5))
    res = MAPDeconvolver(n_epochs=6, beta=0.7, display_progress=False).run(datasets=datasets2, components=comps)
    fl = res.components.to_numpy()
    final, trace = cpu_ref.map_fit_sequential(
        datasets2,
        {"extended": init_ext, "points": init_pts},
        {"extended": cpu_ref.GMMPatchPriorRef(gmm_o), "points": cpu_ref.InverseGammaPriorRef(10, 1.5)},
        n_epochs=6,
        beta=0.7,
    )
    assert np.array_equal(final["extended"], fl["extended"]) and np.array_equal(final["points"], fl["points"])
    assert trace[-1]["total"] == res.trace_loss[-1]["total"]
    out["multi/init/extended"] = init_ext
    out["multi/init/points"] = init_pts
    out["multi/final/extended"] = fl["extended"]
    out["multi/final/points"] = fl["points"]
    out.update({f"multi/{k}": v for k, v in pack_datasets(datasets2).items()})
    out.update({f"multi/{k}": v for k, v in trace_to_arrays(res.trace_loss).items()})
    np.savez_compressed(OUT / "joint_multi.npz", **out)
    print("joint_and_multi ok")


if __name__ == "__main__":
    torch.manual_seed(0)
    if len(sys.argv) > 1 and sys.argv[1] == "upsampling":  # add this fixture without touching the others
        upsampling_case()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "upsampling_mixed_psf":
        upsampling_mixed_psf_case()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "calibration":
        calibration_case()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "compute_error":
        compute_error_case()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "linear_flux":
        linear_flux_case()
        sys.exit(0)
    rng_draws()
    anchor_a()
    anchor_b()
    reference_test_cases()
    stage_vectors()
    joint_and_multi()
    upsampling_case()
    upsampling_mixed_psf_case()
    calibration_case()
    linear_flux_case()
    compute_error_case()
    import os

    for f in sorted(OUT.glob("*.npz")):
        print(f.name, os.path.getsize(f) // 1024, "KiB")
