"""c5's fit (2048^2, 16 observations, two flux components) with ONE PSF per dataset for both components -- the reference's
default when `psf` is an array (models/npred.py:279-295) -- evaluated through the summed flux (one forward model and one
adjoint per dataset) against the per-component launches (JOLIDECO_MERGE_COMPONENTS=0); interleaved, one process."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
import bench
from jolideco_amd import FluxComponents, GMMPatchPrior, InverseGammaPrior, MAPDeconvolver, SpatialFluxComponent
from jolideco_amd.data import synthetic_gmm, synthetic_observations
from jolideco_amd.priors.patches import GaussianMixtureModel, GaussianMixtureModelMeta

dev = torch.device("cuda:0")
n_obs = int(sys.argv[1]) if len(sys.argv) > 1 else 16
rounds, steps = 5, 30


def build():
    datasets, _, flux_init = synthetic_observations(shape=(2048, 2048), n_obs=n_obs, seed=0)
    means, covs, weights = synthetic_gmm(128, bench.D, seed=0)
    gmm = GaussianMixtureModel.from_numpy(means, covs, weights, meta=GaussianMixtureModelMeta(stride=bench.STRIDE))
    comps = FluxComponents()
    comps["extended"] = SpatialFluxComponent.from_numpy(flux=flux_init, prior=GMMPatchPrior(gmm=gmm))
    comps["points"] = SpatialFluxComponent.from_numpy(flux=0.05 * flux_init, prior=InverseGammaPrior(alpha=10, beta=1.5))
    return MAPDeconvolver(n_epochs=1, display_progress=False, device=dev, fit_mode="joint").session(datasets, components=comps)


sessions = {}
for name, env in (("summed flux", "1"), ("per component", "0")):
    os.environ["JOLIDECO_MERGE_COMPONENTS"] = env
    sessions[name] = build()
    for _ in range(24):  # (past the probe epochs of the graph policy)
        sessions[name].epoch()
    torch.cuda.synchronize()
res = {name: [] for name in sessions}
for r in range(rounds):
    for name, s in sessions.items():
        for _ in range(3):
            s.epoch()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            s.epoch()
        e1.record()
        torch.cuda.synchronize()
        res[name].append(e0.elapsed_time(e1) / steps)
for name in sessions:
    ms = np.array(res[name])
    print(f"c5 shape, {n_obs} observations, shared PSF, {name:14s} step {np.median(ms) * 1e3:7.1f} us (min {ms.min() * 1e3:.1f}) "
          f"it/s {1e3 / np.median(ms):7.1f}  policy: {sessions[name].graph_policy}")
