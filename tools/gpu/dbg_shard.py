import sys, numpy as np, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from jolideco_amd import GMMPatchPrior, MAPDeconvolver, SpatialFluxComponent
from jolideco_amd.data import synthetic_gmm, synthetic_observations
from jolideco_amd.distributed import DistContext
from jolideco_amd.priors.patches import GaussianMixtureModel, GaussianMixtureModelMeta

shape, n_obs = (328, 512), 8
datasets, _, flux_init = synthetic_observations(shape=shape, n_obs=n_obs, seed=0)
means, covs, weights = synthetic_gmm(32, 64, seed=0)

def fit(dist, n_epochs):
    gmm = GaussianMixtureModel.from_numpy(means, covs, weights, meta=GaussianMixtureModelMeta(stride=4))
    comp = SpatialFluxComponent.from_numpy(flux=flux_init, prior=GMMPatchPrior(gmm=gmm))
    deco = MAPDeconvolver(n_epochs=n_epochs, display_progress=False, device="cuda:0", fit_mode="joint")
    s = deco.session(datasets, components=comp, dist=dist)
    out = []
    for _ in range(n_epochs):
        s.epoch()
        torch.cuda.synchronize()
        st = s.states[0]
        out.append(dict(flux=st.flux_cur.cpu().numpy().copy(), grad=st.grad.cpu().numpy().copy(), theta=st.theta.cpu().numpy().copy(),
                        scal=s.scalars.cpu().numpy().copy(), shifts=s.priors[0].last_shifts))
    return out

a = fit(DistContext(rank=0, world_size=1, force_collectives=True, dry_run=True), 3)
b = fit(DistContext(), 3)
for e in range(3):
    for k in ("grad", "theta", "flux", "scal"):
        x, y = a[e][k], b[e][k]
        d = np.abs(x.astype(np.float64) - y)
        print(e, k, "equal" if np.array_equal(x, y) else f"DIFF n={np.count_nonzero(d)} max={d.max():.3e} where={np.argwhere(d.reshape(x.shape) > 0)[:5].tolist()}", a[e]["shifts"], b[e]["shifts"])
