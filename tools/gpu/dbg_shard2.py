import sys, numpy as np, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from jolideco_amd import GMMPatchPrior
from jolideco_amd.data import synthetic_gmm, synthetic_observations
from jolideco_amd.ops import add_rolled_bands, band_rows
from jolideco_amd.priors.patches import GaussianMixtureModel, GaussianMixtureModelMeta

shape = (328, 512)
_, _, flux_init = synthetic_observations(shape=shape, n_obs=1, seed=0)
means, covs, weights = synthetic_gmm(32, 64, seed=0)
gmm = GaussianMixtureModel.from_numpy(means, covs, weights, meta=GaussianMixtureModelMeta(stride=4))
prior = GMMPatchPrior(gmm=gmm)
flux = torch.from_numpy(flux_init.astype(np.float32)).to("cuda:0")
shifts = (-2, 2)
H, W = shape
n_rows = prior.n_patch_rows(shape)
base = torch.rand(shape, device="cuda:0") * 1e-6
for coef in (-1.0, -0.37):
    v = torch.zeros(1, device="cuda:0")
    g1 = base.clone()
    prior.device_fwd_bwd(flux, v, grad=g1, coef=coef, shifts=shifts)
    g1b = base.clone()
    prior.device_fwd_bwd(flux, v, grad=g1b, coef=coef, shifts=shifts, patch_rows=(0, n_rows))
    y0, y1 = band_rows((0, n_rows), 4, H)
    band = torch.zeros((y1 - y0) * W + 4, device="cuda:0")
    prior.device_fwd_bwd(flux, v, coef=coef, shifts=shifts, patch_rows=(0, n_rows), band_out=band)
    g2 = base.clone()
    add_rolled_bands(g2, shifts, band, band.numel(), [(y0, y1)])
    z1 = torch.zeros(shape, device="cuda:0"); prior.device_fwd_bwd(flux, v, grad=z1, coef=coef, shifts=shifts)
    z2 = torch.zeros(shape, device="cuda:0"); add_rolled_bands(z2, shifts, band, band.numel(), [(y0, y1)])
    torch.cuda.synchronize()
    for name, x, y in (("accumulate rows=all vs explicit", g1, g1b), ("accumulate vs band+add (into base)", g1, g2), ("into zeros", z1, z2)):
        d = (x.double() - y.double()).abs()
        print(coef, name, "equal" if torch.equal(x, y) else f"DIFF n={int((d > 0).sum())} max={float(d.max()):.3e} rel={float(d.max() / x.abs().max()):.2e}")
