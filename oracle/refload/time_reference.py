"""Build container only: time the LIVE reference's joint step next to oracle/cpu_ref.py on the same
bounded sample bench.py uses for `cpu_baseline`, to show that the oracle is a fair stand-in for
the reference CPU path (ms/step within noise, same outputs).  Result recorded in DESIGN.md.

Run:  python oracle/refload/time_reference.py [edge] [n_obs] [K]
"""
import sys
import time
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
REPO = HERE.parent.parent
sys.path.insert(0, str(HERE))
sys.path.insert(0, str(REPO))

from load_ref import load_reference  # noqa: E402

load_reference()

from jolideco.loss import TotalLoss  # noqa: E402
from jolideco.models import FluxComponents, SpatialFluxComponent  # noqa: E402
from jolideco.priors import GMMPatchPrior  # noqa: E402
from jolideco.priors.patches.gmm import GaussianMixtureModel, GaussianMixtureModelMeta  # noqa: E402
from jolideco.utils.norms import SubtractMeanPatchNorm  # noqa: E402

from jolideco_amd.data import synthetic_gmm, synthetic_observations  # noqa: E402
from oracle import cpu_ref  # noqa: E402


def main():
    edge = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    n_obs = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    K = int(sys.argv[3]) if len(sys.argv) > 3 else 128
    steps = 3
    datasets, _, flux_init = synthetic_observations(shape=(edge, edge), n_obs=n_obs, seed=0)
    means, covs, weights = synthetic_gmm(K, 64, seed=0)

    # --- reference objects, joint objective assembled from the reference's own pieces ---------
    meta = GaussianMixtureModelMeta(stride=4, patch_norm=SubtractMeanPatchNorm())
    gmm_r = GaussianMixtureModel.from_numpy(means=means, covariances=covs, weights=weights, meta=meta)
    comps = FluxComponents()
    comps["flux"] = SpatialFluxComponent.from_numpy(flux=flux_init, prior=GMMPatchPrior(gmm=gmm_r))
    total_loss = TotalLoss.from_datasets_and_components(datasets=datasets, components=comps, beta=1.0)
    opt = torch.optim.Adam(comps.parameters(), lr=0.1)

    def ref_step():
        opt.zero_grad()
        fluxes = comps.to_flux_tuple()
        losses = [
            total_loss.poisson_loss.loss_function(m.evaluate(fluxes=fluxes), c)
            for c, m in total_loss.poisson_loss.iter_by_dataset
        ]
        total = sum(losses) - sum(total_loss.prior_loss.evaluate(fluxes=fluxes))
        total.backward()
        opt.step()

    # --- oracle ---------------------------------------------------------------------------------
    gmm_o = cpu_ref.GMM.from_numpy(means, covs, weights, stride=4)
    prior = cpu_ref.GMMPatchPriorRef(gmm_o)
    theta = cpu_ref.log_flux_parameter(flux_init)
    data = [cpu_ref.DatasetRef.from_numpy(d, ["flux"]) for d in datasets.values()]
    opt_o = torch.optim.Adam([theta], lr=0.1)

    def oracle_step():
        opt_o.zero_grad()
        total, _, _ = cpu_ref.joint_loss(data, (cpu_ref.to_flux(theta),), [prior], 1.0)
        total.backward()
        opt_o.step()

    out = {}
    for name, fn in (("reference", ref_step), ("oracle", oracle_step)):
        fn()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        out[name] = (time.perf_counter() - t0) / steps
    flux_ref = comps["flux"].flux_upsampled.detach().numpy()[0, 0]
    flux_orc = cpu_ref.to_flux(theta).detach().numpy()[0, 0]
    print(f"edge={edge} n_obs={n_obs} K={K} threads={torch.get_num_threads()}")
    print(f"reference: {out['reference'] * 1e3:.1f} ms/step   oracle: {out['oracle'] * 1e3:.1f} ms/step")
    print("identical outputs after", steps + 1, "steps:", bool(np.array_equal(flux_ref, flux_orc)),
          "max |diff| =", float(np.abs(flux_ref - flux_orc).max()))


if __name__ == "__main__":
    main()
