"""The launch- / host-bound regime (round-4 verdict, weak 7): 8 calibrated, up-sampled observations (c6 shape, uniform and GMM
prior) at 512^2 / 1024^2 / 2048^2 flux pixels, one joint step: eagerly enqueued by value (rounds 1-4), planned (device-resident
step scalars, eager), replayed from a captured hipGraph.  Wall time per step over 200 steps (queue kept full) and the host's
own time per step (first 16 steps into an empty queue)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
from jolideco_amd import GMMPatchPrior, MAPDeconvolver, NPredCalibration, NPredCalibrations, SpatialFluxComponent, UniformPrior
from jolideco_amd.data import instrument_observations, synthetic_gmm
from jolideco_amd.priors.patches import GaussianMixtureModel, GaussianMixtureModelMeta

DEV = "cuda:0"
means, covs, weights = synthetic_gmm(128, 64, seed=0)
ONLY = os.environ.get("SMALL_FITS_ONLY")  # e.g. "256:uniform:by-value" (a profiler run of one case)
for counts_shape, n_obs in (((256, 256), 8), ((512, 512), 8), ((1024, 1024), 8)):
    if ONLY and int(ONLY.split(":")[0]) != counts_shape[0]:
        continue
    datasets, _, flux_init, cal = instrument_observations(shape=counts_shape, n_obs=n_obs, seed=0, psf_shape=(33, 33))
    for prior_name in ("uniform", "gmm"):
        for mode in ("by-value", "planned", "graph", "auto"):
            if ONLY and ONLY.split(":")[1:] != [prior_name, mode]:
                continue
            os.environ["JOLIDECO_STEP_SCALARS"] = "host" if mode == "by-value" else "device"
            os.environ["JOLIDECO_GRAPH"] = "1" if mode == "graph" else "0"
            if mode == "auto":  # the default policy: probe epochs, then captured epochs where the host bounds the fit
                del os.environ["JOLIDECO_GRAPH"]
            gmm = GaussianMixtureModel.from_numpy(means, covs, weights, meta=GaussianMixtureModelMeta(stride=4))
            prior = UniformPrior() if prior_name == "uniform" else GMMPatchPrior(gmm=gmm)
            comp = SpatialFluxComponent.from_numpy(flux=flux_init, upsampling_factor=2, prior=prior)
            cals = NPredCalibrations()
            for name, (sx, sy, norm) in cal.items():
                cals[name] = NPredCalibration(shift_x=sx, shift_y=sy, background_norm=norm)
            session = MAPDeconvolver(n_epochs=1, display_progress=False, device=DEV, fit_mode="joint").session(
                datasets, components=comp, calibrations=cals)
            for _ in range(12):
                session.epoch()
            torch.cuda.synchronize()
            walls, hosts = [], []
            for _ in range(5):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(16):
                    session.epoch()
                t1 = time.perf_counter()
                for _ in range(184):
                    session.epoch()
                torch.cuda.synchronize()
                t2 = time.perf_counter()
                walls.append((t2 - t0) / 200)
                hosts.append((t1 - t0) / 16)
            print(f"flux grid {2 * counts_shape[0]}^2 x {n_obs} calibrated obs, {prior_name:7s} prior, {mode:8s}: "
                  f"{np.median(walls) * 1e6:7.1f} us/step, host {np.median(hosts) * 1e6:6.1f} us/step, graphs {len(session._graphs)}"
                  + (f" [{session.graph_policy}]" if mode == "auto" else ""), flush=True)
            del session
