"""Randomised comparison of the batched joint step (jd_npred_poisson_batch_multi_fwd_bwd) against the per-dataset loop on
the GPU: random image shapes (odd widths too), PSF shapes (odd, even, non-square, up to 33), 2-6 observations, 1-3 flux
components with their own PSFs (Gaussian: rank 1; sum of two Gaussians: rank 2).  Two joint steps; the fluxes of all
components must agree bit for bit.  GPU box: `python tools/fuzz_batch.py [n_cases] [seed] [walk]` (walk: option
JD_SEP_WALK = 1 -- the strip-walk kernels wherever their geometry allows, half the PSFs then at most 17 x 17 and the widths
multiples of 4, with up to 16 observations)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from jolideco_amd import FluxComponents, MAPDeconvolver, SpatialFluxComponent, UniformPrior, _hip  # noqa: E402
from jolideco_amd.data import gaussian_kernel  # noqa: E402

DEV = "cuda:0"
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
walk = len(sys.argv) > 3 and sys.argv[3] == "walk"
if walk:
    _hip.set_option("JD_SEP_WALK", 1)
bad = skipped = 0
for case in range(n_cases):
    H, W = int(rs.randint(20, 150)), int(rs.randint(20, 200))
    kh, kw = int(rs.randint(3, 34)), int(rs.randint(3, 34))
    n_obs, n_comp = int(rs.randint(2, 7)), int(rs.randint(1, 4))
    if walk and case % 2 == 0:
        W, kh, kw, n_comp = (W + 3) // 4 * 4, int(rs.randint(3, 18)), int(rs.randint(3, 18)), int(rs.randint(1, 4))
        n_obs = int(rs.randint(2, 17))
    names = ["a", "b", "c"][:n_comp]

    # round 4: every third case gives each PSF its OWN array size (rank 1 only: that is what shares one plan by embedding,
    # models/npred.py::common_kernel_shape) -- operators of both strip-walk frames, and trimmed windows, in one batch
    mixed_sizes = case % 3 == 1

    def psf():
        if mixed_sizes:
            shape = (int(rs.randint(3, 34)), int(rs.randint(3, 34)))
            k = gaussian_kernel(rs.uniform(0.8, 4.0), shape)
            return (k / k.sum()).astype(np.float32)
        k = gaussian_kernel(rs.uniform(0.8, 3.0), (kh, kw))
        if rs.rand() < 0.4:
            k = 0.7 * k + 0.3 * gaussian_kernel(rs.uniform(3.0, 5.0), (kh, kw))
        return (k / k.sum()).astype(np.float32)

    datasets = {}
    for i in range(n_obs):
        exposure = (1.0 + rs.uniform(0, 1)) * (1.0 + 0.3 * np.linspace(-1, 1, H)[:, None] * np.ones((H, W)))
        datasets[f"obs-{i}"] = {
            "counts": rs.poisson(5.0, size=(H, W)).astype(np.float32),
            "psf": {name: psf() for name in names} if n_comp > 1 or rs.rand() < 0.5 else psf(),
            "exposure": exposure.astype(np.float32),
            "background": np.full((H, W), rs.uniform(0.1, 2.0), dtype=np.float32),
        }
        if not isinstance(datasets[f"obs-{i}"]["psf"], dict) and n_comp == 1:
            pass
    flux_init = rs.gamma(5.0, size=(H, W))
    results = {}
    for mode in ("batch", "loop"):
        if mode == "loop":
            os.environ["JOLIDECO_NO_BATCH"] = "1"
        else:
            os.environ.pop("JOLIDECO_NO_BATCH", None)
        comps = FluxComponents()
        for j, name in enumerate(names):
            comps[name] = SpatialFluxComponent.from_numpy(flux=flux_init / (j + 1.0), prior=UniformPrior())
        if n_comp == 1:
            for d in datasets.values():
                if isinstance(d["psf"], dict):
                    d["psf"] = d["psf"]["a"]
        deco = MAPDeconvolver(n_epochs=2, display_progress=False, device=DEV, fit_mode="joint")
        session = deco.session(datasets, components=comps)
        if mode == "batch" and not session.batch_joint:
            skipped += 1  # e.g. a PSF that is not low-rank enough: no separable plan, nothing to compare
            break
        res = deco.run(datasets, components=comps)
        results[mode] = {name: res.components[name].flux_upsampled_numpy for name in names}
    if len(results) < 2:
        continue
    if not all(np.array_equal(results["batch"][n], results["loop"][n]) for n in names):
        bad += 1
        print(f"MISMATCH case {case}: H={H} W={W} psf={kh}x{kw} n_obs={n_obs} n_comp={n_comp} mixed_sizes={mixed_sizes}")
os.environ.pop("JOLIDECO_NO_BATCH", None)
print(f"{n_cases} cases, {skipped} without a batched path, {bad} mismatches")
sys.exit(1 if bad else 0)
