"""Randomised comparison of the screened (fused backward) GMM arg-max prior against the dense fp32 kernel on the GPU:
random image shapes, strides, shifts, component counts, patch-row shards, filtered patches.  Every case must agree
bit for bit in gradient and arg-max and to 2e-7 in the value (different partial-sum partitions).
GPU box: `python tools/fuzz_gmm.py [n_cases] [seed]`."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from jolideco_amd import _hip  # noqa: E402
from jolideco_amd.data import synthetic_gmm  # noqa: E402
from jolideco_amd.priors.patches import GaussianMixtureModel, GaussianMixtureModelMeta  # noqa: E402

DEV = "cuda:0"
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)


def run(handle, flux, stride, shifts, rows, screen, n_patches):
    _hip.set_option("JD_GMM_SCREEN", 1 if screen else 0)
    value, grad = torch.zeros(1, device=DEV), torch.zeros_like(flux)
    argmax = torch.full((n_patches,), -7, dtype=torch.int32, device=DEV)
    handle.prior_fwd_bwd(flux, stride, shifts, value, 0.5, grad=grad, grad_coef=1.5, patch_rows=rows, argmax_out=argmax)
    torch.cuda.synchronize()
    return float(value), grad.cpu().numpy(), argmax.cpu().numpy()


bad = 0
for case in range(n_cases):
    H, W = int(rs.randint(8, 220)), int(rs.randint(8, 260))
    stride = int(rs.choice([1, 2, 3, 4, 4, 4, 5, 8]))
    K = int(rs.choice([1, 2, 3, 7, 16, 33, 64]))
    zero_means = bool(rs.rand() < 0.7)
    means, covs, weights = synthetic_gmm(K, 64, seed=int(rs.randint(1 << 30)))
    if not zero_means:  # mixtures with means: wider screening bounds, the exact stage subtracts them
        means = rs.normal(scale=0.3, size=means.shape)
    if rs.rand() < 0.3:
        covs = covs * np.logspace(-4, 3, K)[:, None, None]
    # pixel weights of the model: stride 4 or none (the patch stride of the prior is independent of it)
    gmm = GaussianMixtureModel.from_numpy(means, covs, weights, meta=GaussianMixtureModelMeta(stride=4 if rs.rand() < 0.7 else None))
    image = rs.gamma(rs.choice([0.5, 5.0, 50.0]), size=(H, W)).astype(np.float32) * float(rs.choice([1e-3, 1.0, 1e3]))
    if rs.rand() < 0.4:
        y, x = rs.randint(0, H), rs.randint(0, W)
        image[y : y + rs.randint(1, 6), x : x + rs.randint(1, 9)] = -3e5
    flux = torch.from_numpy(image).to(DEV)
    n_py, n_px = (H - 8) // stride + 1, (W - 8) // stride + 1
    rows = (0, -1)
    if rs.rand() < 0.4 and n_py > 2:
        lo = int(rs.randint(0, n_py - 1))
        rows = (lo, int(rs.randint(lo + 1, n_py + 1)))
    shifts = (int(rs.randint(-9, 10)), int(rs.randint(-9, 10)))
    handle = gmm.handle(DEV)
    a = run(handle, flux, stride, shifts, rows, True, n_py * n_px)
    b = run(handle, flux, stride, shifts, rows, False, n_py * n_px)
    ok = np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and (
        a[0] == b[0] or abs(a[0] - b[0]) <= 2e-7 * abs(b[0]) or (np.isnan(a[0]) and np.isnan(b[0]))
    )
    if not ok:
        bad += 1
        print(f"MISMATCH case {case}: H={H} W={W} stride={stride} K={K} zero_means={zero_means} rows={rows} shifts={shifts} "
              f"value {a[0]} vs {b[0]} grad_equal={np.array_equal(a[1], b[1])} argmax_equal={np.array_equal(a[2], b[2])}")
print(f"{n_cases} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
