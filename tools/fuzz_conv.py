"""Randomised check of the three convolution methods (separable, direct MFMA, rocFFT) against a float64 'same'
convolution (the reference's convolve_fft_torch semantics: linear convolution of the (H+kh-1, W+kw-1) grid, centre crop,
utils/torch.py:337-370) and of their adjoints by the dot-product identity <A x, y> = <x, A^T y>.  Random image shapes
(smaller than a tile, odd widths), PSF shapes (odd, even, non-square, up to 33), with and without the exposure scale.
GPU box: `python tools/fuzz_conv.py [n_cases] [seed]`.  Round 1: 150 cases, worst relative L-inf direct 2.2e-6 (32 x 28 taps),
fft 5.1e-7, separable 4.0e-7; every adjoint passes the dot-product test."""
import os
import sys

import numpy as np
import torch
from scipy.signal import convolve2d

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from jolideco_amd.data import gaussian_kernel  # noqa: E402
from jolideco_amd.ops import ConvPlan, psf_separable_rank  # noqa: E402

DEV = torch.device("cuda:0")
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)


def same64(image, psf):
    """Centre crop of the full linear convolution, as `_centered` of the reference does it."""
    full = convolve2d(image.astype(np.float64), psf.astype(np.float64), mode="full")
    kh, kw = psf.shape
    y0, x0 = (kh - 1) // 2, (kw - 1) // 2
    return full[y0 : y0 + image.shape[0], x0 : x0 + image.shape[1]]


bad = 0
worst = {}
for case in range(n_cases):
    H, W = int(rs.randint(9, 200)), int(rs.randint(9, 260))
    kh, kw = int(rs.randint(1, 34)), int(rs.randint(1, 34))
    psf = gaussian_kernel(rs.uniform(0.7, 4.0), (kh, kw))
    separable = rs.rand() < 0.6
    if not separable:
        psf = psf * (1.0 + 0.5 * rs.rand(kh, kw))
    psf = (psf / psf.sum()).astype(np.float32)
    image = rs.gamma(2.0, size=(H, W)).astype(np.float32)
    scale = (1.0 + rs.rand(H, W)).astype(np.float32) if rs.rand() < 0.5 else None
    y = rs.normal(size=(H, W)).astype(np.float32)
    ref = same64(image * (scale if scale is not None else 1.0), psf)
    methods = ["direct", "fft"] + (["separable"] if psf_separable_rank(psf) > 0 else [])
    for method in methods:
        try:
            plan = ConvPlan(H, W, kh, kw, DEV, method=method)
        except RuntimeError as error:  # a method may refuse a geometry (e.g. direct beyond 33x33): not a failure
            print(f"case {case}: {method} refuses {H}x{W} psf {kh}x{kw}: {error}")
            continue
        khat = plan.psf_spectrum(torch.from_numpy(psf).to(DEV))
        img_t = torch.from_numpy(image).to(DEV)
        sc_t = None if scale is None else torch.from_numpy(scale).to(DEV)
        out = plan.conv_same(img_t, sc_t, khat).cpu().numpy().astype(np.float64)
        err = np.abs(out - ref).max() / np.abs(ref).max()
        y_t = torch.from_numpy(y).to(DEV)
        adj = plan.conv_same_adjoint(y_t, sc_t, khat).cpu().numpy().astype(np.float64)
        lhs, rhs = float((out * y).sum()), float((image.astype(np.float64) * adj).sum())
        dot = abs(lhs - rhs) / (np.abs(out * y).sum() + 1e-30)
        tol = 2e-5 if method == "fft" else 5e-6  # fp32 accumulation over up to 33 x 33 taps
        worst[method] = max(worst.get(method, 0.0), err)
        if not (err < tol and dot < 5e-6):
            bad += 1
            print(f"MISMATCH case {case}: {method} {H}x{W} psf {kh}x{kw} scale={scale is not None} rel err {err:.2e} dot {dot:.2e}")
        plan.close()
print(f"{n_cases} cases, {bad} mismatches, worst rel Linf per method: " + ", ".join(f"{m} {v:.1e}" for m, v in worst.items()))
sys.exit(1 if bad else 0)
