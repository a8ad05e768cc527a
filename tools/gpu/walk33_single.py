"""Single-dataset likelihood step (forward model + Poisson pass + adjoint), tile kernel against strip-walk kernels, for a
17-tap and a 33-tap Gaussian PSF at 2048^2 and 4096^2: where should the 33-tap frame take single-dataset launches?"""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from jolideco_amd import _hip  # noqa: E402
from jolideco_amd.data import gaussian_kernel  # noqa: E402
from jolideco_amd.ops import ConvPlan, stirling_mean  # noqa: E402

DEV = "cuda:0"
for size in (2048, 4096):
    shape = (size, size)
    rs = np.random.RandomState(0)
    flux = torch.from_numpy(rs.gamma(5.0, size=shape).astype(np.float32)).to(DEV)
    expo = torch.from_numpy(rs.uniform(0.5, 1.5, size=shape).astype(np.float32)).to(DEV)
    bkg = torch.full(shape, 0.7, device=DEV)
    counts_np = rs.poisson(5.0, size=shape).astype(np.float32)
    counts = torch.from_numpy(counts_np).to(DEV)
    st = stirling_mean(counts_np)
    for k, sigma in ((17, 2.0), (33, 3.2)):
        plan = ConvPlan(size, size, k, k, DEV, method="separable")
        khat = plan.psf_spectrum(torch.from_numpy(gaussian_kernel(sigma, (k, k)).astype(np.float32)).to(DEV))
        loss, grad = torch.zeros(1, device=DEV), torch.zeros(shape, device=DEV)
        for walk in (0, 1):
            _hip.set_option("JD_SEP_WALK", walk)
            for _ in range(5):
                plan.npred_poisson_fwd_bwd([flux], [expo], [khat], bkg, counts, st, loss, grads=[grad])
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                plan.npred_poisson_fwd_bwd([flux], [expo], [khat], bkg, counts, st, loss, grads=[grad])
            e1.record()
            torch.cuda.synchronize()
            print(f"{size}^2 psf {k}x{k} {'walk' if walk else 'tile'}: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us per step", flush=True)
        _hip.set_option("JD_SEP_WALK", None)
        plan.close()
