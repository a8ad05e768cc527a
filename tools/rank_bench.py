"""Separable (rank 1..3) against the direct kernel on one 2048^2 convolution: which should `auto` take at rank 2 and 3?
    python tools/rank_bench.py [edge=2048] [k=17]"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from jolideco_amd.ops import ConvPlan, psf_separable_rank

edge = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
k = int(sys.argv[2]) if len(sys.argv) > 2 else 17
dev = "cuda:0"
x = np.arange(k) - (k - 1) / 2
g = lambda s, o=0.0: np.exp(-0.5 * ((x - o) / s) ** 2)
psfs = {1: np.outer(g(2.0), g(2.0)), 2: np.outer(g(1.5), g(1.5)) + 0.3 * np.outer(g(4.0), g(4.0)),
        3: np.outer(g(1.5), g(1.5)) + 0.3 * np.outer(g(4.0), g(3.0, 1.0)) + 0.1 * np.outer(g(2.5, -1.0), g(5.0))}
img = torch.rand(edge, edge, device=dev) + 0.5
sc = torch.rand(edge, edge, device=dev) + 0.5
out = torch.zeros_like(img)
for rank, psf in psfs.items():
    psf = (psf / psf.sum()).astype(np.float32)
    print("rank", rank, "-> psf_separable_rank", psf_separable_rank(psf))
    for m in ("separable", "direct"):
        plan = ConvPlan(edge, edge, k, k, dev, method=m)
        khat = plan.psf_spectrum(torch.from_numpy(psf).to(dev))
        for name, fn in (("fwd", lambda: plan.conv_same(img, sc, khat)),
                         ("adj", lambda: plan.conv_same_adjoint(img, sc, khat, grad_image=out, accumulate=True))):
            for _ in range(5): fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50): fn()
            e1.record(); torch.cuda.synchronize()
            print(f"  {m:10s} {name}: {e0.elapsed_time(e1) / 50 * 1e3:7.1f} us")
        plan.close()
