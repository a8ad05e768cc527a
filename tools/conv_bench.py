"""Time one 'same' convolution (forward, with exposure) and its adjoint per method on the GPU.
Usage: python tools/conv_bench.py [edge=2048] [k=17] [methods=separable,direct,fft]"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from jolideco_amd.ops import ConvPlan

edge = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
k = int(sys.argv[2]) if len(sys.argv) > 2 else 17
methods = (sys.argv[3] if len(sys.argv) > 3 else "separable,direct,fft").split(",")
dev = "cuda:0"
x = np.arange(k) - (k - 1) / 2
g = np.exp(-0.5 * (x / (k / 8)) ** 2)
psf = np.outer(g, g); psf = (psf / psf.sum()).astype(np.float32)
img = torch.rand(edge, edge, device=dev) + 0.5
sc = torch.rand(edge, edge, device=dev) + 0.5
out = torch.zeros_like(img)
for m in methods:
    try:
        plan = ConvPlan(edge, edge, k, k, dev, method=m)
    except RuntimeError as e:
        print(m, "unsupported:", str(e)[:60]); continue
    khat = plan.psf_spectrum(torch.from_numpy(psf).to(dev))
    for name, fn in (("fwd", lambda: plan.conv_same(img, sc, khat)),
                     ("adj", lambda: plan.conv_same_adjoint(img, sc, khat, grad_image=out, accumulate=True))):
        for _ in range(5): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 50
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        print(f"{m:10s} {name} {edge}^2 k={k}: {e0.elapsed_time(e1) / n * 1e3:8.1f} us")
    plan.close()
