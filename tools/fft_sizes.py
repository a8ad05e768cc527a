"""Tuning tool: time the rocFFT convolution (R2C + k-space multiply + C2R + pad/crop) for candidate padded sizes."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, "/root/repo")
edge, k = int(sys.argv[1]), int(sys.argv[2])
cands = [int(v) for v in sys.argv[3:]]
from jolideco_amd.ops import ConvPlan
img = torch.rand((edge, edge), device="cuda")
psf = torch.rand((k, k), device="cuda"); psf /= psf.sum()
for c in cands:
    os.environ["JD_FFT_FORCE_PAD"] = f"{c}:{c}"
    t0 = time.perf_counter()
    plan = ConvPlan(edge, edge, k, k, "cuda", method="fft")
    khat = plan.psf_spectrum(psf)
    torch.cuda.synchronize(); t_plan = time.perf_counter() - t0
    for _ in range(3): plan.conv_same(img, None, khat)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): plan.conv_same(img, None, khat)
    e1.record(); torch.cuda.synchronize()
    print(f"edge {edge} psf {k} pad {plan.Hp}x{plan.Wp}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per conv (plan {t_plan:.2f} s)")
    plan.close()
