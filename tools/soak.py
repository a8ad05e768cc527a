"""Soak run of the headline workload: 3000 joint steps, then device memory before / after, finiteness of the losses and
of the flux (GPU box: `python tools/soak.py`).  Round 1: 0.620 ms/step, no memory growth."""
import sys, time
sys.path.insert(0, ".")
import torch, numpy as np
import bench
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
s = bench.build_session("c3", dev)
for _ in range(20): s.epoch()
torch.cuda.synchronize()
free0, total = torch.cuda.mem_get_info()
t0 = time.perf_counter()
n = 3000
for _ in range(n): s.epoch()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
free1, _ = torch.cuda.mem_get_info()
sc = s.scalars.cpu().numpy()
print("steps", n, "ms/step", 1e3 * dt / n, "free MB before/after", free0 >> 20, free1 >> 20, "finite", bool(np.all(np.isfinite(sc))), "loss sum", float(sc.sum()))
flux = s.states[0].flux_cur.cpu().numpy()
print("flux min/max", flux.min(), flux.max(), "finite", bool(np.isfinite(flux).all()))
