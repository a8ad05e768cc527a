"""Time the GMM prior forward(+backward) at a given image size:
    python tools/gmm_bench.py [edge=2048] [K=128] [grad=0] [gmm=synthetic|image] [image=noise|truth|mix] [lse=0]
(JD_GMM_SCREEN_DEBUG=1 prints the candidate records and survivors per patch of every call)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from jolideco_amd import _hip
from jolideco_amd.data import image_like_gmm, synthetic_gmm, synthetic_observations
from jolideco_amd.priors.patches import GaussianMixtureModel, GaussianMixtureModelMeta

edge = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
K = int(sys.argv[2]) if len(sys.argv) > 2 else 128
with_grad = len(sys.argv) > 3 and sys.argv[3] == "1"
dev = "cuda:0"
kind = sys.argv[4] if len(sys.argv) > 4 else "synthetic"
image = sys.argv[5] if len(sys.argv) > 5 else "noise"
lse = len(sys.argv) > 6 and sys.argv[6] == "1"  # marginalize=True: logsumexp over the components
means, covs, weights = synthetic_gmm(K, 64, seed=0) if kind == "synthetic" else image_like_gmm(K, 8, seed=0)
gmm = GaussianMixtureModel.from_numpy(means, covs, weights, meta=GaussianMixtureModelMeta(stride=4))
h = gmm.handle(dev)
noise = np.random.RandomState(0).gamma(30, size=(edge, edge))
if image == "noise":
    flux_np = noise
else:
    _, truth, _ = synthetic_observations(shape=(edge, edge), n_obs=1, seed=0)
    flux_np = truth * (1 + 0.02 * np.random.RandomState(1).normal(size=truth.shape)) if image == "truth" else 0.5 * (truth + noise)
flux = torch.from_numpy(flux_np.astype(np.float32)).to(dev)
v = torch.zeros(1, device=dev)
g = torch.zeros_like(flux) if with_grad else None
for _ in range(8):  # (the record buffer of the screened paths grows over the first passes)
    h.prior_fwd_bwd(flux, 4, (1, -1), v, 1.0, grad=g, grad_coef=1.0, marginalize=lse)
    torch.cuda.synchronize()
_hip.profile_enable(capacity=4096)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 10
e0.record()
for _ in range(n):
    h.prior_fwd_bwd(flux, 4, (1, -1), v, 1.0, grad=g, grad_coef=1.0, marginalize=lse)
e1.record(); torch.cuda.synchronize()
prof = _hip.profile_read()
print(f"{edge}^2 K={K} gmm={kind} image={image} lse={int(lse)}: {e0.elapsed_time(e1) / n * 1e3:.1f} us per call;",
      {k: round(t / c * 1e3, 1) for k, (t, c) in prof.items() if c}, "value", float(v), "stats", h.screen_stats())
