"""Does the forward launch of the likelihood run faster when it does not directly follow the optimizer step's stores?
Experiment: the c3 fit with a pause (a synchronisation, or a pure vector-ALU kernel of ~0.2 ms) inserted before the
likelihood launch of every step; the library's own timers bracket the launches only.  GPU box: python tools/cold_probe.py"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

import bench  # noqa: E402
from jolideco_amd import _hip  # noqa: E402

dev = torch.device("cuda:0")
session = bench.build_session(sys.argv[1] if len(sys.argv) > 1 else "c3", dev)
poisson = session.total_loss.poisson_loss
real = poisson.fwd_bwd_batch


def timers(label, before=None):
    def wrapped(*a, **k):
        if before:
            before()
        return real(*a, **k)

    poisson.fwd_bwd_batch = wrapped
    for _ in range(5):
        session.epoch()
    torch.cuda.synchronize()
    _hip.profile_enable(capacity=4096)
    for _ in range(20):
        session.epoch()
    prof = _hip.profile_read()
    print(label, {k: round(t / c * 1e3, 1) for k, (t, c) in prof.items() if c and k in ("poisson_fused", "sep_conv", "gmm_gather", "gmm_stage", "gmm_screen")}, flush=True)


timers("as is               ")
timers("sync before         ", lambda: torch.cuda.synchronize())
timers("0.2 ms ALU kernel   ", lambda: _hip.clock_probe(0.2))
timers("as is               ")
