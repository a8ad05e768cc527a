"""Where the prior's phase 1 runs relative to the likelihood launches (interleaved sessions of one config in one process):
    base        side stream beside the likelihood (the round's default form)
    prio        the side stream at high priority
    fence       (removed after the measurement) the main stream waits for the SCREEN launch
    one         one stream
usage: python tools/gpu/r5_h1.py c3 [rounds=5] [steps=40]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
os.environ["JOLIDECO_GRAPH"] = "0"
import numpy as np, torch
import bench

cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
dev = torch.device("cuda:0")
variants = [
    ("base", {"JOLIDECO_PRIOR_OVERLAP": "1"}),
    ("prio", {"JOLIDECO_PRIOR_OVERLAP": "1", "JOLIDECO_PRIOR_STREAM_PRIORITY": "-1"}),
    # ("fence", ...), ("fence+prio", ...): JOLIDECO_PRIOR_FENCE / jd_gmm_set_screen_fence were measured with this script
    # (profiles/r05/ab_prior_schedule.txt: no better than one stream) and removed again
    ("one", {"JOLIDECO_PRIOR_OVERLAP": "0"}),
]
keys = sorted({k for _, e in variants for k in e})
sessions = {}
for name, env in variants:
    for k in keys:
        os.environ.pop(k, None)
    os.environ.update(env)
    sessions[name] = bench.build_session_c6(dev) if cfg == "c6" else bench.build_session(cfg, dev)
    for _ in range(8):
        sessions[name].epoch()
    torch.cuda.synchronize()
res = {name: [] for name, _ in variants}
for r in range(rounds):
    for name, _ in variants:
        s = sessions[name]
        for _ in range(3):
            s.epoch()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            s.epoch()
        e1.record()
        torch.cuda.synchronize()
        res[name].append(e0.elapsed_time(e1) / steps)
# same bits whatever the schedule
ref = [st.flux_cur.clone() for st in sessions["base"].states]
n_ref = sessions["base"].step
for name, _ in variants:
    ms = np.array(res[name])
    s = sessions[name]
    same = s.step == n_ref and all(torch.equal(a, st.flux_cur) for a, st in zip(ref, s.states))
    print(f"{cfg} {name:12s} step {np.median(ms) * 1e3:7.1f} us (min {ms.min() * 1e3:.1f} max {ms.max() * 1e3:.1f}) "
          f"it/s {1.0 / np.median(ms) * 1e3:7.1f}  bits equal to base: {same}")
