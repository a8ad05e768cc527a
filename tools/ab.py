"""A/B timing of the bench step under different environment switches IN ONE PROCESS (box-to-box clock differences
are +-10 %, so variants must be compared inside one gpurun call, interleaved):
    python tools/ab.py [config=c3] [rounds=5] [steps=30] -- NAME1:VAR=VAL,VAR2=VAL NAME2: ...
Every variant is a set of library switches (JD_*: jd_set_option) and / or environment variables read by the Python
side at launch time (JOLIDECO_*)."""
import os, sys, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import bench
from jolideco_amd import _hip

args = sys.argv[1:]
split = args.index("--") if "--" in args else len(args)
pos, variants = args[:split], args[split + 1:]
cfg = pos[0] if len(pos) > 0 else "c3"
rounds = int(pos[1]) if len(pos) > 1 else 5
steps = int(pos[2]) if len(pos) > 2 else 30
if not variants:
    variants = ["base:"]
parsed = []
for v in variants:
    name, _, rest = v.partition(":")
    env = dict(kv.split("=", 1) for kv in rest.split(",") if kv)
    parsed.append((name, env))
all_keys = sorted({k for _, e in parsed for k in e})
dev = torch.device("cuda:0")
if cfg == "c6":  # calibrations + up-sampling x2 + general 65x65 PSFs (bench.build_session_c6)
    session = bench.build_session_c6(dev)
elif cfg.endswith("fft"):  # e.g. c3fft: the configuration through the FFT path
    os.environ["JOLIDECO_CONV_METHOD"] = "fft"
    session = bench.build_session(cfg[:-3], dev)
else:
    session = bench.build_session(cfg, dev)
for _ in range(10):
    session.epoch()
torch.cuda.synchronize()
res = {name: {"ms": [], "k": {}} for name, _ in parsed}
for r in range(rounds):
    for name, env in parsed:
        for k in all_keys:
            if k.startswith("JD_"):
                _hip.set_option(k, env.get(k))
            else:
                os.environ.pop(k, None)
        os.environ.update({k: v for k, v in env.items() if not k.startswith("JD_")})
        for _ in range(3):
            session.epoch()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            session.epoch()
        e1.record()
        torch.cuda.synchronize()
        res[name]["ms"].append(e0.elapsed_time(e1) / steps)
        _hip.profile_enable(capacity=4096)
        for _ in range(5):
            session.epoch()
        prof = _hip.profile_read()
        for k, (t, c) in prof.items():
            if c:
                res[name]["k"].setdefault(k, []).append(t / c * 1e3)
for name, _ in parsed:
    ms = np.array(res[name]["ms"])
    ks = {k: round(float(np.median(v)), 1) for k, v in res[name]["k"].items()}
    print(f"{name:24s} step {np.median(ms) * 1e3:7.1f} us (min {ms.min() * 1e3:.1f} max {ms.max() * 1e3:.1f})  kernels us/launch {ks}")
