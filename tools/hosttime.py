"""Host enqueue time against total time per step: is a configuration bound by the Python / ctypes launch path?
GPU box: python tools/hosttime.py [config ...]   (c3 also as one rank of eight: its share of a strong-scaling run)"""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402
from jolideco_amd.distributed import DistContext  # noqa: E402

for cfg in sys.argv[1:] or ["c3"]:
    for shard in (1, 8) if cfg == "c3" else (1,):
        fake = DistContext(rank=0, world_size=shard, dry_run=True) if shard > 1 else None
        s = bench.build_session(cfg, torch.device("cuda:0"), dist=fake)
        for _ in range(5):
            s.epoch()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(100):
            s.epoch()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"{cfg} shard-of {shard}: host enqueue {1e3 * (t1 - t0) / 100:.3f} ms/step, total {1e3 * (t2 - t0) / 100:.3f} ms/step", flush=True)
