"""Every rank's share of an N-rank joint step, timed ONE RANK AFTER THE OTHER in this process (no multi-GPU node needed):
`DistContext(rank, N, dry_run=True)` builds exactly what rank `rank` of N builds -- its datasets (cost-aware placement,
jolideco_amd/distributed.py), its band of the prior's patch rows, the band sum + optimizer step -- and skips only the
transport.  The step time of the real job is the MAX over the ranks (+ what the collectives do not hide).

    python tools/shard_table.py [config=c3] [steps=60] [placement=cost|round-robin] [N ...=2 4 8]

Prints one line per (N, rank) and a summary line per N: max, min, max / min, and the estimated loads of the placement."""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

import bench
from jolideco_amd.distributed import DistContext

cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
placement = sys.argv[3] if len(sys.argv) > 3 else "cost"
worlds = [int(v) for v in sys.argv[4:]] or [2, 4, 8]
os.environ["JOLIDECO_DIST_PLACEMENT"] = placement
dev = torch.device("cuda:0")
table = {"config": cfg, "placement": placement, "steps": steps, "worlds": {}}
for world in worlds:
    rows = []
    for rank in range(world):
        session = bench.build_session(cfg, dev, dist=DistContext(rank=rank, world_size=world, dry_run=True))
        for _ in range(10):
            session.epoch()
        torch.cuda.synchronize()
        times = []
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(steps):
                session.epoch()
            e1.record()
            torch.cuda.synchronize()
            times.append(e0.elapsed_time(e1) / steps)
        n_rows = [item["rows"][1] - item["rows"][0] for item in (session.band_plan or [])]
        row = {"rank": rank, "ms_per_step": float(np.median(times)), "datasets": [g for g, _ in session.local_idx],
               "prior_patch_rows": n_rows, "estimated_load": getattr(session, "rank_loads", [None] * world)[rank]}
        rows.append(row)
        print(f"{cfg} N={world} rank {rank}: {row['ms_per_step'] * 1e3:7.1f} us/step  datasets {row['datasets']}  prior rows {n_rows}", flush=True)
        del session
        torch.cuda.empty_cache()
    ms = [r["ms_per_step"] for r in rows]
    summary = {"max_ms": max(ms), "min_ms": min(ms), "max_over_min": max(ms) / min(ms), "mean_ms": float(np.mean(ms))}
    table["worlds"][str(world)] = {"ranks": rows, **summary}
    print(f"{cfg} N={world} [{placement}]: max {summary['max_ms'] * 1e3:.1f} us, min {summary['min_ms'] * 1e3:.1f} us, max/min "
          f"{summary['max_over_min']:.3f}", flush=True)
print(json.dumps(table))
