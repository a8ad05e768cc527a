import sys, time, torch
sys.path.insert(0, '/root/repo')
import bench
from jolideco_amd.distributed import DistContext
for shard in (1, 8):
    fake = DistContext(rank=0, world_size=shard, dry_run=True) if shard > 1 else None
    s = bench.build_session("c3", torch.device("cuda:0"), dist=fake)
    for _ in range(3): s.epoch()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50): s.epoch()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"shard-of {shard}: host enqueue {1e3*(t1-t0)/50:.3f} ms/step, total {1e3*(t2-t0)/50:.3f} ms/step")
