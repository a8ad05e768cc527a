"""What a sharded joint step pays for its two collectives, measured on the hardware there is -- ONE GPU, a one-rank RCCL
process group (`JOLIDECO_FORCE_COLLECTIVES=1`: the sharded code path with its collectives): the HOST time of the
`torch.distributed` calls FitSession makes per step (asynchronous all-reduce + wait, all-gather of the bands) and the
ON-STREAM time RCCL takes for them with one rank (launch + copy: the floor under any N; the xGMI transfer itself cannot be
measured here).  Prints one JSON line.   python tools/rccl_probe.py [buffer_floats=4194312] [band_floats=139264]"""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ.setdefault("JOLIDECO_FORCE_COLLECTIVES", "1")
os.environ.setdefault("MASTER_PORT", "29577")
import numpy as np, torch
import torch.distributed as dist
from jolideco_amd.distributed import init_from_env

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048 * 2048 + 8
band = int(sys.argv[2]) if len(sys.argv) > 2 else 68 * 2048
ctx = init_from_env("nccl")
dev = torch.device("cuda", 0)
buf = torch.zeros(n, dtype=torch.float32, device=dev)
piece = torch.zeros(band, dtype=torch.float32, device=dev)
out = torch.zeros(band * ctx.world_size, dtype=torch.float32, device=dev)
for _ in range(20):
    h = ctx.all_reduce_sum_async(buf); ctx.all_gather_flat(out, piece); h.wait()
torch.cuda.synchronize()
host_ar, host_ag, host_wait, dev_ar, dev_ag = [], [], [], [], []
for _ in range(200):
    torch.cuda.synchronize()
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    t0 = time.perf_counter()
    h = ctx.all_reduce_sum_async(buf)
    t1 = time.perf_counter()
    e0.record()
    ctx.all_gather_flat(out, piece)
    e1.record()
    t2 = time.perf_counter()
    h.wait()
    e2.record()
    t3 = time.perf_counter()
    torch.cuda.synchronize()
    host_ar.append(t1 - t0), host_ag.append(t2 - t1), host_wait.append(t3 - t2)
    dev_ag.append(e0.elapsed_time(e1)), dev_ar.append(e0.elapsed_time(e2))
# the all-reduce alone, on-stream (blocking form)
alone = []
for _ in range(100):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ctx.all_reduce_sum(buf); e1.record(); torch.cuda.synchronize()
    alone.append(e0.elapsed_time(e1))
med = lambda v: float(np.median(v))
print(json.dumps({
    "backend": dist.get_backend(), "world_size": dist.get_world_size(), "all_reduce_bytes": 4 * n, "all_gather_piece_bytes": 4 * band,
    "host_us": {"all_reduce_async_call": 1e6 * med(host_ar), "all_gather_call": 1e6 * med(host_ag), "wait_call": 1e6 * med(host_wait)},
    "on_stream_us_one_rank": {"all_gather": 1e3 * med(dev_ag), "all_reduce_until_waited": 1e3 * med(dev_ar), "all_reduce_blocking": 1e3 * med(alone)},
    "note": "one rank: RCCL launches its kernels and copies in place -- the floor under the collectives of any N; the xGMI transfer "
            "time of N > 1 is not in these numbers"}))
dist.destroy_process_group()
