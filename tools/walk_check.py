"""Strip-walk separable convolution kernels (csrc/walkconv.hip) against the tile kernel (csrc/sepconv.hip) and float64:
correctness over shapes / PSF sizes, then timings at the benchmark sizes.  Run on a GPU box:
    python tools/walk_check.py [check] [time]
"""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from jolideco_amd import _hip  # noqa: E402
from jolideco_amd.data import gaussian_kernel  # noqa: E402
from jolideco_amd.ops import ConvPlan, stirling_mean  # noqa: E402

DEV = torch.device("cuda:0")


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32), device=DEV)


def scene(H, W, n, kshape, seed):
    rs = np.random.RandomState(seed)
    flux = rs.gamma(5.0, size=(H, W)).astype(np.float32)
    data = []
    for i in range(n):
        psf = gaussian_kernel(1.2 + 0.3 * i, kshape).astype(np.float32)
        exposure = ((1.0 + 0.1 * i) * (1.0 + 0.4 * np.linspace(-1, 1, H)[:, None] * np.ones((H, W)))).astype(np.float32)
        bkg = np.full((H, W), 0.5 + 0.1 * i, dtype=np.float32)
        counts = rs.poisson(5.0, size=(H, W)).astype(np.float32)
        data.append((psf, exposure, bkg, counts))
    return flux, data


def run_step(plan, flux, khats, data_dev, stirlings, batch):
    """loss per dataset, gradient of the joint step (batched or per-dataset calls)"""
    n = len(khats)
    losses = [torch.zeros(1, device=DEV) for _ in range(n)]
    grad = torch.zeros_like(flux)
    if batch:
        plan.npred_poisson_batch_fwd_bwd(flux, [d[1] for d in data_dev], khats, [d[2] for d in data_dev],
                                         [d[3] for d in data_dev], stirlings, losses, grad=grad, accumulate=False)
    else:
        for i in range(n):
            plan.npred_poisson_fwd_bwd([flux], [data_dev[i][1]], [khats[i]], data_dev[i][2], data_dev[i][3], stirlings[i],
                                       losses[i], grads=[grad], accumulate=i > 0)
    torch.cuda.synchronize()
    return np.array([float(v) for v in losses]), grad.cpu().numpy()


def check():
    from scipy.signal import fftconvolve

    bad = 0
    cases = [((64, 128), (17, 17), 3), ((100, 260), (17, 17), 2), ((37, 64), (9, 13), 2), ((200, 512), (16, 17), 5),
             ((130, 300), (17, 8), 9), ((96, 256), (5, 5), 1), ((257, 132), (17, 17), 4), ((64, 1028), (11, 11), 8)]
    for (H, W), kshape, n in cases:
        flux, data = scene(H, W, n, kshape, seed=H + W)
        plan = ConvPlan(H, W, kshape[0], kshape[1], DEV, method="separable")
        fdev = dev(flux)
        data_dev = [tuple(dev(x) for x in d) for d in data]
        khats = [plan.psf_spectrum(d[0]) for d in data_dev]
        stirlings = [stirling_mean(d[3]) for d in data]
        # plain convolution and adjoint against float64
        for i in (0, n - 1):
            ref = fftconvolve(flux.astype(np.float64) * data[i][1], data[i][0].astype(np.float64), mode="full")
            oy, ox = (kshape[0] - 1) // 2, (kshape[1] - 1) // 2
            ref = ref[oy:oy + H, ox:ox + W]
            out = {}
            for walk in (0, 1):
                with _hip.options(JD_SEP_WALK=walk):
                    out[walk] = plan.conv_same(fdev, data_dev[i][1], khats[i]).cpu().numpy()
            e0, e1 = (np.abs(out[w] - ref).max() / np.abs(ref).max() for w in (0, 1))
            g = dev(np.random.RandomState(1).normal(size=(H, W)))
            adj = {}
            for walk in (0, 1):
                with _hip.options(JD_SEP_WALK=walk):
                    adj[walk] = plan.conv_same_adjoint(g, data_dev[i][1], khats[i]).cpu().numpy()
            ea = np.abs(adj[1] - adj[0]).max() / np.abs(adj[0]).max()
            ok = e0 < 2e-6 and e1 < 2e-6 and ea < 2e-6
            bad += not ok
            print(f"{H}x{W} psf {kshape} obs {i}: conv vs f64 tile {e0:.1e} walk {e1:.1e}; adjoint walk vs tile {ea:.1e} {'ok' if ok else 'BAD'}")
        res = {}
        for walk in (0, 1):
            for batch in (True, False):
                with _hip.options(JD_SEP_WALK=walk, JD_SEP_JOINT=0):
                    res[walk, batch] = run_step(plan, fdev, khats, data_dev, stirlings, batch)
        # the fused likelihood step (forward + Poisson + adjoint + dataset sum in one launch): same gradient bits as
        # the walk kernels' two launches, batched and per dataset; losses equal to rounding
        for batch in (True, False):
            with _hip.options(JD_SEP_WALK=1, JD_SEP_JOINT=1):
                fused = run_step(plan, fdev, khats, data_dev, stirlings, batch)
            okf = np.array_equal(fused[1], res[1, True][1]) and np.allclose(fused[0], res[1, True][0], rtol=1e-6, atol=0)
            bad += not okf
            print(f"{H}x{W} psf {kshape} n={n}: fused step ({'batch' if batch else 'loop'}) == walk kernels: gradient "
                  f"{np.array_equal(fused[1], res[1, True][1])}, max |grad diff| {np.abs(fused[1] - res[1, True][1]).max():.1e}, "
                  f"loss rel {np.abs(fused[0] / res[1, True][0] - 1).max():.1e} {'ok' if okf else 'BAD'}")
        same = np.array_equal(res[1, True][1], res[1, False][1])
        same_old = np.array_equal(res[0, True][1], res[0, False][1])
        eg = np.abs(res[1, True][1] - res[0, True][1]).max() / np.abs(res[0, True][1]).max()
        el = np.abs(res[1, True][0] - res[0, True][0]).max() / np.abs(res[0, True][0]).max()
        el2 = np.abs(res[1, True][0] - res[1, False][0]).max() / np.abs(res[0, True][0]).max()
        ok = same and same_old and eg < 3e-6 and el < 2e-6 and el2 < 1e-6
        bad += not ok
        print(f"{H}x{W} psf {kshape} n={n}: walk batch == walk loop {same} (tile: {same_old}); walk vs tile grad {eg:.1e} "
              f"loss {el:.1e}; loss batch vs loop {el2:.1e} {'ok' if ok else 'BAD'}")
        plan.close()
    print("MISMATCHES:", bad)
    return bad


def timed(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3  # us


FLUSH = None


def flush_caches():
    """Stream through 1.5 GB that nobody else uses: what the prior's kernels do to the Infinity Cache between two
    likelihood passes of a fit (the timers bracket the library's launches only)."""
    global FLUSH
    if FLUSH is None:
        FLUSH = torch.zeros(3 * (1 << 27), dtype=torch.float32, device=DEV)
    FLUSH.add_(1.0)


def bench(only=None, rows_list=(38, 56, 74, 128), cold=False, joint_rows=()):
    for name, (H, W, n) in {"c3": (2048, 2048, 8), "c4": (4096, 4096, 1), "c2": (1024, 1024, 1), "seq": (2048, 2048, 1),
                            "r4": (2048, 2048, 4), "r2": (2048, 2048, 2), "r16": (2048, 2048, 16), "s16": (1024, 1024, 16)}.items():
        if only and name not in only:
            continue
        flux, data = scene(H, W, n, (17, 17), seed=1)
        plan = ConvPlan(H, W, 17, 17, DEV, method="separable")
        fdev = dev(flux)
        data_dev = [tuple(dev(x) for x in d) for d in data]
        khats = [plan.psf_spectrum(d[0]) for d in data_dev]
        stirlings = [stirling_mean(d[3]) for d in data]
        losses = [torch.zeros(1, device=DEV) for _ in range(n)]
        grad = torch.zeros_like(fdev)
        _hip.profile_enable(1 << 14)

        def step():
            if cold:
                flush_caches()
            if n > 1:
                plan.npred_poisson_batch_fwd_bwd(fdev, [d[1] for d in data_dev], khats, [d[2] for d in data_dev],
                                                 [d[3] for d in data_dev], stirlings, losses, grad=grad, accumulate=False)
            else:
                plan.npred_poisson_fwd_bwd([fdev], [data_dev[0][1]], [khats[0]], data_dev[0][2], data_dev[0][3], stirlings[0],
                                           losses[0], grads=[grad], accumulate=False)

        variants = [dict(JD_SEP_WALK=0, JD_SEP_JOINT=0), dict(JD_SEP_WALK=1, JD_SEP_JOINT=0)]
        if n > 8:
            variants.append(dict(JD_SEP_WALK=1, JD_SEP_JOINT=0, JD_SEP_WALK_ADJ_ALL=0))
        for rows in joint_rows:
            variants.append(dict(JD_SEP_WALK=1, JD_SEP_JOINT=1, JD_SEP_JOINT_ROWS=rows))
        for cols in (() if joint_rows else (2, 4)):
            for rows in rows_list:
                variants.append(dict(JD_SEP_WALK=1, JD_SEP_JOINT=0, JD_SEP_WALK_COLS=cols, JD_SEP_WALK_ROWS=rows,
                                     JD_SEP_WALK_ADJ_COLS=cols, JD_SEP_WALK_ADJ_ROWS=rows))
        for v in variants:
            with _hip.options(**v):
                for _ in range(3):
                    step()
                torch.cuda.synchronize()
                _hip.profile_enable(1 << 14)
                t = timed(step, n=20, warm=0)
                prof = _hip.profile_read()
            fw = prof["poisson_fused"][0] / max(prof["poisson_fused"][1], 1) * 1e3
            ad = prof["sep_conv"][0] / max(prof["sep_conv"][1], 1) * 1e3 or float("nan")  # (fused step: no adjoint launch)
            fb, ab = 20 * H * W * n, (8 * n + 8) * H * W
            print(f"{name}{' cold' if cold else ''} {v}: step {t:.1f} us; fwd+poisson {fw:.1f} us = {fb / fw / 1e6:.2f} TB/s ({fb / fw / 8e6 * 100:.0f} %); "
                  f"adjoint {ad:.1f} us = {ab / ad / 1e6:.2f} TB/s ({ab / ad / 8e6 * 100:.0f} %)", flush=True)
        plan.close()


if __name__ == "__main__":
    what = sys.argv[1:] or ["check", "time"]
    rc = 0
    if "check" in what:
        rc = check()
    for w in what:
        if w.startswith("time"):
            parts = w.split(":")
            bench(only=parts[1].split(",") if len(parts) > 1 and parts[1] else None,
                  rows_list=tuple(int(r) for r in parts[2].split(",") if r) if len(parts) > 2 else (38, 56, 74, 128),
                  cold=w.startswith("timecold"),
                  joint_rows=tuple(int(r) for r in parts[3].split(",")) if len(parts) > 3 else ())
    sys.exit(1 if rc else 0)
