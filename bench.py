#!/usr/bin/env python3
"""Benchmark of the MAP deconvolution inner loop on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N = 1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json `metric`, configs[2], inputs as SURVEY.md section 8(d) specifies them): 2048x2048 image, 8
synthetic observations with varying PSF (Gaussian, sigma = 1.5 + 0.25 i, on 17x17 arrays and -- sigma >= 3: observations 6
and 7 -- on 33x33 arrays; `config.psf_shapes`) / exposure / background, one flux component with a GMM patch prior (8x8
patches, stride 4, K = 128 components), fp32, JOINT fit: one "step" = one optimizer iteration on
sum_d L_d - beta * logprior = the forward models of all observations (PSF convolution, clip, + background) with
the fused Poisson NLL + gradient pass, the adjoint convolutions, the GMM prior value + gradient, (N > 1) ONE RCCL
all-reduce of the likelihood gradient (started before the prior, overlapped with it) and one all-gather of the compact
prior bands, the fused chain rule + Adam update.  The convolution method is "auto": the
benchmark's Gaussian PSFs are rank 1, so the headline runs the separable strip-walk kernels (the PSFs of both sizes share
one plan and one batched step: forward launch = convolution + Poisson pass of all local observations); the same fit
with the PSFs treated as general kernels (MFMA direct convolution up to 17 taps, native FFT beyond: `general_psf`) and through the FFT path (`fft_psf`,
the path the north star names: the native FFT convolution of csrc/fftnative.hip on these sizes) is timed beside it, and
so is `c6_chandra_like` (calibrations + up-sampling x2 + general 65x65 PSFs: the reference's Chandra example).
Total work is fixed as N grows (observations round-robin over the ranks, the prior split by patch
rows): strong scaling.  All inputs are resident in HBM before the timed region.

Timing protocol: W warm-up steps, then a settle phase of at least 0.3 s of steps (`settle`, its own field: clocks and
caches reach their sustained state), then R = `--repeats` timed regions of EXACTLY K steps each, every one bracketed by
barrier + synchronize and reduced by MAX over the ranks; `ms_per_step` / `value` are the MEDIAN region (min and max are
reported beside it).  The kernel timers (hipEvent pairs recorded by the library around every launch, on the launch
stream) run in a separate, untimed phase of >= 16 steps after the timed regions, so they neither perturb the timed
steps nor rest on two samples.  `clock_mhz` is the shader clock the device holds under a vector-ALU load on every CU
(jd_clock_probe): boards differ by several percent, and the line says which one it was measured on.

Prints ONE JSON line (rank 0).  `roofline` describes the dominant kernel of the step (the fp16 screen of the
GMM arg-max: matrix-core roof; `roofline_section8d` prices the whole GMM forward pass by SURVEY section 8(d)'s
dense fp32 definition), `roofline_poisson` the fused Poisson pass (HBM roof), `roofline_fft` the five launches of one
observation's likelihood step on the native FFT convolution (the FFT side run).  `--config c2|c4|c5` run the other BASELINE configurations (parity-test cases; the default c3
is the one the metric is quoted on).  `cpu_baseline` times oracle/cpu_ref.py (the PyTorch-CPU restatement of the
reference) on a bounded sample on rank 0 at N = 1.
"""
import argparse
import gc
import json
import os
import sys
import time
from pathlib import Path

REPO = Path(__file__).resolve().parent
if str(REPO) not in sys.path:
    sys.path.insert(0, str(REPO))

import numpy as np  # noqa: E402
import torch  # noqa: E402

CONFIGS = {
    # name: (H, W, n_obs, K)
    "c1": (128, 128, 1, 0),  # BASELINE configs[0]: point source, Gaussian PSF, uniform prior, the reference's own loop (sequential)
    "c2": (1024, 1024, 1, 128),
    "c3": (2048, 2048, 8, 128),
    "c4": (4096, 4096, 1, 128),
    "c5": (2048, 2048, 16, 128),  # two flux components: "extended" (GMM prior) + "points" (inverse-gamma prior)
    # c3's workload on an image with an odd number of rows and a width that is no multiple of 4 (side run `odd_size_fft`:
    # until round 5 such sizes left the native FFT path for un-batched rocFFT plans)
    "c3odd": (2047, 2050, 8, 128),
}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP32_MATRIX_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 = fp32 vector rate
F16_MATRIX_PEAK_TFLOPS = 2516.6  # 16 x the fp32 rate: v_mfma_f32_32x32x16_f16, dense (MI355X_MICROARCH.md: ~2.5 PF)
PATCH, D, STRIDE = 8, 64, 4
PROFILE_STEPS = 16  # steps of the untimed profile phase: every launch of every one of them is bracketed by a hipEvent pair
SETTLE_SECONDS = 0.3


def build_session(cfg_name, device, seed=0, dist=None, fit_mode="joint", shared_psf=False):
    from jolideco_amd import GMMPatchPrior, MAPDeconvolver, SpatialFluxComponent
    from jolideco_amd.data import synthetic_gmm, synthetic_observations
    from jolideco_amd.priors.patches import GaussianMixtureModel, GaussianMixtureModelMeta

    H, W, n_obs, K = CONFIGS[cfg_name]
    if cfg_name == "c1":
        # 128^2, delta source of 1000 counts, PSF sigma 3 on 17x17, exposure 1, background 2,
        # uniform prior, seed 428723, flux_init ~ gamma(30) (examples/first-steps.py:53), the reference's sequential loop
        from jolideco_amd import UniformPrior
        from jolideco_amd.data import point_source_gauss_psf

        rs = np.random.RandomState(428723)
        data = point_source_gauss_psf(shape=(H, W), shape_psf=(17, 17), sigma_psf=3, source_level=1000, random_state=rs)
        data.pop("flux")
        comp = SpatialFluxComponent.from_numpy(flux=rs.gamma(30, size=(H, W)), prior=UniformPrior())
        deconvolver = MAPDeconvolver(n_epochs=1, display_progress=False, device=device, fit_mode="sequential")
        return deconvolver.session({"obs-0": data}, components=comp, dist=dist)
    datasets, _, flux_init = synthetic_observations(shape=(H, W), n_obs=n_obs, seed=seed)
    means, covs, weights = synthetic_gmm(K, D, seed=0)
    gmm = GaussianMixtureModel.from_numpy(means, covs, weights, meta=GaussianMixtureModelMeta(stride=STRIDE))
    comp = SpatialFluxComponent.from_numpy(flux=flux_init, prior=GMMPatchPrior(gmm=gmm))
    if cfg_name == "c5":
        from jolideco_amd import FluxComponents, InverseGammaPrior
        from jolideco_amd.data import gaussian_kernel

        comps = FluxComponents()
        comps["extended"] = comp
        comps["points"] = SpatialFluxComponent.from_numpy(
            flux=0.05 * flux_init, prior=InverseGammaPrior(alpha=10, beta=1.5)
        )
        # per-component PSFs: the point sources see a sharper core.  shared_psf (side run): ONE PSF
        # array per dataset for both components, the reference's default (models/npred.py:279-295) -- the components are then
        # evaluated as their sum (PoissonLoss.fwd_bwd_batch)
        for i, d in enumerate(datasets.values()):
            if not shared_psf:
                d["psf"] = {"extended": d["psf"], "points": gaussian_kernel(1.0 + 0.1 * i, (17, 17)).astype(np.float32)}
        comp = comps
    deconvolver = MAPDeconvolver(n_epochs=1, display_progress=False, device=device, fit_mode=fit_mode)
    return deconvolver.session(datasets, components=comp, dist=dist)


def build_session_c6(device, shape=(2048, 2048), n_obs=8, seed=0, K=128):
    """Config "c6" (round-3 verdict): the fit the reference's Chandra example runs (examples/chandra-e0102-filament.py:
    91-93,178-203) at the benchmark's size -- 2048^2 counts grid, 8 observations, ``upsampling_factor=2`` (flux, exposure,
    PSF and the GMM prior live on the 4096^2 grid), general 65x65 PSFs (130x130 after up-sampling: the native FFT
    convolution, with the sum-pool + Poisson pass fused between its column passes), one
    `NPredCalibration` per observation (trained sub-pixel shift + background norm), joint fit."""
    from jolideco_amd import GMMPatchPrior, MAPDeconvolver, NPredCalibration, NPredCalibrations, SpatialFluxComponent
    from jolideco_amd.data import instrument_observations, synthetic_gmm
    from jolideco_amd.priors.patches import GaussianMixtureModel, GaussianMixtureModelMeta

    datasets, _, flux_init, cal = instrument_observations(shape=shape, n_obs=n_obs, seed=seed)
    means, covs, weights = synthetic_gmm(K, D, seed=0)
    gmm = GaussianMixtureModel.from_numpy(means, covs, weights, meta=GaussianMixtureModelMeta(stride=STRIDE))
    comp = SpatialFluxComponent.from_numpy(flux=flux_init, prior=GMMPatchPrior(gmm=gmm), upsampling_factor=2)
    calibrations = NPredCalibrations()
    for name, (sx, sy, norm) in cal.items():
        calibrations[name] = NPredCalibration(shift_x=sx, shift_y=sy, background_norm=norm)
    deconvolver = MAPDeconvolver(n_epochs=1, display_progress=False, device=device, fit_mode="joint")
    return deconvolver.session(datasets, components=comp, calibrations=calibrations)


def c6_roofline(plan, counts_shape, n_obs, u, kernel_ms):
    """The likelihood step of ONE calibrated, up-sampled observation on the native FFT path against the HBM roof: six
    launches -- rows (bilinear shift x exposure -> row spectra), columns (x kernel spectrum), pooled middle (rows^-1 x U,
    sum-pool, Poisson pass, rows of the up-sampled g), columns (conjugate), rows^-1 + adjoint epilogue, transposed shift --
    with the ALGORITHMIC bytes of each (every array counted once per launch that streams it) over the
    launch durations the library's hipEvent pairs measured."""
    H, W = u * counts_shape[0], u * counts_shape[1]
    hh, kh, kw = H // 2, plan.kh, plan.kw

    def fft_length(n, with_three=True):
        best = None
        for odd in (1, 3, 9) if with_three else (1, 9):
            m = 8
            while m * odd < max(n, 32):
                m *= 2
            best = m * odd if best is None or m * odd < best else best
        return best

    nx, ny = fft_length(W + max((kw - 1) // 2, kw - 1 - (kw - 1) // 2)), fft_length(hh + kh - 1, False)
    pool = u if (u > 1 and ny % u == 0 and hh % u == 0 and os.environ.get("JD_FFT_POOL_IO", "1") != "0") else 1
    img, cnt = 4.0 * H * W, 4.0 * counts_shape[0] * counts_shape[1]
    spec, kept, khat = 8.0 * hh * nx, 8.0 * (hh + kh - 1) * nx, 8.0 * nx * ny
    launches = {  # timer -> (what, algorithmic bytes per observation, launches per observation)
        "fft_r2c": ("rows: flux (shifted) + exposure in, row spectra out", 2 * img + spec, 1),
        # (round 5: the column passes hand the pooled launch the SUMS of U rows and take ONE row per counts row back)
        "cmul": ("columns: spectra + kernel spectrum in, kept rows out (forward: sums of U rows out; adjoint: one row per counts row in)",
                 2 * khat + (spec + kept) * (1 + 1 / pool), 2),
        "poisson_fused": ("pooled middle: kept row sums + background + counts in, one row spectrum of g per counts row out",
                          (kept + spec) / pool + 2 * cnt, 1),
        "fft_c2r": ("rows^-1 + adjoint epilogue: kept rows + exposure in, exposure x corr out", kept + 2 * img, 1),
        # (one launch over all observations: each reads its own exposure x corr image; the flux and the gradient -- read and
        # written ONCE by the launch since round 5, in registers over the datasets -- are shared: counted once per launch)
        "shift": ("transposed shift: exposure x corr in; flux in + gradient out shared by the observations of the launch",
                  img + 2 * img / n_obs, 1),
    }
    rows, total_bytes, total_ms = {}, 0.0, 0.0
    for name, (what, nbytes, per_obs) in launches.items():
        ms = kernel_ms.get(name)
        if not ms:
            continue
        ms_obs = ms / n_obs
        rows[name] = {"what": what, "bytes_per_observation": nbytes, "ms_per_observation": ms_obs, "launches_per_observation": per_obs,
                      "achieved": nbytes / (ms_obs * 1e-3) / 1e9, "frac": nbytes / (ms_obs * 1e-3) / 1e9 / HBM_PEAK_GBS}
        total_bytes += nbytes
        total_ms += ms_obs
    if not total_ms:
        return None
    achieved = total_bytes / (total_ms * 1e-3) / 1e9
    return {"kernel": "native FFT likelihood step of one calibrated, up-sampled observation (6 launches)", "bound": "hbm",
            "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
            "bytes_per_observation": total_bytes, "ms_per_observation": total_ms, "padded_fft_grid": [ny, nx], "launches": rows,
            "note": "PMC traffic of these kernels: profiles/r05/pmc_hbm_traffic.csv rows c6"}


def c6_run(device, dist_ctx, steps=20, warmup=3, repeats=3, shape=(2048, 2048), n_obs=8):
    """Time config c6 (see `build_session_c6`): it/s, the per-kernel table and `roofline_c6` (the six launches of one
    observation's likelihood step against the HBM roof)."""
    session = build_session_c6(device, shape=shape, n_obs=n_obs)
    for _ in range(max(warmup, 10)):
        session.epoch()
    warm_policy(session)  # (the "auto" policy's probe and trial epochs)
    torch.cuda.synchronize(device)
    stats = region_stats(timed_regions(session, steps, repeats, device, dist_ctx), steps)
    host_ms = 1e3 * float(np.median(HOST_ENQUEUE[-1])) / steps
    replayed = bool(getattr(session, "_graphs", None))
    prof = profile_phase(session, device, n_obs, steps=4)
    nested = ("gmm_stage", "gmm_screen", "gmm_sort", "gmm_exact")
    models = session.total_loss.poisson_loss.npred_models_all
    plan = models[0].plan
    scal = session.scalars.detach().cpu().numpy()
    if not np.all(np.isfinite(scal)):
        raise SystemExit(f"c6: non-finite losses {scal}")
    out = {
        "value": stats["value"], "unit": "iters/s", "ms_per_step": stats["ms_per_step"], "steps": steps, "repeats": repeats,
        "host_enqueue_ms_per_step": host_ms,
        "workload": f"c6: {shape[0]}x{shape[1]} counts grid, {n_obs} observations, upsampling_factor 2 (flux grid "
                    f"{2 * shape[0]}x{2 * shape[1]}), general 65x65 PSFs ({plan.kh}x{plan.kw} up-sampled), one NPredCalibration "
                    "(shift + background norm, trained) per observation, GMM patch prior K=128 on the flux grid, joint fit",
        "conv_method": "+".join(sorted({m.plan.method for m in models})),
        "padded_grid": [plan.Hp, plan.Wp],
        "batched": bool(getattr(session, "batch_joint", False)),
        # the calibrated batched entry (jd_npred_poisson_calibrated_batch_fwd_bwd) is called; beyond 2048 flux rows the
        # library runs its per-dataset launches (measured faster there)
        "batched_calibrated_entry": bool(getattr(session, "batch_joint_calibrated", False)),
        "graph_policy": getattr(session, "graph_policy", None), "epochs_replayed_from_graphs": replayed,
        "kernel_ms_per_step": {k: v[0] / 4 for k, v in prof.items() if v[1] and k not in nested},
        "launches_per_step": {k: v[1] / 4 for k, v in prof.items() if v[1] and k not in nested},
    }
    out["roofline_c6"] = c6_roofline(plan, shape, n_obs, 2, out["kernel_ms_per_step"])
    del session
    return out


def e0102_run(device, n_epochs=250, shape=(256, 256), n_obs=24, psf_shape=(128, 128), K=128):
    """Time-to-solution of the fit the reference publishes a runtime for (examples/chandra-e0102-filament.py:91-93,178-222:
    24 Chandra observations of one field, 128x128 MARX PSFs, ``upsampling_factor=2``, one `NPredCalibration` per
    observation, GMM patch prior, ``n_epochs=250`` in the reference's own SEQUENTIAL mode -- "takes about 30 min on an M1
    cpu"), end to end through `MAPDeconvolver.run()`: session set-up, 250 epochs x 24 optimizer steps (each with the full
    prior) + the trace evaluation of every epoch, the trace read back at the end.  The tutorial's counts grid is not in the
    reference tree: ASSUMED 256 x 256 (flux grid 512 x 512); synthetic counts, synthetic general PSFs and a synthetic
    K = 128 mixture of the same shapes."""
    from jolideco_amd import GMMPatchPrior, MAPDeconvolver, NPredCalibration, NPredCalibrations, SpatialFluxComponent
    from jolideco_amd.data import instrument_observations, synthetic_gmm
    from jolideco_amd.priors.patches import GaussianMixtureModel, GaussianMixtureModelMeta

    datasets, _, flux_init, cal = instrument_observations(shape=shape, n_obs=n_obs, seed=0, psf_shape=psf_shape)
    means, covs, weights = synthetic_gmm(K, D, seed=0)

    def build():
        gmm = GaussianMixtureModel.from_numpy(means, covs, weights, meta=GaussianMixtureModelMeta(stride=STRIDE))
        comp = SpatialFluxComponent.from_numpy(flux=flux_init, prior=GMMPatchPrior(gmm=gmm), upsampling_factor=2)
        calibrations = NPredCalibrations()
        for name, (sx, sy, norm) in cal.items():
            calibrations[name] = NPredCalibration(shift_x=sx, shift_y=sy, background_norm=norm)
        return comp, calibrations

    # a short run first: library load, kernel attributes, allocator warm-up (the timed run still builds its own session)
    comp, calibrations = build()
    MAPDeconvolver(n_epochs=2, display_progress=False, device=device, fit_mode="sequential").run(
        datasets, components=comp, calibrations=calibrations)
    torch.cuda.synchronize(device)
    comp, calibrations = build()
    deco = MAPDeconvolver(n_epochs=n_epochs, display_progress=False, device=device, fit_mode="sequential")
    t0 = time.perf_counter()
    result = deco.run(datasets, components=comp, calibrations=calibrations)
    torch.cuda.synchronize(device)
    wall = time.perf_counter() - t0
    totals = np.asarray(result.trace_loss["total"], dtype=np.float64)
    if not np.all(np.isfinite(totals)) or not np.all(np.isfinite(result.flux_total)):
        raise SystemExit("e0102: non-finite trace or flux")
    # the epochs alone (no set-up), eagerly enqueued against replayed: a session of the same fit
    comp, calibrations = build()
    session = MAPDeconvolver(n_epochs=1, display_progress=False, device=device, fit_mode="sequential").session(
        datasets, components=comp, calibrations=calibrations)
    for _ in range(16):
        session.epoch()
    warm_policy(session)  # (the "auto" policy's probe and trial epochs)
    torch.cuda.synchronize(device)
    gc.collect()
    gc.disable()
    try:
        n = 30
        t0 = time.perf_counter()
        for _ in range(n):
            session.epoch()
        t_enq = time.perf_counter() - t0
        torch.cuda.synchronize(device)
        t_epoch = (time.perf_counter() - t0) / n
    finally:
        gc.enable()
    steps = n_epochs * n_obs
    return {
        "metric": "time to solution, reference example chandra-e0102-filament (sequential MAP fit)", "unit": "s", "value": wall,
        "higher_is_better": False, "n_gpus": 1, "dtype": "f32", "data": "synthetic", "scaling": "strong", "vs_baseline": None,
        "steps": steps, "warmup": 0, "ms_per_step": 1e3 * wall / steps, "epochs": n_epochs,
        "config": {"workload": f"e0102: {n_obs} observations, counts grid {shape[0]}x{shape[1]} (ASSUMED: the tutorial's size is not in "
                               f"the reference tree), {psf_shape[0]}x{psf_shape[1]} general PSFs, upsampling_factor 2 (flux grid "
                               f"{2 * shape[0]}x{2 * shape[1]}), one NPredCalibration per observation, GMM patch prior K={K}, "
                               f"fit_mode sequential, {n_epochs} epochs = {steps} optimizer steps + {n_epochs} trace evaluations, "
                               "MAPDeconvolver.run() end to end"},
        "reference_runtime": "about 30 min on an M1 cpu (examples/chandra-e0102-filament.py:216-222; real data, other hardware: "
                             "quoted for scale, not a measured baseline)",
        "epoch_ms": 1e3 * t_epoch, "host_enqueue_ms_per_step": 1e3 * t_enq / (n * n_obs),
        "epochs_replayed_from_graphs": bool(session._graphs), "graph_policy": session.graph_policy,
        "check": {"total_first": float(totals[0]), "total_last": float(totals[-1]), "decreasing": bool(totals[-1] < totals[0])},
    }


PMC_TRAFFIC_FILES = ("profiles/r05/pmc_hbm_traffic.csv", "profiles/r04/pmc_hbm_traffic.csv", "profiles/r03/pmc_hbm_traffic.csv", "profiles/r02/pmc_hbm_traffic.csv", "profiles/r01/pmc_hbm_traffic.csv")
_TRAFFIC_USED = {}  # kernel -> file its traffic figure came from


def pmc_traffic_source(kernel=None):
    """Where `traffic` comes from: PMC counters cannot be read from inside this process, so the bench line quotes
    the committed rocprofv3 PMC passes -- file and the commit that last touched it (stale once a kernel changes)."""
    rel = _TRAFFIC_USED.get(kernel) if kernel else next((r for r in PMC_TRAFFIC_FILES if (REPO / r).exists()), None)
    if rel is None:
        return None
    commit = "unknown"
    try:
        import subprocess

        commit = subprocess.run(["git", "log", "-n1", "--format=%h", "--", rel], cwd=REPO, capture_output=True,
                                text=True, timeout=10).stdout.strip() or "unknown"
    except Exception:  # no git on the GPU box's copy: the sidecar written when the file was committed
        pass
    sidecar = REPO / (rel + ".commit")
    if commit == "unknown" and sidecar.exists():
        commit = sidecar.read_text().strip() or "unknown"
    return f"{rel} @{commit} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, not measured by this run)"


def pmc_traffic_bytes(cfg_name, kernel):
    """HBM bytes per launch of `kernel` (a prefix of its rocprofv3 name) from the committed rocprofv3 PMC passes
    (FETCH_SIZE and WRITE_SIZE collected in separate passes, KB per launch); the newest round's file that holds the
    kernel wins.  gfx950 correction of MI355X_MICROARCH.md: reads = 2 x FETCH_SIZE, writes = WRITE_SIZE.  None when
    the workload was not profiled.  (PMC counters cannot be read from inside this process.)"""
    import csv

    for rel in PMC_TRAFFIC_FILES:
        if not (REPO / rel).exists():
            continue
        fetch = write = None
        with open(REPO / rel) as fh:
            rows = list(csv.DictReader(fh))
        if not rows or not {"config", "kernel", "counter", "avg_per_launch_KB"} <= set(rows[0]):
            continue  # (a malformed profile file must not stop the benchmark: traffic is reported as null)
        # "<config>s" / "<config>f" / "<config>g" rows: the same workload profiled on later builds -- the last one
        # found wins (the .commit sidecar names the build of the newest rows)
        for label in (cfg_name, cfg_name + "s", cfg_name + "f", cfg_name + "g", cfg_name + "h"):
            for row in rows:
                if row["config"] == label and kernel in row["kernel"]:
                    if row["counter"] == "FETCH_SIZE":
                        fetch = float(row["avg_per_launch_KB"])
                    elif row["counter"] == "WRITE_SIZE":
                        write = float(row["avg_per_launch_KB"])
        if fetch is not None and write is not None:
            _TRAFFIC_USED[kernel] = rel
            return (2.0 * fetch + write) * 1024.0
    return None


def host_cores(cap=16):
    """Cores this process may really use: affinity mask, cgroup CPU quota and the GPU box's share
    (16 cores per GPU) -- over-subscribing a quota-limited container makes the CPU leg crawl."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = Path("/sys/fs/cgroup/cpu.max").read_text().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, cap))


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def cpu_baseline_c1(n_epochs=200):
    """BASELINE configs[0] is the reference's own CPU-runnable case: the oracle's sequential loop on the full workload."""
    from jolideco_amd.data import point_source_gauss_psf
    from oracle import cpu_ref

    cores = host_cores()
    torch.set_num_threads(cores)
    rs = np.random.RandomState(428723)
    data = point_source_gauss_psf(shape=(128, 128), shape_psf=(17, 17), sigma_psf=3, source_level=1000, random_state=rs)
    data.pop("flux")
    flux_init = rs.gamma(30, size=(128, 128))
    cpu_ref.map_fit_sequential({"obs-0": data}, {"flux": flux_init}, {"flux": cpu_ref.UniformPriorRef()}, n_epochs=5)  # warm-up
    t0 = time.perf_counter()
    cpu_ref.map_fit_sequential({"obs-0": data}, {"flux": flux_init}, {"flux": cpu_ref.UniformPriorRef()}, n_epochs=n_epochs)
    dt = (time.perf_counter() - t0) / n_epochs
    return {"value": 1.0 / dt, "unit": "iters/s", "cores": cores, "kind": "port",
            "sample": f"oracle/cpu_ref.py map_fit_sequential (autograd, torch {torch.__version__} CPU, {cores} threads), the full "
                      f"128x128 workload, {n_epochs} epochs (1 step + 1 trace evaluation each): {1e3 * dt:.2f} ms per epoch",
            "sample_seconds_per_step": dt}


def cpu_baseline(cfg_name, sample_edge=1024, max_steps=10, budget_s=15.0):
    """Time the CPU oracle (the PyTorch-CPU restatement of the reference's joint step) on a
    `sample_edge`^2 crop of the same workload (same number of observations, same PSFs, same GMM)
    and scale to the full image by the pixel ratio (the cost is linear in the patch count)."""
    from jolideco_amd.data import synthetic_gmm, synthetic_observations
    from oracle import cpu_ref

    if cfg_name == "c1":
        return cpu_baseline_c1()
    H, W, n_obs, K = CONFIGS[cfg_name]
    cores = host_cores()
    torch.set_num_threads(cores)
    edge = min(sample_edge, H)
    datasets, _, flux_init = synthetic_observations(shape=(edge, edge), n_obs=n_obs, seed=0)
    means, covs, weights = synthetic_gmm(K, D, seed=0)
    gmm = cpu_ref.GMM.from_numpy(means, covs, weights, stride=STRIDE)
    prior = cpu_ref.GMMPatchPriorRef(gmm)
    theta = cpu_ref.log_flux_parameter(flux_init)
    data = [cpu_ref.DatasetRef.from_numpy(d, ["flux"]) for d in datasets.values()]
    optimizer = torch.optim.Adam([theta], lr=0.1)

    def step():
        optimizer.zero_grad()
        fluxes = (cpu_ref.to_flux(theta),)
        total, _, _ = cpu_ref.joint_loss(data, fluxes, [prior], 1.0)
        total.backward()
        optimizer.step()

    step()  # warm-up
    log("cpu_baseline: warm-up step done")
    t0 = time.perf_counter()
    steps = 0
    while steps < max_steps and (steps == 0 or time.perf_counter() - t0 < budget_s):
        step()
        steps += 1
    dt = (time.perf_counter() - t0) / steps
    scale = (H * W) / float(edge * edge)
    return {
        "value": 1.0 / (dt * scale),
        "unit": "iters/s",
        "cores": cores,
        "kind": "port",
        "sample": (
            f"oracle/cpu_ref.py joint step (autograd, torch {torch.__version__} CPU, {cores} threads) on a "
            f"{edge}x{edge} crop, {n_obs} obs, GMM K={K}: {dt:.3f} s/step over {steps} steps after 1 warm-up; "
            f"scaled by the pixel ratio {scale:.0f} to {H}x{W}"
        ),
        "sample_seconds_per_step": dt,
    }


def warm_policy(session, cap=128):
    """Epochs until the session's "auto" policy has decided (probe epochs on two streams / one, a trial of replayed
    epochs where the host weighs in): nothing of that belongs into a timed region."""
    n = 0
    while n < cap and (getattr(session, "graph_policy", "") == "undecided" or getattr(session, "_trial", None) is not None):
        session.epoch()
        n += 1
    return n


def settle(session, device, dist_ctx, seconds=SETTLE_SECONDS, chunk=20):
    """Steps for at least `seconds` before the timed regions (chunks of `chunk` steps; every rank runs the same number:
    the ranks agree after each chunk whether all of them have had enough)."""
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    n = 0
    while True:
        for _ in range(chunk):
            session.epoch()
        n += chunk
        torch.cuda.synchronize(device)
        more = time.perf_counter() - t0 < seconds
        if dist_ctx.world_size > 1:
            t = torch.tensor([1.0 if more else 0.0], dtype=torch.float64, device=device)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            more = bool(t.item() > 0)
        if not more:
            break
    return {"seconds": time.perf_counter() - t0, "steps": n}


HOST_ENQUEUE = []  # per call of timed_regions: seconds the host needs to enqueue a region's steps (MAX over the ranks)
# The host's own cost of a step is measured on the first steps of a region, issued into an EMPTY queue: once a few dozen
# captured epochs are in flight hipGraphLaunch waits for the device, and the time until "everything is enqueued" becomes
# the device's time (200 replayed steps: 0.44 ms per step "to enqueue" against 0.04 for the first 16).
ENQUEUE_BURST = 16


def timed_regions(session, steps, repeats, device, dist_ctx):
    """`repeats` regions of exactly `steps` steps, each bracketed by barrier + synchronize; per region the MAX over the
    ranks.  Returns the list of region times in seconds."""
    times, enqueue = [], []
    # Python's cycle collector stays out of the timed steps: when it frees the device buffers of an earlier session
    # (set-up objects, a side run) inside the loop, every hipFree synchronises the device
    gc.collect()
    gc.disable()
    try:
        for _ in range(repeats):
            torch.cuda.synchronize(device)
            dist_ctx.barrier()
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            burst = min(steps, ENQUEUE_BURST)
            for _ in range(burst):
                session.epoch()
            t_enqueued = time.perf_counter()  # the host has issued the first `burst` steps into an empty queue
            for _ in range(steps - burst):
                session.epoch()
            torch.cuda.synchronize(device)
            dist_ctx.barrier()
            torch.cuda.synchronize(device)
            times.append(time.perf_counter() - t0)
            enqueue.append((t_enqueued - t0) * steps / burst)  # (scaled to the region: callers divide by `steps`)
    finally:
        gc.enable()
    if dist_ctx.world_size > 1:
        t = torch.tensor([times, enqueue], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        times, enqueue = [float(v) for v in t[0].cpu()], [float(v) for v in t[1].cpu()]
    HOST_ENQUEUE.append(enqueue)
    return times


def profile_phase(session, device, n_obs, steps=PROFILE_STEPS):
    """Untimed: `steps` steps with every launch bracketed by a hipEvent pair.  {kernel: (total ms, launches)}"""
    from jolideco_amd import _hip

    _hip.profile_enable(capacity=min(1 << 16, 64 * (n_obs + 2) * steps))
    for _ in range(steps):
        session.epoch()
    torch.cuda.synchronize(device)
    return _hip.profile_read()


def region_stats(times, steps):
    med = float(np.median(times))
    return {"ms_per_step": 1e3 * med / steps, "ms_per_step_min": 1e3 * min(times) / steps,
            "ms_per_step_max": 1e3 * max(times) / steps, "value": steps / med}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--repeats", type=int, default=9, help="timed regions of --steps steps each; the median is reported")
    ap.add_argument("--settle-seconds", type=float, default=SETTLE_SECONDS,
                    help="steps run for at least this long between the warm-up and the timed regions (0: none)")
    ap.add_argument("--config", default="c3", choices=sorted(c for c in CONFIGS if c != "c3odd") + ["c6", "e0102"])
    ap.add_argument("--epochs", type=int, default=250, help="--config e0102: epochs of the sequential fit")
    ap.add_argument("--no-c6", action="store_true", help="skip the c6 side run (calibrations + up-sampling + general 65x65 PSFs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-general-psf", action="store_true",
                    help="skip the extra run that convolves the PSFs as general (not separable) kernels")
    ap.add_argument("--shard-of", type=int, default=0,
                    help="tuning only: time one rank's share (--rank, default 0) of an N-rank joint step in ONE process, "
                         "without the collective (the printed value is NOT a benchmark result); tools/shard_table.py "
                         "times every rank of N = 2 / 4 / 8 that way")
    ap.add_argument("--rank", type=int, default=0, help="with --shard-of N: the rank whose share is timed")
    args = ap.parse_args()

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an AMD GPU (libjolideco_hip.so has no CPU fallback)")
    from jolideco_amd import _hip
    from jolideco_amd.distributed import init_from_env

    dist_ctx = init_from_env()
    world = dist_ctx.world_size
    # what the BACKEND saw (not what the flags said): a SCALE record can be checked for "RCCL saw N ranks"
    dist_info = {"backend": None, "world_size_seen_by_backend": 1, "rank": 0}
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        dist_info = {
            "backend": torch.distributed.get_backend(), "world_size_seen_by_backend": torch.distributed.get_world_size(),
            "rank": torch.distributed.get_rank(),
        }
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count()
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)

    if args.config == "e0102":
        if world != 1:
            raise SystemExit("--config e0102 is a single-GPU run (sequential mode: replicas only)")
        log("e0102: time to solution of the reference's Chandra example shape")
        print(json.dumps(e0102_run(device, n_epochs=args.epochs)))
        return
    if args.config == "c6":
        if world != 1:
            raise SystemExit("--config c6 is a single-GPU side run")
        log("c6: calibrations + up-sampling x2 + general 65x65 PSFs")
        out = c6_run(device, dist_ctx, steps=max(args.steps // 10, 5), warmup=max(args.warmup // 5, 2), repeats=min(args.repeats, 3))
        out = dict({"metric": "MAP iters/sec (c6)", "n_gpus": 1, "higher_is_better": True, "dtype": "f32", "data": "synthetic",
                    "scaling": "strong", "vs_baseline": None, "warmup": max(args.warmup // 5, 2), "config": {"workload": out.pop("workload")}}, **out)
        print(json.dumps(out))
        return
    H, W, n_obs, K = CONFIGS[args.config]
    if args.config == "c1":
        args.no_general_psf = True  # (the side runs are variants of the joint GMM fits)
    log(f"building {args.config}: {H}x{W}, {n_obs} obs, K={K} on {device}")
    fake = None
    if args.shard_of > 1:
        from jolideco_amd.distributed import DistContext

        if not 0 <= args.rank < args.shard_of:
            raise SystemExit(f"--rank {args.rank} is not a rank of --shard-of {args.shard_of}")
        fake = DistContext(rank=args.rank, world_size=args.shard_of, dry_run=True)
    session = build_session(args.config, device, dist=fake)
    torch.cuda.synchronize(device)
    log("session ready; warm-up")

    for _ in range(args.warmup):
        session.epoch()
    if dist_ctx.world_size == 1:
        warm_policy(session)  # (a single-process fit's "auto" policy: its probe and trial epochs are not timed)
    torch.cuda.synchronize(device)
    settled = settle(session, device, dist_ctx, args.settle_seconds) if args.settle_seconds > 0 else {"seconds": 0.0, "steps": 0}
    log(f"settled: {settled['steps']} steps in {settled['seconds']:.3f} s; timed regions")
    times = timed_regions(session, args.steps, max(args.repeats, 1), device, dist_ctx)
    stats = region_stats(times, args.steps)
    host_enqueue_ms = 1e3 * float(np.median(HOST_ENQUEUE[-1])) / args.steps
    replayed, graph_policy = bool(getattr(session, "_graphs", None)), getattr(session, "graph_policy", None)
    elapsed = args.steps / stats["value"]  # the median region
    log(f"timed regions done: {stats['ms_per_step']:.4f} ms/step (median of {len(times)}; "
        f"{stats['ms_per_step_min']:.4f} .. {stats['ms_per_step_max']:.4f})")
    clock_mhz = _hip.clock_probe(2.0, device)
    # numerics of the run itself: the loss scalars after the last TIMED step (before the untimed profile phase)
    torch.cuda.synchronize(device)
    scal = session.scalars.detach().cpu().numpy()
    epochs_run = args.warmup + settled["steps"] + args.steps * len(times)
    if world > 1:
        session.comm_events = []
    prof = profile_phase(session, device, n_obs)
    comm_ms = session.comm_times_ms() if world > 1 else None
    session.comm_events = None
    # the shader clock INSIDE the screen kernel (round-4 verdict item 6): an untimed phase on the stamped instantiation of
    # the kernel (jd_gmm_screen_clock) -- s_memtime / s_memrealtime ticks of up to 4096 blocks per launch
    screen_clock = None
    handles = [p.gmm.handle(device) for p in session.priors if hasattr(p, "gmm")]
    if handles and world == 1 and args.shard_of <= 1:
        handles[0].screen_clock()  # arms the handle
        session.reset_graphs()
        for _ in range(max(args.warmup, 8)):
            session.epoch()
        mhz, samples = handles[0].screen_clock()
        screen_clock = {"mhz": mhz, "blocks_sampled": samples} if samples else None

    # sanity: the fit must have produced finite numbers
    if not np.all(np.isfinite(scal)):
        raise SystemExit(f"non-finite losses after the timed region: {scal}")

    if dist_ctx.rank != 0:
        if world > 1:
            torch.distributed.destroy_process_group()
        return

    ms_per_step = 1e3 * elapsed / args.steps
    n_py, n_px = (H - PATCH) // STRIDE + 1, (W - PATCH) // STRIDE + 1
    shares = getattr(session, "prior_shares", None)  # (cost-aware placement: a rank with costly datasets takes fewer rows)
    rows = dist_ctx.shard_range(n_py, shares) if world > 1 else (0, n_py)
    if fake is not None:
        rows = fake.shard_range(n_py, shares)
    np_local = (rows[1] - rows[0]) * n_px

    def avg_ms(name):
        total, count = prof[name]
        return (total / count if count else None), count

    # GMM forward pass.  Algorithmic work per (patch, component): P_k is upper triangular, so D (D + 1) multiply-adds
    # + 4 D epilogue flop = D^2 + 5 D (the SURVEY section 8(d) figure 2 D^2 + 4 D counts the structural zeros of P_k).
    #  * dense kernel (logsumexp mode, mixtures with means, JD_GMM_SCREEN=0): fp32 matrix cores, 40 non-zero
    #    16 x 16 x 4 MFMA blocks per 16 patches = 5120 flop + 4 D executed;
    #  * screened arg-max (default, max mode): stage 1 `gmm_screen_kernel` evaluates every (patch, component) with ONE
    #    fp16 MFMA product (6 blocks of 32 x 32 x 16 per 32 patches = 6144 flop) plus a rigorous error bound, stage 3
    #    re-evaluates the few survivors in fp32 -- same bits out as the dense kernel.  The dominant kernel is then
    #    the screen: its roofline is the fp16 matrix peak, `achieved` counts the fp16 flop it executes, and the
    #    fp32-equivalent algorithmic rate of the whole forward pass is reported next to it.
    gmm_ms, gmm_n = avg_ms("gmm_fwd")
    scr_ms, scr_n = avg_ms("gmm_screen")
    gmm_flop = np_local * K * (D * D + 5 * D)
    gmm_flop_executed = np_local * K * (40 * 2 * 16 * 16 * 4 // 16 + 4 * D)
    gmm_flop_dense = np_local * K * (2 * D * D + 4 * D)
    roof_gmm = None
    if scr_ms:
        f16_flop = np_local * K * (6 * 2 * 32 * 32 * 16 // 32)
        achieved = f16_flop / (scr_ms * 1e-3) / 1e12
        roof_gmm = {
            "kernel": "gmm_screen_kernel", "bound": "mfma", "achieved": achieved, "peak": F16_MATRIX_PEAK_TFLOPS,
            "unit": "TFLOP/s", "frac": achieved / F16_MATRIX_PEAK_TFLOPS,
            "traffic": pmc_traffic_bytes(args.config, "gmm_screen_kernel") if world == 1 and fake is None else None,
            "avg_launch_ms": scr_ms, "launches": scr_n, "flop_per_launch": f16_flop, "operand_dtype": "f16 (screen only)",
            "forward_pass_ms": gmm_ms, "stage_ms": {k: avg_ms(k)[0] for k in ("gmm_stage", "gmm_screen", "gmm_sort", "gmm_exact")},
            "algorithmic_fp32_flop": gmm_flop,
            "algorithmic_fp32_equivalent_tflops": gmm_flop / (gmm_ms * 1e-3) / 1e12,
            "traffic_source": pmc_traffic_source("gmm_screen_kernel"),
            # `frac` prices the kernel against the roof at the NOMINAL 2400 MHz; the board does not hold that clock inside
            # this kernel (power): in_kernel_clock_mhz is what its own blocks measured (s_memtime against the 100 MHz
            # reference clock, jd_gmm_screen_clock), frac_at_in_kernel_clock the fraction of the roof at THAT clock
            "in_kernel_clock_mhz": screen_clock["mhz"] if screen_clock else None,
            "in_kernel_clock_blocks_sampled": screen_clock["blocks_sampled"] if screen_clock else None,
            "frac_at_in_kernel_clock": (achieved / (F16_MATRIX_PEAK_TFLOPS * screen_clock["mhz"] / 2400.0)) if screen_clock else None,
            "note": "results are bit-identical to the fp32 MFMA kernel; the fp16 product only decides which "
                    "components can NOT be the arg-max; gmm_exact also writes the gradient rows of the survivors "
                    "(the backward pass of the arg-max prior has no kernel of its own)",
        }
    elif gmm_ms:
        achieved = gmm_flop / (gmm_ms * 1e-3) / 1e12
        executed = gmm_flop_executed / (gmm_ms * 1e-3) / 1e12
        roof_gmm = {
            "kernel": "gmm_fwd_kernel", "bound": "mfma", "achieved": achieved, "peak": FP32_MATRIX_PEAK_TFLOPS,
            "unit": "TFLOP/s", "frac": achieved / FP32_MATRIX_PEAK_TFLOPS,
            "traffic": pmc_traffic_bytes(args.config, "gmm_fwd_kernel") if world == 1 and fake is None else None,
            "avg_launch_ms": gmm_ms, "launches": gmm_n, "flop_per_launch": gmm_flop,
            "executed_flop_per_launch": gmm_flop_executed, "executed_achieved": executed,
            "executed_frac": executed / FP32_MATRIX_PEAK_TFLOPS,
            "dense_equivalent_achieved": gmm_flop_dense / (gmm_ms * 1e-3) / 1e12,
        }
    # fused Poisson pass.  Stand-alone kernel (general PSFs, several components, up-sampling): 16 B/pixel (conv,
    # background, counts in; g out).  With one separable component the pass is the EPILOGUE of the forward convolution
    # (`sep_conv_kernel<.., POISSON>`): flux, exposure, background, counts in; g out = 20 B/pixel, and the convolution
    # image (4 B/pixel written + 4 read back) never exists.
    poi_ms, poi_n = avg_ms("poisson_fused")
    models_all = session.total_loss.poisson_loss.npred_models_all
    methods = sorted({m.plan.method for m in models_all})
    n_comp = len(session.components)
    # batched joint step: ONE launch covers all local datasets (jd_npred_poisson_batch_[multi_]fwd_bwd)
    batched = bool(getattr(session, "batch_joint", False))
    per_launch = len(session.local_idx) if batched else 1
    # where the Poisson pass runs, which kernel that is (prefix of its rocprofv3 name) and its algorithmic bytes per
    # (pixel, dataset) -- decided from what the library says it launches, never from the timer's name alone:
    #  * one separable / direct component: EPILOGUE of the forward convolution: flux, exposure, background, counts in,
    #    g out = 20 B; the convolution image (4 B written + 4 read back) never exists;
    #  * several separable components in a batched step (config 5): epilogue of the MULTI tile kernel: per component
    #    flux + exposure in and g out (12 B each), background + counts once (8 B): 12 C + 8;
    #  * otherwise the stand-alone kernel: conv_c, background, counts in, g_c out: 8 C + 8 (16 B at C = 1).
    fused_one = methods in (["separable"], ["direct"]) and n_comp == 1 and _hip.get_option("JD_SEP_NO_FUSION") is None
    fused_multi = methods == ["separable"] and n_comp > 1 and batched
    # (the strip-walk kernels take launches of >= 2^24 (pixel, dataset, component) triples; several components: batched only)
    walk = (methods == ["separable"] and (n_comp == 1 or (batched and n_comp <= 4))
            and models_all[0].plan.takes_walk(per_launch * n_comp))
    # frames of the local operators in the strip-walk kernels (17 / 33 taps; jd_conv_operator_walk_frame)
    frames = [models.plan.walk_frame(next(iter(models.values())).khat) for models in models_all] if walk and n_comp == 1 else []
    frame_runs = 1 + sum(1 for a, b in zip(frames, frames[1:]) if a != b) if frames else 1
    if fused_one:
        poi_px_bytes = 20
        walk_name = ("walk_mixed_kernel<4, 2>" if batched and len(set(frames)) == 2 else
                     "walk_kernel<33, 2, 2, true, true, 0, 8>" if set(frames) == {33} else
                     "walk_kernel<17, 4, 2, true, true, 0, 8>")
        poi_kernel = ("direct_conv_kernel<" if methods == ["direct"] else
                      walk_name if walk else "sep_conv_kernel<true, true, true, false>")
        poi_what = "forward convolution + Poisson pass"
    elif fused_multi:
        poi_px_bytes = 12 * n_comp + 8
        poi_kernel = "walk_multi_kernel<4, 2>" if walk else "sep_conv_kernel<true, true, true, true>"
        poi_what = f"forward convolutions of {n_comp} components + Poisson pass"
    else:
        poi_px_bytes = 8 * n_comp + 8
        poi_kernel, poi_what = "poisson_fused_kernel", "stand-alone Poisson pass"
    # MINIMUM traffic of one launch: a batched launch reads the flux image(s) it shares between its datasets ONCE
    # (round-3 verdict: 8 x (exposure + counts + background + g) + flux = 553.6 MB at config 3, not 8 x 20 B = 671 MB)
    shared_bytes = 4 * n_comp * (per_launch - 1) if (fused_one or fused_multi) and per_launch > 1 else 0
    poi_bytes = (poi_px_bytes * per_launch - shared_bytes) * H * W
    roof_poi = None
    if poi_ms:
        achieved = poi_bytes / (poi_ms * 1e-3) / 1e9
        roof_poi = {
            "kernel": f"{poi_kernel} ({poi_what})", "bound": "hbm",
            "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": pmc_traffic_bytes(args.config, poi_kernel) if world == 1 and fake is None else None,
            "traffic_source": pmc_traffic_source(poi_kernel),
            "avg_launch_ms": poi_ms, "launches": poi_n, "bytes_per_launch": poi_bytes, "datasets_per_launch": per_launch,
            "bytes_per_pixel_and_dataset": poi_px_bytes,
            "bytes_note": "minimum traffic: the flux image shared by the datasets of a launch counted once",
            "frames": {str(f): frames.count(f) for f in sorted(set(frames))} if frames else None,
        }
    poisson_in_conv = fused_one
    n_profiled = PROFILE_STEPS
    nested = ("gmm_stage", "gmm_screen", "gmm_sort", "gmm_exact")  # stage timers inside the gmm_fwd bracket
    kernel_ms_per_step = {k: (v[0] / n_profiled) for k, v in prof.items() if v[1] and k not in nested}
    dominant = max(kernel_ms_per_step, key=kernel_ms_per_step.get) if kernel_ms_per_step else None
    roofline = roof_poi if dominant == "poisson_fused" else roof_gmm

    # PSF array sizes of the observations as SURVEY.md section 8(d) gives them (17x17; 33x33 for sigma >= 3)
    from jolideco_amd.data import psf_shape

    psf_shapes = ["%dx%d" % psf_shape(1.5 + 0.25 * i) for i in range(n_obs)]
    runs = []
    for shp in psf_shapes:
        if runs and runs[-1][0] == shp:
            runs[-1][1] += 1
        else:
            runs.append([shp, 1])
    psf_sizes = " + ".join(f"{n} x {shp}" for shp, n in runs)
    out = {
        "metric": ("MAP iters/sec at 2048x2048, 8-obs joint fit" if args.config == "c3" else f"MAP iters/sec ({args.config})")
        + (f" [TUNING: rank {args.rank} of {args.shard_of}, no collective]" if args.shard_of > 1 else ""),
        "value": args.steps / elapsed,
        "unit": "iters/s",
        "n_gpus": world,
        "distributed": dist_info,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "ms_per_step_min": stats["ms_per_step_min"], "ms_per_step_max": stats["ms_per_step_max"],
        # time the HOST needs to issue one step (python + ctypes + hipLaunch / hipGraphLaunch, and at N > 1 the
        # torch.distributed calls): over the first ENQUEUE_BURST steps of a timed region, issued into an empty queue;
        # median region, MAX over the ranks.  A step cannot be faster than this: where it approaches ms_per_step the run
        # is host bound
        "host_enqueue_ms_per_step": host_enqueue_ms,
        # the timed steps were replayed from captured hipGraphs (device-resident step scalars, jolideco_amd/core.py)
        "epochs_replayed_from_graphs": replayed,
        # what the session's "auto" policy decided after its probe epochs (jolideco_amd/core.py: captured epochs only
        # where the host bounds the fit)
        "graph_policy": graph_policy,
        "repeats": len(times), "timing": "median of `repeats` regions of `steps` steps, each bracketed by barrier + synchronize",
        "settle": settled,
        "clock_mhz": clock_mhz,
        "kernel_profile_steps": PROFILE_STEPS,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": ("c1: 128x128, one point-source observation (1000 counts at the centre, Gaussian PSF sigma 3 on 17x17, "
                         "exposure 1, background 2), uniform prior, the reference's sequential loop (one step + one trace "
                         "evaluation per epoch), Adam lr 0.1" if args.config == "c1" else
                         f"{args.config}: {H}x{W}, {n_obs} observations (Gaussian PSFs sigma = 1.5 + 0.25 i on {psf_sizes}, "
                         f"varying exposure/background), GMM patch prior 8x8 stride 4 K={K}, joint fit, Adam lr 0.1"),
            "psf_shapes": psf_shapes,
            "global_observations": n_obs,
            "sharding": (f"observations over {world} rank(s) by estimated cost (longest processing time first), prior by patch "
                         "rows in shares that even the ranks out; per step 1 all-reduce of the "
                         "likelihood gradient started before the prior and overlapped with it + 1 all-gather of the prior "
                         "bands" if os.environ.get("JOLIDECO_DIST_OVERLAP", "1") != "0" else
                         f"observations over {world} rank(s) by estimated cost, prior by patch rows, 1 all-reduce/step")
            if world > 1 else "single GPU",
        },
        # numerics of the run itself: the loss scalars of the last timed step [dataset losses | log-priors] -- the same
        # for any number of ranks (tests/test_gpu_distributed.py compares a 2-rank run with a single process)
        "check": {"epochs_run": epochs_run, "scalars_last_step": [float(v) for v in scal]},
        "roofline": roofline,
        "roofline_poisson": roof_poi,
        # SURVEY section 8(d)'s own definition for the GMM prior: Np K (2 D^2 + 4 D) fp32 flop over the duration of the
        # whole forward pass, against the fp32 matrix / vector roof.  Above 1 on the screened path: the fp16 screen
        # replaces the dense fp32 evaluation by a bound + ~1.2 exact evaluations per patch (same bits out).
        "roofline_section8d": None if not gmm_ms else {
            "kernel": "GMM prior forward pass (all stages)", "bound": "mfma", "unit": "TFLOP/s",
            "achieved": gmm_flop_dense / (gmm_ms * 1e-3) / 1e12, "peak": FP32_MATRIX_PEAK_TFLOPS,
            "frac": gmm_flop_dense / (gmm_ms * 1e-3) / 1e12 / FP32_MATRIX_PEAK_TFLOPS,
            "flop_per_launch": gmm_flop_dense, "forward_pass_ms": gmm_ms,
            "note": "dense-equivalent fp32 flop of section 8(d); > 1 means the work was avoided, not executed",
        },
        "kernel_ms_per_step": kernel_ms_per_step,
        "dominant_kernel": dominant,
    }
    if comm_ms is not None:
        # rank 0's compute stream, per step: time between enqueueing the wait for the all-reduce that was started before
        # the prior and its completion (what the overlap did NOT hide), the all-gather of the prior bands, or the one
        # blocking all-reduce of the non-overlapped schedule
        out["comm_ms_per_step"] = comm_ms
    # which convolution the fit used: Gaussian PSFs are rank 1, so "auto" takes the separable kernel
    out["config"]["conv_method"] = "+".join(methods)
    # convolution kernel, HBM bound: forward reads flux + exposure and writes the convolution (12 B/pixel), the
    # adjoint reads g + exposure + the gradient accumulator and writes it back (16 B/pixel): 14 B/pixel on average
    conv_key = "sep_conv" if "sep_conv" in kernel_ms_per_step else "direct_conv" if "direct_conv" in kernel_ms_per_step else None
    if conv_key:
        conv_ms, conv_n = avg_ms(conv_key)
        # (with the Poisson pass fused into the forward launch only the adjoint carries this timer: 16 B/pixel)
        # a batched adjoint reads g + exposure per dataset and reads / writes the gradient once: (8 n + 8) B/pixel
        conv_bytes = ((8 * per_launch + 8) if ((poisson_in_conv or fused_multi) and per_launch > 1) else 16 if poisson_in_conv else 14) * H * W
        batched_adjoint_runs = frame_runs if poisson_in_conv and walk and batched and per_launch > 1 else 1
        if batched_adjoint_runs > 1:
            # operators of both frames: one launch per run of consecutive datasets of one frame, each reads and writes the
            # gradient image; `avg_launch_ms` is then the time of ALL the step's adjoint launches
            conv_bytes += 8 * (batched_adjoint_runs - 1) * H * W
            conv_ms = conv_ms * batched_adjoint_runs
        # several components, strip-walk kernels: ONE launch adds up the datasets of every component (blocks of one wave
        # per dataset, up to 16; walk_conv_adjoint_batch_all)
        adjoint_all = fused_multi and walk and 2 <= per_launch <= 16
        if adjoint_all:
            conv_bytes *= n_comp
        achieved = conv_bytes / (conv_ms * 1e-3) / 1e9
        conv_traffic = None
        if world == 1 and fake is None:
            if conv_key != "sep_conv":
                names = ("direct_conv_kernel",)
            elif adjoint_all:
                names = ("walk_kernel<17, 2, 2, false, false, 6, 16>",) if per_launch > 8 else ()
            elif poisson_in_conv and walk and batched_adjoint_runs > 1:
                names = ()  # (several kernels per step: see profiles/<round>/pmc_hbm_traffic.csv)
            elif poisson_in_conv and walk:
                names = (("walk_kernel<17, 4, 3, false, false, 6, 8>",) if per_launch >= 6 else
                         ("walk_kernel<17, 4, 2, false, false, 0, 8>",))
            elif poisson_in_conv:
                names = ("sep_conv_kernel<true, false, false",)
            else:
                names = ("sep_conv_kernel<true, true, false", "sep_conv_kernel<true, false, false")
            parts = [pmc_traffic_bytes(args.config, name) for name in names]
            conv_traffic = sum(parts) / len(parts) if parts and all(p is not None for p in parts) else None
        out["roofline_conv"] = {
            "kernel": ("walk_kernel (adjoint)" if conv_key == "sep_conv" and walk else
                       _hip.lib().jd_kernel_name(_hip.KERNEL_IDS[conv_key]).decode()), "bound": "hbm",
            "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": conv_traffic, "avg_launch_ms": conv_ms, "launches": conv_n, "bytes_per_launch": conv_bytes,
            "datasets_per_launch": per_launch, "launches_per_step": batched_adjoint_runs,
        }
    # The same fit with the PSFs treated as general (not low-rank) kernels -- what an instrument PSF that is not a sum of
    # <= 3 outer products gets: the MFMA Toeplitz convolution -- and through rocFFT (R2C, k-space multiply, C2R: the
    # path BASELINE.json's north star names; the default for PSFs larger than 33x33).  Reported next to the headline,
    # never as it.
    side_repeats = min(max(args.repeats, 1), 3)

    def conv_method_run(method, note):
        log(f"{method}-convolution run (JOLIDECO_CONV_METHOD={method})")
        previous = os.environ.get("JOLIDECO_CONV_METHOD")
        os.environ["JOLIDECO_CONV_METHOD"] = method
        try:
            other = build_session(args.config, device)
            for _ in range(args.warmup):
                other.epoch()
            warm_policy(other)
            torch.cuda.synchronize(device)
            side = region_stats(timed_regions(other, args.steps, side_repeats, device, dist_ctx), args.steps)
            prof_other = profile_phase(other, device, n_obs)
            used = sorted({m.plan.method for m in other.total_loss.poisson_loss.npred_models_all})
            plan_info = [(m.plan.kh, m.plan.kw, bool(m.plan.native_fft))
                         for mm in other.total_loss.poisson_loss.npred_models_all for m in mm.values()]
            result = {
                "value": side["value"], "unit": "iters/s", "ms_per_step": side["ms_per_step"],
                "conv_method": "+".join(used), "note": note,
                "kernel_ms_per_step": {k: v[0] / n_profiled for k, v in prof_other.items() if v[1] and k not in nested},
                "plans": plan_info,
            }
            del other
            return result, prof_other
        finally:
            if previous is None:
                os.environ.pop("JOLIDECO_CONV_METHOD", None)
            else:
                os.environ["JOLIDECO_CONV_METHOD"] = previous

    if world == 1 and fake is None and "separable" in methods and not args.no_general_psf:
        out["general_psf"], _ = conv_method_run(
            "general", "same workload with the PSFs convolved as general kernels, each by the method \"auto\" gives a general "
                       "PSF of its size: 17x17 MFMA Toeplitz convolution (fp16 x 3 split operands, Poisson pass in the forward "
                       "launch's epilogue), 33x33 native FFT convolution")
        out["direct_psf"], _ = conv_method_run(
            "direct", "same workload with every PSF through the MFMA Toeplitz convolution (the 33x33 ones included: rounds "
                      "1-3's general_psf)")
        out["direct_psf"].pop("plans", None)
        out["fft_psf"], prof_fft = conv_method_run(
            "fft", "same workload through the FFT path (native FFT convolution, csrc/fftnative.hip): per observation rows, "
                   "columns (FFT x kernel spectrum x inverse FFT), rows^-1 + Poisson pass + rows of g, columns, rows^-1 + "
                   "adjoint epilogue; the timers fft_r2c / cmul / fft_c2r carry the rows / columns / rows^-1 launches")
        # The native FFT convolution against the HBM roof: algorithmic bytes of the FIVE launches of one observation's
        # likelihood step (rows: flux + exposure in, row spectra out; columns: spectra + kernel spectrum in, spectra out,
        # twice; rows^-1 + Poisson + rows of g: spectra + background + counts in, spectra out; rows^-1 + adjoint epilogue:
        # spectra + exposure + gradient in, gradient out) over their summed duration.
        plan_list = out["fft_psf"].pop("plans")
        out["general_psf"].pop("plans", None)
        plans = sorted(set(plan_list))
        out["fft_psf"]["native_fft"] = all(p[2] for p in plans)

        def fft_length(n, with_three=True):
            best = None
            for odd in (1, 3, 9) if with_three else (1, 9):
                m = 8
                while m * odd < max(n, 32):
                    m *= 2
                best = m * odd if best is None or m * odd < best else best
            return best

        if out["fft_psf"]["native_fft"] and n_comp == 1:
            per_obs_bytes = 0.0
            for kh, kw, _ in plans:
                n_with = sum(1 for q in plan_list if q[:2] == (kh, kw))
                nx, ny, hh = fft_length(W + max((kw - 1) // 2, kw - 1 - (kw - 1) // 2)), fft_length(H // 2 + kh - 1, False), H // 2
                spec, kept, khat_b, img = hh * nx * 8, (hh + kh - 1) * nx * 8, nx * ny * 8, 4 * H * W
                per_obs = (2 * img + spec) + 2 * (spec + khat_b + kept) + (kept + 2 * img + spec) + (kept + 2 * img + img)
                per_obs_bytes += per_obs * n_with
            per_obs_bytes /= max(len(plan_list), 1)
            k = out["fft_psf"]["kernel_ms_per_step"]
            ms_obs = sum(k.get(name, 0.0) for name in ("fft_r2c", "cmul", "poisson_fused", "fft_c2r")) / max(len(plan_list), 1)
            achieved = per_obs_bytes / (ms_obs * 1e-3) / 1e9
            out["roofline_fft"] = {
                "kernel": "native FFT convolution: fftn_rows_fwd + 2 x fftn_cols + fftn_rows_poisson + fftn_rows_inv<adjoint> "
                          "(one observation's likelihood step)", "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None, "ms_per_observation": ms_obs,
                "bytes_per_observation": per_obs_bytes,
                "launch_ms_per_step": {name: k.get(name) for name in ("fft_r2c", "cmul", "poisson_fused", "fft_c2r")},
                "note": "PMC traffic of these kernels: profiles/r05/pmc_hbm_traffic.csv rows c3fft",
            }
    if world == 1 and fake is None and args.config == "c3" and not args.no_general_psf:
        # any image size takes the native FFT path since round 5 (odd H: a lower half one row short; W % 4 != 0: rows at
        # 4-byte alignment): c3's workload at 2047 x 2050 through the FFT path, native against rocFFT (JD_FFT_NATIVE=0:
        # the fallback such sizes took until round 4 -- un-batched plans, one observation after the other)
        log("odd-size FFT run (2047 x 2050)")
        odd = {}
        previous = os.environ.get("JOLIDECO_CONV_METHOD")
        os.environ["JOLIDECO_CONV_METHOD"] = "fft"
        try:
            from jolideco_amd.ops import ConvPlan

            for label, native in (("native", None), ("rocfft", 0)):
                # (plans are cached by geometry and the switch is read when a plan is created: a cache of its own per run;
                # the plans of the sessions that are still alive stay where they are)
                saved_cache, ConvPlan._cache = ConvPlan._cache, {}
                with _hip.options(JD_FFT_NATIVE=native):
                    other = build_session("c3odd", device)
                    for _ in range(max(args.warmup, 6)):
                        other.epoch()
                    warm_policy(other)
                    torch.cuda.synchronize(device)
                    side = region_stats(timed_regions(other, args.steps, side_repeats, device, dist_ctx), args.steps)
                    plans = {m.plan for mm in other.total_loss.poisson_loss.npred_models_all for m in mm.values()}
                    odd[label] = {"value": side["value"], "unit": "iters/s", "ms_per_step": side["ms_per_step"],
                                  "native_fft": all(bool(p.native_fft) for p in plans), "batched_joint_step": bool(other.batch_joint)}
                    del other, plans
                gc.collect()
                for plan in ConvPlan._cache.values():
                    plan.close()
                ConvPlan._cache = saved_cache
                torch.cuda.empty_cache()
        finally:
            if previous is None:
                os.environ.pop("JOLIDECO_CONV_METHOD", None)
            else:
                os.environ["JOLIDECO_CONV_METHOD"] = previous
        odd["workload"] = "c3's observations and prior on a 2047 x 2050 image (odd rows, width % 4 == 2), JOLIDECO_CONV_METHOD=fft"
        out["odd_size_fft"] = odd
    # The same fit with the GMM arg-max evaluated by the dense fp32 MFMA kernel for every (patch, component) pair
    # (JD_GMM_SCREEN=0; bit-identical results): reported next to the headline for whoever wants the number without the
    # fp16 screen.
    if world == 1 and fake is None and roof_gmm and roof_gmm["kernel"] == "gmm_screen_kernel" and not args.no_general_psf:
        log("dense-GMM run (option JD_GMM_SCREEN=0)")
        with _hip.options(JD_GMM_SCREEN=0):
            for _ in range(args.warmup):
                session.epoch()
            warm_policy(session)
            torch.cuda.synchronize(device)
            side = region_stats(timed_regions(session, args.steps, side_repeats, device, dist_ctx), args.steps)
            out["dense_fp32_gmm"] = {
                "value": side["value"], "unit": "iters/s", "ms_per_step": side["ms_per_step"],
                "note": "same workload, GMM arg-max by the dense fp32 MFMA kernel (option JD_GMM_SCREEN=0)",
            }
    # The reference's own loop (fit_mode="sequential", jolideco/core.py:209-247): one optimizer step per dataset, each
    # with a full prior evaluation, then the per-epoch trace on all datasets -- SURVEY.md section 8(d) asks for both rates.
    if world == 1 and fake is None and not args.no_general_psf:
        log("sequential-mode run")
        seq = build_session(args.config, device, fit_mode="sequential")
        n_epochs = max(args.steps // 10, 2)
        for _ in range(2):
            seq.epoch()
        torch.cuda.synchronize(device)
        dt = float(np.median(timed_regions(seq, n_epochs, side_repeats, device, dist_ctx)))
        out["sequential_mode"] = {
            "epochs_per_s": n_epochs / dt, "steps_per_s": n_epochs * n_obs / dt, "ms_per_epoch": 1e3 * dt / n_epochs,
            "epochs": n_epochs,
            "note": "reference-exact trajectory: one Adam step per observation (each with the full prior) + the trace "
                    "evaluation of every observation per epoch",
        }
        del seq
    if world == 1 and fake is None and not args.no_general_psf:
        # the same steps REPLAYED from captured hipGraphs (JOLIDECO_GRAPH=1; device-resident step scalars): what the host
        # then pays per step, and what the device pays for the replay
        log("graph-replay run (JOLIDECO_GRAPH=1)")
        previous = os.environ.get("JOLIDECO_GRAPH")
        os.environ["JOLIDECO_GRAPH"] = "1"
        try:
            other = build_session(args.config, device, fit_mode="sequential" if args.config == "c1" else "joint")
            for _ in range(max(args.warmup, 10)):
                other.epoch()
            warm_policy(other)
            torch.cuda.synchronize(device)
            side = region_stats(timed_regions(other, args.steps, side_repeats, device, dist_ctx), args.steps)
            out["graph_replay"] = {
                "value": side["value"], "unit": "iters/s", "ms_per_step": side["ms_per_step"],
                "host_enqueue_ms_per_step": 1e3 * float(np.median(HOST_ENQUEUE[-1])) / args.steps,
                "epochs_replayed_from_graphs": bool(other._graphs),
                "note": "same workload, every epoch replayed from a captured hipGraph (the default policy captures only fits the "
                        "host bounds)",
            }
            del other
        finally:
            if previous is None:
                os.environ.pop("JOLIDECO_GRAPH", None)
            else:
                os.environ["JOLIDECO_GRAPH"] = previous
    if world == 1 and fake is None and args.config == "c5" and not args.no_general_psf:
        log("c5 with one PSF per dataset for both components (evaluated as their sum)")
        other = build_session("c5", device, shared_psf=True)
        merged = bool(other.total_loss.poisson_loss.mergeable([li for _, li in other.local_idx]) and other.flux_nonneg)
        for _ in range(max(args.warmup, 10)):
            other.epoch()
        warm_policy(other)
        torch.cuda.synchronize(device)
        side = region_stats(timed_regions(other, args.steps, side_repeats, device, dist_ctx), args.steps)
        out["shared_psf"] = {
            "value": side["value"], "unit": "iters/s", "ms_per_step": side["ms_per_step"], "components_evaluated_as_their_sum": merged,
            "note": "the same fit with ONE PSF array per dataset for both components (the reference's default when `psf` is no "
                    "dict): one forward model and one adjoint per dataset (jd_sum_images / jd_copy_image_to)",
        }
        del other
    if world == 1 and fake is None and args.config == "c3" and not args.no_general_psf and not args.no_c6:
        log("c6 side run (calibrations + up-sampling x2 + general 65x65 PSFs)")
        del session
        gc.collect()
        torch.cuda.empty_cache()
        out["c6_chandra_like"] = c6_run(device, dist_ctx)
    log("gpu result: " + json.dumps(out))
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.config)
    print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
