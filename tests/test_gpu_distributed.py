"""GPU, world_size 2: the sharded joint fit through the real `init_from_env` / `FitSession` /
`torch.distributed.all_reduce` path with the HIP kernels.  A GPU box has one device, so both ranks
share cuda:0 and the collective runs over gloo (JOLIDECO_DIST_BACKEND=gloo; device tensors are
staged through the host).  The production backend (RCCL) differs only in the transport of the one
all-reduce per step."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

REPO = Path(__file__).resolve().parent.parent
for p in (str(REPO), str(REPO / "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world_size, port, out_dir, overlap):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world_size),
                      LOCAL_RANK=str(rank), JOLIDECO_DIST_BACKEND="gloo", JOLIDECO_DIST_OVERLAP=overlap)
    from conftest import unpack_datasets
    from jolideco_amd import GMMPatchPrior, MAPDeconvolver, SpatialFluxComponent
    from jolideco_amd.distributed import init_from_env
    from jolideco_amd.priors.patches import GaussianMixtureModel, GaussianMixtureModelMeta

    ctx = init_from_env()
    assert ctx.world_size == world_size and ctx.rank == rank
    j = dict(np.load(REPO / "tests" / "golden" / "joint_multi.npz"))
    datasets = unpack_datasets(j, "joint/data/")
    gmm = GaussianMixtureModel.from_numpy(j["gmm/means"], j["gmm/covariances"], j["gmm/weights"],
                                          meta=GaussianMixtureModelMeta(stride=4))
    comp = SpatialFluxComponent.from_numpy(flux=j["joint/flux_init"], prior=GMMPatchPrior(gmm=gmm))
    res = MAPDeconvolver(n_epochs=12, display_progress=False, device="cuda:0", fit_mode="joint").run(
        datasets, components=comp
    )
    np.savez(Path(out_dir) / f"rank{rank}.npz", flux=res.flux_total,
             **{f"trace/{n}": np.asarray(res.trace_loss[n]) for n in res.trace_loss.colnames if n != "filename"})
    torch.distributed.destroy_process_group()


def _calibrated_joint_fit(n_epochs=8):
    """The joint fit of the calibration fixture's three observations (trained sub-pixel shift + background norm, a zero
    shift that must stay zero, a frozen calibration), GMM patch prior: result, calibration initial values"""
    from conftest import unpack_datasets
    from jolideco_amd import GMMPatchPrior, MAPDeconvolver, NPredCalibration, NPredCalibrations, SpatialFluxComponent
    from jolideco_amd.priors.patches import GaussianMixtureModel, GaussianMixtureModelMeta

    c = dict(np.load(REPO / "tests" / "golden" / "calibration.npz"))
    datasets = unpack_datasets(c, "u1/data/")
    gmm = GaussianMixtureModel.from_numpy(c["u1/gmm_means"], c["u1/gmm_covariances"], c["u1/gmm_weights"],
                                          meta=GaussianMixtureModelMeta(stride=4))
    cals = NPredCalibrations()
    for name in datasets:
        sx, sy, norm, psf_scale, frozen = (float(v) for v in c[f"u1/cal_init/{name}"])
        cals[name] = NPredCalibration(shift_x=sx, shift_y=sy, background_norm=norm, psf_scale=psf_scale, frozen=bool(frozen))
    comp = SpatialFluxComponent.from_numpy(flux=c["u1/flux_init"], prior=GMMPatchPrior(gmm=gmm))
    res = MAPDeconvolver(n_epochs=n_epochs, display_progress=False, device="cuda:0", fit_mode="joint").run(
        datasets, components=comp, calibrations=cals
    )
    cal_values = np.array([[d["shift_x"], d["shift_y"], d["background_norm"], d["psf_scale"]]
                           for d in (res.calibrations[name].to_dict() for name in datasets)])
    return res, cal_values, c, datasets


def _worker_calibrated(rank, world_size, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world_size),
                      LOCAL_RANK=str(rank), JOLIDECO_DIST_BACKEND="gloo", JOLIDECO_DIST_OVERLAP="1")
    from jolideco_amd.distributed import init_from_env

    ctx = init_from_env()
    assert ctx.world_size == world_size and ctx.rank == rank
    res, cal_values, _, _ = _calibrated_joint_fit()
    np.savez(Path(out_dir) / f"rank{rank}.npz", flux=res.flux_total, cal=cal_values,
             **{f"trace/{n}": np.asarray(res.trace_loss[n]) for n in res.trace_loss.colnames if n != "filename"})
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(900)
def test_two_rank_calibrated_joint_fit(tmp_path):
    """NPredCalibrations in a SHARDED joint fit (both reference examples calibrate every observation,
    examples/fermi-vela-junior.py:193-203; model: jolideco/models/npred.py:298-402, optimizer: core.py:197-204): a
    dataset's calibration -- parameters, gradients, its Adam state -- lives on the rank that owns the dataset; the
    values are gathered for the result.  Both ranks end with identical fluxes AND identical calibrations; they agree
    with the single-process fit to rounding (the prior's bands are added in another order) and with the oracle's
    joint harness (one Adam over fluxes and calibration parameters) within the north-star tolerance."""
    from conftest import rel_linf
    from oracle import cpu_ref

    mp.spawn(_worker_calibrated, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0, r1 = dict(np.load(tmp_path / "rank0.npz")), dict(np.load(tmp_path / "rank1.npz"))
    assert np.array_equal(r0["flux"], r1["flux"]) and np.array_equal(r0["cal"], r1["cal"])
    res, cal_single, c, datasets = _calibrated_joint_fit()
    assert rel_linf(r0["flux"], res.flux_total) < 2e-6
    np.testing.assert_allclose(r0["cal"], cal_single, rtol=1e-5, atol=1e-6)
    for name in res.trace_loss.colnames:
        if name != "filename":
            np.testing.assert_allclose(r0[f"trace/{name}"], np.asarray(res.trace_loss[name]), rtol=2e-5, atol=1e-6, err_msg=name)
    # the oracle: jolideco's pieces in the joint harness, calibration parameters in the same optimizer
    gmm_o = cpu_ref.GMM.from_numpy(c["u1/gmm_means"], c["u1/gmm_covariances"], c["u1/gmm_weights"], stride=4)
    cals_o = {}
    for name in datasets:
        sx, sy, norm, psf_scale, frozen = c[f"u1/cal_init/{name}"]
        cals_o[name] = cpu_ref.CalibrationRef.create(sx, sy, norm, psf_scale, bool(frozen))
    final_o, trace_o = cpu_ref.map_fit_joint(datasets, {"flux": c["u1/flux_init"]}, {"flux": cpu_ref.GMMPatchPriorRef(gmm_o)},
                                             n_epochs=8, calibrations=cals_o, upsampling_factors={"flux": 1})
    err = rel_linf(r0["flux"], final_o["flux"])
    print("2-rank calibrated joint fit vs oracle: rel Linf", err)
    assert err < 1e-5
    for i, name in enumerate(datasets):
        d = cals_o[name].to_dict()
        want = np.array([d["shift_x"], d["shift_y"], d["background_norm"], d["psf_scale"]])
        np.testing.assert_allclose(r0["cal"][i], want, rtol=2e-4, atol=2e-5, err_msg=name)
    assert r0["cal"][1][0] == 0.0  # the zero shift of o1 never moves (utils/torch.py:211)
    np.testing.assert_allclose(r0["trace/total"], [row["total"] for row in trace_o], rtol=2e-5)


@pytest.mark.timeout(900)
@pytest.mark.parametrize("overlap", ["1", "0"])
def test_two_rank_joint_fit_matches_the_single_process_reference(tmp_path, golden, overlap):
    """overlap "1" (default): all-reduce of the likelihood gradient in flight while the prior runs, the prior's bands in
    one all-gather; "0": the prior accumulated into the flat buffer before its single all-reduce."""
    from conftest import rel_linf

    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path), overlap), nprocs=2, join=True)
    j = golden("joint_multi")
    r0, r1 = dict(np.load(tmp_path / "rank0.npz")), dict(np.load(tmp_path / "rank1.npz"))
    # replicas apply the identical update after the all-reduce: bit-identical parameters, no broadcast
    assert np.array_equal(r0["flux"], r1["flux"])
    err = rel_linf(r0["flux"], j["joint/flux_final"])
    print("2-rank joint rel Linf", err)
    assert err < 1e-5
    for key, ref in j.items():
        if key.startswith("joint/trace/"):
            np.testing.assert_allclose(r0[key[len("joint/"):]], ref, rtol=2e-5, atol=1e-6, err_msg=key)


@pytest.mark.timeout(1500)
def test_bench_two_ranks_over_gloo_reports_what_the_backend_saw(tmp_path):
    """`bench.py --gpus 2` itself, launched the way the driver launches it (torch.distributed.run, one fresh child
    process per rank, started before any GPU call of theirs), with JOLIDECO_DIST_BACKEND=gloo so that both ranks can
    share this box's single GPU.  The JSON line must report the world size the BACKEND saw and the loss scalars of
    the sharded fit must equal those of the same fit in one process."""
    import json
    import subprocess

    import bench

    steps, warmup = 3, 1
    env = dict(os.environ, JOLIDECO_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for key in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(key, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(REPO / "bench.py"), "--gpus", "2", "--steps", str(steps), "--warmup",
           str(warmup), "--no-general-psf", "--repeats", "1", "--settle-seconds", "0"]
    done = subprocess.run(cmd, cwd=REPO, env=env, capture_output=True, text=True, timeout=1400)
    assert done.returncode == 0, done.stderr[-3000:]
    lines = [ln for ln in done.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, done.stdout[-2000:]  # rank 0 prints ONE line
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == steps and out["warmup"] == warmup
    assert out["distributed"] == {"backend": "gloo", "world_size_seen_by_backend": 2, "rank": 0}
    assert out["scaling"] == "strong" and out["value"] > 0 and "cpu_baseline" not in out
    # the line says how long rank 0's stream waited for the collectives it overlaps (a SCALE run diagnoses itself)
    assert set(out["comm_ms_per_step"]) == {"all_gather_bands", "all_reduce_wait"}
    assert all(v >= 0 for v in out["comm_ms_per_step"].values())
    # the same fit in this process: identical scalars (the replicas apply the identical update after the all-reduce)
    session = bench.build_session("c3", torch.device("cuda:0"))
    for _ in range(warmup + steps):
        session.epoch()
    torch.cuda.synchronize()
    single = session.scalars.cpu().numpy()
    assert out["check"]["epochs_run"] == warmup + steps
    np.testing.assert_allclose(np.array(out["check"]["scalars_last_step"]), single, rtol=1e-5)


@pytest.mark.parametrize("world_size", [8, 3])
def test_every_rank_of_a_sharded_joint_step_in_one_process(world_size):
    """The sharded joint step as EVERY rank of an N-GPU job computes it (DistContext(rank, N, dry_run=True): the
    partition of the datasets, the band plan of the prior incl. the ragged last band, the band kernels and
    jd_add_rolled_bands -- everything but the transport), one rank after the other in this process on the bench's
    workload shape (8 observations, K = 128).  What the collectives would deliver, summed on the host in float64, must be
    the single-GPU step: gradient buffer and loss scalars.  (N = 8: one observation per rank, the driver's scaling run;
    N = 3: uneven observation counts and band heights.)  The workload is SURVEY section 8(d)'s mixed one (observations 6 and 7
    carry 33x33 PSFs): the cost-aware placement gives their ranks fewer patch rows of the prior -- bands of different heights
    in one all-gather -- and the sum must still be the single-GPU step."""
    from conftest import rel_linf
    from jolideco_amd import GMMPatchPrior, MAPDeconvolver, SpatialFluxComponent
    from jolideco_amd.data import synthetic_gmm, synthetic_observations
    from jolideco_amd.distributed import DistContext
    from jolideco_amd.priors.patches import GaussianMixtureModel, GaussianMixtureModelMeta

    shape, n_obs = (328, 512), 8  # 81 patch rows: not a multiple of 8 or 3
    datasets, _, flux_init = synthetic_observations(shape=shape, n_obs=n_obs, seed=0)
    means, covs, weights = synthetic_gmm(128, 64, seed=0)

    def one_step(dist):
        gmm = GaussianMixtureModel.from_numpy(means, covs, weights, meta=GaussianMixtureModelMeta(stride=4))
        comp = SpatialFluxComponent.from_numpy(flux=flux_init, prior=GMMPatchPrior(gmm=gmm))
        deco = MAPDeconvolver(n_epochs=1, display_progress=False, device="cuda:0", fit_mode="joint")
        session = deco.session(datasets, components=comp, dist=dist)
        session.cfg._optimizer_step = lambda states, step: None  # keep the gradient buffer, no update
        session.epoch()
        torch.cuda.synchronize()
        rows = session.band_plan[0]["rows"] if session.band_plan else None
        return (session.states[0].grad.double().cpu().numpy().copy(), session.scalars.double().cpu().numpy().copy(),
                session.priors[0].last_shifts, [g for g, _ in session.local_idx], rows)

    grad_1, scalars_1, shifts_1, local, _ = one_step(DistContext())
    assert local == list(range(n_obs))
    grad_sum, scalars_sum, owned, band_rows = np.zeros_like(grad_1), np.zeros_like(scalars_1), [], []
    for rank in range(world_size):
        grad_r, scalars_r, shifts_r, local, rows = one_step(DistContext(rank=rank, world_size=world_size, dry_run=True))
        assert shifts_r == shifts_1  # identically seeded generators: every rank rolls the image the same way
        grad_sum += grad_r
        scalars_sum += scalars_r
        owned += local
        band_rows.append(rows)
    assert sorted(owned) == list(range(n_obs))
    # the bands tile the 81 patch rows; with one observation per rank the two ranks of the 33x33 PSFs take fewer of them
    assert band_rows[0][0] == 0 and band_rows[-1][1] == 81 and all(a[1] == b[0] for a, b in zip(band_rows[:-1], band_rows[1:]))
    if world_size == 8:
        heights = [b - a for a, b in band_rows]
        assert sorted(heights)[1] < sorted(heights)[2] and sum(heights) == 81
    err = rel_linf(grad_sum, grad_1)
    print(f"{world_size} emulated ranks: gradient rel Linf {err:.2e}, scalars", np.abs(scalars_sum / scalars_1 - 1).max())
    assert err < 2e-6
    np.testing.assert_allclose(scalars_sum, scalars_1, rtol=2e-6)


@pytest.mark.parametrize("optimizer", ["adam", "sgd"])
def test_optimizer_step_inside_the_band_sum_changes_no_bit(monkeypatch, optimizer):
    """Sharded fits add the bands of the prior's gradient and apply the optimizer step in one launch
    (jd_add_rolled_bands_step).  Same bits as jd_add_rolled_bands followed by the optimizer kernel
    (JOLIDECO_NO_FUSED_STEP=1): four steps of one rank of three (dry run: everything but the transport) on a width that
    takes the fused kernel, and on one that cannot (not a multiple of 4: both runs use the two launches)."""
    from jolideco_amd import GMMPatchPrior, MAPDeconvolver, SpatialFluxComponent
    from jolideco_amd.data import synthetic_gmm, synthetic_observations
    from jolideco_amd.distributed import DistContext
    from jolideco_amd.priors.patches import GaussianMixtureModel, GaussianMixtureModelMeta

    means, covs, weights = synthetic_gmm(16, 64, seed=2)
    for shape in ((96, 132), (80, 101)):
        datasets, _, flux_init = synthetic_observations(shape=shape, n_obs=3, seed=4)
        results = {}
        for fused in (True, False):
            if fused:
                monkeypatch.delenv("JOLIDECO_NO_FUSED_STEP", raising=False)
            else:
                monkeypatch.setenv("JOLIDECO_NO_FUSED_STEP", "1")
            gmm = GaussianMixtureModel.from_numpy(means, covs, weights, meta=GaussianMixtureModelMeta(stride=4))
            comp = SpatialFluxComponent.from_numpy(flux=flux_init, prior=GMMPatchPrior(gmm=gmm))
            deco = MAPDeconvolver(n_epochs=4, display_progress=False, device="cuda:0", fit_mode="joint", optimizer_type=optimizer,
                                  learning_rate=0.1 if optimizer == "adam" else 1e-3)
            session = deco.session(datasets, components=comp, dist=DistContext(rank=1, world_size=3, dry_run=True))
            took = session._fuse_band_step(session.states[0])
            assert took == (fused and shape[1] % 4 == 0)
            for _ in range(4):
                session.epoch()
            torch.cuda.synchronize()
            results[fused] = session.states[0].flux_cur.cpu().numpy().copy()
        assert np.array_equal(results[True], results[False])
        assert not np.array_equal(results[True], flux_init.astype(np.float32))


def _worker_rccl_one_rank(rank, port, out_dir):
    """Child process: a ONE-rank "nccl" (= RCCL) process group on cuda:0, the sharded joint step forced through its
    collectives (JOLIDECO_FORCE_COLLECTIVES=1), then the same fit un-sharded in the same process."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                      JOLIDECO_FORCE_COLLECTIVES="1", JOLIDECO_DIST_OVERLAP="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    os.environ.pop("JOLIDECO_DIST_BACKEND", None)
    from jolideco_amd import GMMPatchPrior, MAPDeconvolver, SpatialFluxComponent
    from jolideco_amd.data import synthetic_gmm, synthetic_observations
    from jolideco_amd.distributed import DistContext, init_from_env
    from jolideco_amd.priors.patches import GaussianMixtureModel, GaussianMixtureModelMeta

    ctx = init_from_env()
    backend = torch.distributed.get_backend()
    assert backend == "nccl" and torch.distributed.get_world_size() == 1
    assert ctx.world_size == 1 and ctx.rank == 0 and ctx.sharded and ctx.force_collectives and not ctx.dry_run
    shape, n_obs, n_epochs = (328, 512), 8, 6  # config 3 in small: 17x17 and 33x33 PSFs, 81 patch rows
    datasets, _, flux_init = synthetic_observations(shape=shape, n_obs=n_obs, seed=0)
    means, covs, weights = synthetic_gmm(32, 64, seed=0)

    def fit(dist, events):
        gmm = GaussianMixtureModel.from_numpy(means, covs, weights, meta=GaussianMixtureModelMeta(stride=4))
        comp = SpatialFluxComponent.from_numpy(flux=flux_init, prior=GMMPatchPrior(gmm=gmm))
        deco = MAPDeconvolver(n_epochs=n_epochs, display_progress=False, device="cuda:0", fit_mode="joint")
        session = deco.session(datasets, components=comp, dist=dist)
        if events:
            session.comm_events = []
        scalars = []
        for _ in range(n_epochs):
            session.epoch()
            scalars.append(session.scalars.clone())
        torch.cuda.synchronize()
        comm = session.comm_times_ms() if events else {}
        return session, session.states[0].flux_cur.cpu().numpy().copy(), torch.stack(scalars).cpu().numpy(), comm

    sharded, flux_s, scalars_s, comm = fit(ctx, True)
    # the sharded schedule really ran: band plan over one rank, band sum + optimizer step in one launch, both collectives
    assert sharded.band_plan and len(sharded.band_plan[0]["y_ranges"]) == 1 and sharded._fuse_band_step(sharded.states[0])
    assert not sharded._fuse_step(sharded.states[0], sharded.priors[0])
    assert set(comm) == {"all_gather_bands", "all_reduce_wait"}
    plain, flux_p, scalars_p, _ = fit(DistContext(), False)
    assert plain.band_plan is None and plain._fuse_step(plain.states[0], plain.priors[0])
    # MAPDeconvolver.run picks the forced context up from the process group + environment
    gmm = GaussianMixtureModel.from_numpy(means, covs, weights, meta=GaussianMixtureModelMeta(stride=4))
    comp = SpatialFluxComponent.from_numpy(flux=flux_init, prior=GMMPatchPrior(gmm=gmm))
    res = MAPDeconvolver(n_epochs=n_epochs, display_progress=False, device="cuda:0", fit_mode="joint").run(datasets, components=comp)
    np.savez(Path(out_dir) / "rccl.npz", flux_sharded=flux_s, flux_plain=flux_p, scalars_sharded=scalars_s,
             scalars_plain=scalars_p, flux_run=res.flux_total, flux_init=flux_init.astype(np.float32),
             comm=np.array([comm["all_gather_bands"], comm["all_reduce_wait"]]))
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(900)
def test_rccl_one_rank_sharded_step_equals_the_unsharded_step_bit_for_bit(tmp_path):
    """RCCL on the hardware there is: `init_process_group("nccl", world_size=1, device_id=cuda:0)` in a child process, and
    `FitSession`'s SHARDED joint step forced through the collectives (no `world_size == 1` short-cut): the asynchronous
    all-reduce of the flat [gradient | scalars] buffer on RCCL's stream with `wait()` ordering the compute stream behind
    it, `all_gather_into_tensor` on the device bands of the prior, the band sum + optimizer step, the set-up exchanges
    (`assert_same_on_all_ranks` on device tensors), HSA_ENABLE_IPC_MODE_LEGACY=0.  With one rank the collectives are
    identities, so six steps must give the un-sharded fit BIT FOR BIT: fluxes and every loss scalar of every step."""
    mp.spawn(_worker_rccl_one_rank, args=(_free_port(), str(tmp_path)), nprocs=1, join=True)
    r = dict(np.load(tmp_path / "rccl.npz"))
    assert np.array_equal(r["flux_sharded"], r["flux_plain"])
    assert np.array_equal(r["scalars_sharded"], r["scalars_plain"])
    assert np.array_equal(r["flux_run"], r["flux_sharded"])
    assert not np.array_equal(r["flux_sharded"], r["flux_init"]) and np.all(np.isfinite(r["scalars_sharded"]))
    assert np.all(r["comm"] >= 0)
