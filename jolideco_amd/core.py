"""MAP deconvolver driver (reference: jolideco/core.py:46-282).

`MAPDeconvolver.run()` keeps the reference's constructor, call signature, result object and loss
trace.  Two fit modes:

* ``fit_mode="sequential"`` (default) -- the reference's semantics, exactly: one optimizer step per
  dataset on ``L_d - beta * logprior / n_datasets`` (core.py:214-229), a fresh cycle-spin draw per
  prior evaluation, and one no-grad trace row per epoch evaluated on the STALE fluxes of the last
  step (core.py:247, loss.py:212-250).  Steps depend on each other through the optimizer, so with
  several GPUs every rank runs the same replica (no collective).
* ``fit_mode="joint"`` -- one optimizer step per epoch on ``sum_d L_d - beta * logprior``.  With
  `torch.distributed` initialised, datasets are sharded round-robin over the ranks, the GMM prior
  is sharded by patch rows, and ONE all-reduce (RCCL over xGMI) sums the flux gradients and the
  loss scalars per step; every rank then applies the identical parameter update.

All arithmetic of the step runs in libjolideco_hip.so; there is no CPU fallback.
"""
import copy
import ctypes
import os
import logging
from pathlib import Path

import numpy as np
import torch

from . import _hip
from ._hip import check, ptr, ptr_array, stream_ptr
from .distributed import DistContext
from .loss import TotalLoss
from .models import FluxComponents, SpatialFluxComponent
from .ops import adam_bias_terms
from .utils.torch import TORCH_DEFAULT_DEVICE

log = logging.getLogger(__name__)

__all__ = ["MAPDeconvolver", "MAPDeconvolverResult", "FitSession"]

OPTIMIZER = ("adam", "sgd")
FIT_MODES = ("sequential", "joint")


class _ComponentState:
    """Device buffers of one flux component during a fit: theta (the nn.Parameter's storage),
    two flux buffers (current / previous, so the trace can see the pre-step flux), the flux
    gradient accumulator and the Adam moments."""

    def __init__(self, name, component, grad):
        self.name = name
        self.component = component
        theta = component._flux_upsampled.data
        if not theta.is_cuda:
            raise RuntimeError("components must be on a HIP device: jolideco_amd has no CPU path")
        self.theta = theta.reshape(theta.shape[-2:])
        self.shape = tuple(self.theta.shape)
        self.mask = None
        if component.mask is not None:
            self.mask = component.mask.to(theta.device, torch.float32).reshape(self.shape).contiguous()
        self.flux = [torch.empty_like(self.theta), torch.empty_like(self.theta)]
        self.cur = 0
        self.grad = grad.reshape(self.shape)  # view into the flat communication buffer
        self.exp_avg = torch.zeros_like(self.theta)
        self.exp_avg_sq = torch.zeros_like(self.theta)
        self.frozen = component.frozen
        self.use_log_flux = bool(component.use_log_flux)
        # reference quirk: with use_log_flux=False and no mask `flux_upsampled` IS the
        # parameter tensor, which the optimizer updates in place, so the per-epoch trace sees the POST-step flux
        self.trace_sees_current = (not self.use_log_flux) and self.mask is None
        check(
            _hip.lib().jd_flux_from_theta(
                ptr(self.theta), ptr(self.mask), ptr(self.flux[0]), self.theta.numel(), int(self.use_log_flux),
                stream_ptr(theta.device),
            )
        )

    @property
    def flux_trace(self):
        """Flux the reference's `append_trace` evaluates (jolideco/core.py:247)."""
        return self.flux_cur if self.trace_sees_current else self.flux_prev

    @property
    def flux_cur(self):
        return self.flux[self.cur]

    @property
    def flux_prev(self):
        return self.flux[1 - self.cur]


class MAPDeconvolver:
    """Maximum A-Posteriori deconvolver

    Attributes
    ----------
    n_epochs : int
        Number of epochs to train
    beta : float
        Scale factor for the prior.
    learning_rate : float
        Learning rate
    compute_error : bool
        Whether to compute flux errors after the fit (`TotalLoss.fluxes_error`)
    stop_early : bool
        Stop once the validation loss stops improving (average over the last n epochs).
    stop_early_n_average : int
        Number of epochs to average over.
    device : str
        HIP device, "cuda" or "cuda:<i>".
    display_progress : bool
        Show a tqdm progress bar (forces one device->host sync per epoch).
    optimizer_type : {"adam", "sgd"}
    optimizer_kwargs : dict
        ``lr``, and for Adam ``betas`` and ``eps``.
    checkpoint_path : str
        Directory for per-epoch checkpoints (``checkpoint-epoch-<n>.asdf``, the reference's name and format,
        core.py:77,234-245).  Forces one device->host copy per epoch.
    fit_mode : {"sequential", "joint"}
        See the module docstring.
    """

    _default_flux_component = "flux"
    _default_checkpoint_filename = "checkpoint-epoch-{epoch}.asdf"

    def __init__(
        self,
        n_epochs=1_000,
        beta=1,
        learning_rate=0.1,
        compute_error=False,
        stop_early=False,
        stop_early_n_average=10,
        device=TORCH_DEFAULT_DEVICE,
        display_progress=True,
        optimizer_type="adam",
        optimizer_kwargs=None,
        checkpoint_path=None,
        fit_mode="sequential",
    ):
        self.n_epochs = n_epochs
        self.beta = beta
        self.learning_rate = learning_rate
        self.compute_error = compute_error
        self.stop_early = stop_early
        self.stop_early_n_average = stop_early_n_average
        self.display_progress = display_progress
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError(
                f"jolideco_amd runs on an AMD GPU through libjolideco_hip.so only; got device {device}. "
                "There is no CPU fallback."
            )
        self.device = device
        if optimizer_type not in OPTIMIZER:
            raise ValueError(f"Unknown optimizer: {optimizer_type}, must be one of {OPTIMIZER}")
        self.optimizer_type = optimizer_type
        self.optimizer_kwargs = dict(optimizer_kwargs or {})
        self.optimizer_kwargs.setdefault("lr", self.learning_rate)
        unknown = set(self.optimizer_kwargs) - {"lr", "betas", "eps"}
        if unknown:
            raise NotImplementedError(f"optimizer_kwargs {sorted(unknown)} are not implemented in jolideco_amd")
        if checkpoint_path is not None:
            checkpoint_path = Path(checkpoint_path)
            checkpoint_path.mkdir(exist_ok=True, parents=True)
        self.checkpoint_path = checkpoint_path
        if fit_mode not in FIT_MODES:
            raise ValueError(f"Unknown fit_mode: {fit_mode}, must be one of {FIT_MODES}")
        self.fit_mode = fit_mode

    def to_dict(self):
        data = {k: v for k, v in self.__dict__.items() if k not in ("optimizer_kwargs",)}
        data["device"] = str(self.device)
        data["checkpoint_path"] = str(self.checkpoint_path)
        return data

    def __str__(self):
        lines = [self.__class__.__name__, "-" * len(self.__class__.__name__), ""]
        lines += [f"  {k:22s}: {v}" for k, v in self.to_dict().items()]
        return "\n".join(lines)

    # ------------------------------------------------------------------------------------------
    def _step_args(self, st, step, bias_dev=None):
        """`_hip.Step` of component state ``st`` for optimizer step number ``step``: what a prior needs to apply the
        update in the epilogue of its own last kernel (`device_fwd_bwd_step`).  ``bias_dev``: device tensor [step_size,
        bias2_sqrt] the kernel reads instead of the by-value terms (planned epochs)."""
        lr = self.optimizer_kwargs["lr"]
        args = _hip.Step()
        args.theta, args.flux_in, args.flux_out = st.theta.data_ptr(), st.flux_cur.data_ptr(), st.flux[1 - st.cur].data_ptr()
        args.grad_flux = st.grad.data_ptr()
        args.mask = None if st.mask is None else st.mask.data_ptr()
        args.use_log_flux = int(st.use_log_flux)
        if self.optimizer_type == "adam":
            beta1, beta2 = self.optimizer_kwargs.get("betas", (0.9, 0.999))
            args.step_size, args.bias2_sqrt = adam_bias_terms(step, lr, beta1, beta2)
            args.beta1, args.beta2, args.one_minus_beta1, args.one_minus_beta2 = beta1, beta2, 1 - beta1, 1 - beta2
            args.eps = self.optimizer_kwargs.get("eps", 1e-8)
            args.exp_avg, args.exp_avg_sq = st.exp_avg.data_ptr(), st.exp_avg_sq.data_ptr()
            args.sgd = 0
            args.bias_dev = None if bias_dev is None else bias_dev.data_ptr()
        else:
            args.lr, args.sgd = lr, 1
        return args

    def _optimizer_step(self, states, step, stepped=()):
        """Fused chain rule + optimizer update of every non-frozen component; swaps the flux
        buffers so `flux_prev` is the flux the step was computed with.  ``stepped``: components whose prior has
        already applied the update (`device_fwd_bwd_step`): only their buffers are swapped."""
        lib = _hip.lib()
        lr = self.optimizer_kwargs["lr"]
        for ci, st in enumerate(states):
            n = st.theta.numel()
            stream = stream_ptr(st.theta.device)
            if ci in stepped:
                pass
            elif st.frozen:
                st.flux[1 - st.cur].copy_(st.flux_cur)
                st.grad.zero_()
            elif self.optimizer_type == "adam":
                beta1, beta2 = self.optimizer_kwargs.get("betas", (0.9, 0.999))
                eps = self.optimizer_kwargs.get("eps", 1e-8)
                step_size, bias2_sqrt = adam_bias_terms(step, lr, beta1, beta2)
                check(
                    lib.jd_adam_step(
                        ptr(st.theta), ptr(st.flux_cur), ptr(st.flux[1 - st.cur]), ptr(st.grad), ptr(st.exp_avg),
                        ptr(st.exp_avg_sq), ptr(st.mask), n, step_size, beta1, beta2, 1 - beta1, 1 - beta2, bias2_sqrt, eps,
                        0,  # no zeroing pass: the first gradient term of every step OVERWRITES the buffer (FitSession.epoch)
                        int(st.use_log_flux), ptr(getattr(st, "bias_dev", None)), stream,
                    )
                )
            else:
                check(
                    lib.jd_sgd_step(
                        ptr(st.theta), ptr(st.flux_cur), ptr(st.flux[1 - st.cur]), ptr(st.grad), ptr(st.mask), n, lr,
                        0, int(st.use_log_flux), stream,
                    )
                )
            st.cur = 1 - st.cur

    def run(self, datasets, datasets_validation=None, components=None, calibrations=None):
        """Run the MAP deconvolver

        Parameters
        ----------
        datasets : dict of [str, dict]
            name -> dict with "counts", "psf" (array or {component: array}), "exposure", "background".
        datasets_validation : dict of [str, dict]
            Validation datasets (trace column and early stopping only).
        components : `FluxComponents` or `SpatialFluxComponent`
        calibrations : `NPredCalibrations`
            Per-dataset calibrations (sub-pixel shift, background norm, PSF scale), trained with the
            fluxes; keys are dataset names.

        Returns
        -------
        result : `MAPDeconvolverResult`
        """
        if self.stop_early and datasets_validation is None:
            raise ValueError("Early stopping requires providing test datasets")
        if isinstance(components, SpatialFluxComponent):
            components = {self._default_flux_component: components}
        components = FluxComponents(components)
        components_init = copy.deepcopy(components)
        calibrations_init = copy.deepcopy(calibrations) if calibrations is not None else None

        _hip.lib()  # fail loudly before touching anything if the extension is missing
        dist = DistContext.current()
        with torch.cuda.device(self.device):
            return self._run(datasets, datasets_validation, components, components_init, dist, calibrations,
                             calibrations_init)

    def session(self, datasets, datasets_validation=None, components=None, dist=None, calibrations=None):
        """Set up a fit without running it: uploads the datasets, builds the FFT plans / kernel
        spectra / GMM handles and returns a `FitSession` whose ``epoch()`` enqueues one epoch of
        the fit on the current HIP stream (used by `run` and by bench.py)."""
        if isinstance(components, SpatialFluxComponent):
            components = {self._default_flux_component: components}
        components = FluxComponents(components)
        _hip.lib()
        with torch.cuda.device(self.device):
            return FitSession(self, datasets, datasets_validation, components, dist or DistContext.current(),
                              calibrations)

    def _run(self, datasets, datasets_validation, components, components_init, dist, calibrations=None,
             calibrations_init=None):
        from tqdm.auto import tqdm

        session = FitSession(self, datasets, datasets_validation, components, dist, calibrations)
        total_loss = session.total_loss
        n_d, n_c, n_val = session.n_d, session.n_c, session.n_val
        trace_dev = torch.zeros((self.n_epochs, session.scalars.numel()), dtype=torch.float32, device=self.device)
        n_epochs_run = 0
        host_rows = []
        filenames = []
        disable = not self.display_progress
        write_checkpoints = self.checkpoint_path is not None and dist.rank == 0
        with tqdm(total=self.n_epochs * len(datasets), disable=disable) as pbar:
            for epoch in range(self.n_epochs):
                pbar.set_description(f"Epoch {epoch + 1}")
                session.epoch()
                pbar.update(len(datasets))
                trace_dev[epoch].copy_(session.scalars)
                n_epochs_run = epoch + 1

                filename = ""
                if self.checkpoint_path is not None:
                    session.gather_calibrations()  # (a collective: every rank, not only the one that writes)
                if write_checkpoints:
                    # like the reference (core.py:234-245): written before this epoch's trace row exists
                    filename = self._default_checkpoint_filename.format(epoch=epoch)
                    while len(host_rows) < epoch:
                        i = len(host_rows)
                        host_rows.append(self._row(total_loss, trace_dev[i].cpu().numpy(), n_d, n_c, n_val, filenames[i]))
                    trace_so_far = total_loss.trace.copy()
                    for row in host_rows[:epoch]:
                        trace_so_far.add_row(row)
                    checkpoint = MAPDeconvolverResult(
                        config=self.to_dict(), trace_loss=trace_so_far, components=session.components,
                        calibrations=calibrations,
                    )
                    log.info(f"Writing checkpoint to {self.checkpoint_path / filename}")
                    checkpoint.write(filename=self.checkpoint_path / filename)
                filenames.append(filename)

                if self.stop_early or self.display_progress:
                    # the only per-epoch device->host sync, and only when the caller asked for it
                    while len(host_rows) < n_epochs_run:
                        i = len(host_rows)
                        host_rows.append(self._row(total_loss, trace_dev[i].cpu().numpy(), n_d, n_c, n_val, filenames[i]))
                    row = host_rows[-1]
                    if self.stop_early and n_epochs_run > self.stop_early_n_average:
                        recent = [r["datasets-validation-total"] for r in host_rows[-self.stop_early_n_average :]]
                        if row["datasets-validation-total"] > np.mean(recent):
                            break
                    pbar.set_postfix(
                        total=row["total"], datasets_total=row["datasets-total"], priors_total=row["priors-total"]
                    )

        session.gather_calibrations()
        trace = total_loss.trace
        values = trace_dev[:n_epochs_run].cpu().numpy()
        for epoch in range(n_epochs_run):
            trace.add_row(self._row(total_loss, values[epoch], n_d, n_c, n_val, filenames[epoch]))

        if self.compute_error:
            # on the fluxes of the last step, like the reference (core.py:269-271)
            stale = [st.flux_prev if session.joint else st.flux_trace for st in session.states]
            flux_errors = total_loss.fluxes_error(fluxes=[f.reshape((1, 1) + tuple(f.shape)) for f in stale])
            session.components.set_flux_errors(flux_errors=flux_errors)

        return MAPDeconvolverResult(
            config=self.to_dict(),
            components=session.components,
            components_init=components_init,
            trace_loss=trace,
            calibrations=calibrations,
            calibrations_init=calibrations_init,
            wcs=None,
        )

    @staticmethod
    def _row(total_loss, values, n_d, n_c, n_val, filename=""):
        names_d = getattr(total_loss.poisson_loss, "names_all_global", total_loss.poisson_loss.names_all)
        loss_datasets = [float(v) for v in values[:n_d]]
        loss_priors = [float(v) for v in values[n_d : n_d + n_c]]
        loss_val = [float(v) for v in values[n_d + n_c :]] if n_val else None
        # make_row iterates poisson_loss.names_all: give it the global names
        local = total_loss.poisson_loss.names_all
        total_loss.poisson_loss.names_all = names_d
        try:
            return total_loss.make_row(loss_datasets, loss_priors, filename, loss_val)
        finally:
            total_loss.poisson_loss.names_all = local


def _adam_step_many(cfg, items, cache, bias_dev=None):
    """ONE launch for the Adam steps of the small parameter tensors `items` = [(parameter, state)] that have a gradient
    (jd_adam_step_multi: each tensor with its own step count); `cache`: pointer arrays by tensor set.
    ``bias_dev`` (planned epochs, `FitSession._epoch_planned`): device tensor of 2 floats per item holding the bias terms
    of this step -- every item steps, and the kernel reads its terms from there."""
    import ctypes

    lr = cfg.optimizer_kwargs["lr"]
    beta1, beta2 = cfg.optimizer_kwargs.get("betas", (0.9, 0.999))
    todo = []
    for p, st in items:
        if p.grad is None:
            if bias_dev is not None:  # (the plan listed the parameters that had a gradient after the previous epoch)
                raise RuntimeError("planned epoch: a calibration parameter lost its gradient between two epochs")
            continue
        st["step"] += 1
        todo.append((p, st, adam_bias_terms(st["step"], lr, beta1, beta2)))
    lib = _hip.lib()
    for start in range(0, len(todo), 64):
        part = todo[start : start + 64]
        n = len(part)
        # (the pointer arrays of an unchanged set of tensors are built once: gradients live in persistent buffers)
        key = tuple((p.data_ptr(), p.grad.data_ptr()) for p, _, _ in part)
        arrays = cache.get(key)
        if arrays is None:
            if len(cache) >= 64:  # (a fit has a handful of tensor sets; gradients re-allocated every step must not pile up)
                cache.clear()
            arrays = cache[key] = (
                ptr_array([p.data for p, _, _ in part]), ptr_array([p.grad for p, _, _ in part]),
                ptr_array([st["exp_avg"] for _, st, _ in part]), ptr_array([st["exp_avg_sq"] for _, st, _ in part]),
                (ctypes.c_int * n)(*[p.numel() for p, _, _ in part]),
            )
        check(lib.jd_adam_step_multi(
            n, *arrays, (ctypes.c_float * n)(*[b[0] for _, _, b in part]), (ctypes.c_float * n)(*[b[1] for _, _, b in part]),
            beta1, beta2, 1 - beta1, 1 - beta2, cfg.optimizer_kwargs.get("eps", 1e-8),
            None if bias_dev is None else ptr(bias_dev[2 * start : 2 * (start + n)]), stream_ptr(part[0][0].device),
        ))


class _CalibrationStepper:
    """The optimizer of ONE dataset's calibration parameters (a (1, 2) shift and a (1,) log background norm) on the
    library's step kernel: `jd_adam_step` / `jd_sgd_step` with ``use_log_flux=0`` is the plain `torch.optim.Adam` /
    `SGD` update of a parameter vector (per-parameter state and step count, a parameter whose ``grad`` is None is
    skipped -- jolideco/core.py:197-204,229).  A `torch.optim` step on two tiny device tensors costs the host ~0.5 ms
    (a dozen launches each); eight calibrated observations made the step host bound (bench config c6)."""

    def __init__(self, params, deconvolver):
        self.params = list(params)
        self.cfg = deconvolver
        self.state = [
            {"step": 0, "exp_avg": torch.zeros_like(p.data), "exp_avg_sq": torch.zeros_like(p.data)} for p in self.params
        ]

    def zero_grad(self, set_to_none=True):
        for p in self.params:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    def step(self, bias_dev=None):
        lib, cfg = _hip.lib(), self.cfg
        if cfg.optimizer_type == "adam":  # both parameters of the dataset in one launch
            _adam_step_many(cfg, list(zip(self.params, self.state)), self.__dict__.setdefault("_arrays", {}), bias_dev)
            return
        lr = cfg.optimizer_kwargs["lr"]
        for p, st in zip(self.params, self.state):
            if p.grad is None:
                continue
            st["step"] += 1
            data, n, stream = p.data, p.numel(), stream_ptr(p.device)
            check(lib.jd_sgd_step(ptr(data), ptr(data), ptr(data), ptr(p.grad), None, n, lr, 0, 0, stream))


class StepScalars:
    """The per-step scalars of an epoch in DEVICE memory: the cycle-spin shifts of every prior evaluation and the Adam bias
    terms of every optimizer step (include/jolideco_hip.h: device-resident step scalars).  The launch arguments of an
    epoch are then the same from epoch to epoch -- the values travel in ONE small host-to-device copy per epoch, enqueued
    in front of the epoch's launches -- which is what lets `FitSession` capture an epoch in a hipGraph and replay it.

    One int32 device buffer [shifts: 2 per slot | bias terms: 2 floats per slot, as bit patterns]; a ring of pinned host
    rows so that a row is not overwritten before its copy has run (an event per row, waited for only when the ring
    wraps around a copy that has not finished)."""

    RING = 64

    def __init__(self, device, n_shift, n_bias):
        self.n_shift, self.n_bias = n_shift, n_bias
        n = max(2 * (n_shift + n_bias), 4)
        self.dev = torch.zeros(n, dtype=torch.int32, device=device)
        self.dev_f = self.dev.view(torch.float32)
        self.host = torch.zeros((self.RING, n), dtype=torch.int32).pin_memory()
        self.host_ptr, self.row_bytes = self.host.data_ptr(), 4 * n
        self.host_i = self.host.numpy()
        self.host_f = self.host.view(torch.float32).numpy()
        self.events = [torch.cuda.Event() for _ in range(self.RING)]
        self.used = [False] * self.RING
        self.slot = 0
        self.shift_slots = [self.dev[2 * k : 2 * k + 2] for k in range(n_shift)]
        self.bias_slots = [self.dev_f[2 * (n_shift + k) : 2 * (n_shift + k) + 2] for k in range(n_bias)]

    def bias_range(self, k, n):
        """Device view of the bias slots k .. k + n - 1 (2 n floats)."""
        return self.dev_f[2 * (self.n_shift + k) : 2 * (self.n_shift + k + n)]

    def upload(self, shifts, biases):
        """shifts: [(y, x)] residues per slot; biases: [(step_size, bias2_sqrt)] per slot -> device (asynchronous)."""
        slot = self.slot
        if self.used[slot]:
            self.events[slot].synchronize()
        if shifts:
            self.host_i[slot, : 2 * len(shifts)] = np.asarray(shifts, dtype=np.int32).reshape(-1)
        if biases:
            base = 2 * self.n_shift
            self.host_f[slot, base : base + 2 * len(biases)] = np.asarray(biases, dtype=np.float32).reshape(-1)
        # (one small block reads the pinned row over the host link: cheaper on the stream than a copy-engine transfer)
        check(_hip.lib().jd_step_scalars_fetch(ctypes.c_void_p(self.host_ptr + slot * self.row_bytes), ptr(self.dev),
                                               self.dev.numel(), stream_ptr(self.dev.device)))
        self.events[slot].record()
        self.used[slot] = True
        self.slot = (slot + 1) % self.RING


class FitSession:
    """Device state of one fit and the per-epoch step sequence (jolideco/core.py:209-247).

    ``epoch()`` enqueues, without any host synchronisation:

    * sequential mode -- for every dataset: fused forward model + Poisson NLL + gradient, prior
      value + gradient scaled by ``-beta / n_datasets``, fused chain rule + optimizer step; then
      the trace evaluation of every dataset loss and every prior on the STALE fluxes (core.py:247);
    * joint mode -- the gradients of this rank's datasets and of its share of the prior
      accumulated into one flat buffer, ONE all-reduce (world size > 1), one optimizer step.

    ``scalars`` holds the epoch's [dataset losses (global order) | log-priors | validation losses].
    """

    def __init__(self, deconvolver, datasets, datasets_validation, components, dist, calibrations=None):
        self.cfg = deconvolver
        self.dist = dist
        # (sharded joint fit: a dataset's calibration lives -- parameters, gradients, its small optimizer -- on the rank
        # that owns the dataset; `gather_calibrations` hands the values to every rank for results and checkpoints)
        self.calibrations = calibrations
        device = deconvolver.device
        self.components = components = components.to(device)
        self.joint = deconvolver.fit_mode == "joint"
        names_all = list(datasets)
        # joint mode shards the datasets over the ranks; sequential mode runs full replicas
        self.prior_shares = None
        if self.joint and dist.sharded:
            # cost-aware placement (identical on every rank): datasets by longest-processing-time-first on their estimated
            # cost, then the prior's patch rows in shares that top every rank up to the same estimated load
            from .distributed import balanced_shares, lpt_assignment
            from .models.npred import COST_PRIOR_FIXED, COST_PRIOR_PER_128_COMPONENTS, estimate_dataset_cost

            if os.environ.get("JOLIDECO_DIST_PLACEMENT", "cost") == "round-robin":
                local_names = dist.shard_items(names_all)
            else:
                costs = [estimate_dataset_cost(datasets[n], components, bool(calibrations is not None and n in calibrations))
                         for n in names_all]
                owners, loads = lpt_assignment(costs, dist.world_size)
                local_names = [n for n, owner in zip(names_all, owners) if owner == dist.rank]
                prior_cost = 0.0
                for comp in components.values():
                    prior = getattr(comp, "prior", None)
                    if getattr(prior, "shardable", False) and not comp.frozen:
                        k = getattr(getattr(prior, "gmm", None), "n_components", 128)
                        up = comp.upsampling_factor or 1
                        prior_cost += (COST_PRIOR_FIXED + COST_PRIOR_PER_128_COMPONENTS * k / 128.0) * up * up
                self.prior_shares = balanced_shares(loads, prior_cost)
                self.rank_loads = loads
        else:
            local_names = names_all
        local_datasets = {n: datasets[n] for n in local_names}
        if not local_datasets and not self.joint:
            raise ValueError("no datasets given")

        self.total_loss = TotalLoss.from_datasets_and_components(
            datasets=local_datasets if local_datasets else {},
            datasets_validation=datasets_validation,
            components=components,
            beta=deconvolver.beta,
            calibrations=calibrations,
            device=device,
        )
        # One small Adam per calibrated dataset, stepped when (and only when) that dataset took part in
        # the step: exactly what the reference's single optimizer does, because torch.optim.Adam keeps
        # per-parameter state and skips parameters whose grad is None (jolideco/core.py:197-204,215,229).
        self.cal_optimizers = []
        for models in self.total_loss.poisson_loss.npred_models_all:
            cal = models.calibration
            params = [] if cal is None else [p for p in cal.parameters() if p.requires_grad]
            self.cal_optimizers.append(_CalibrationStepper(params, deconvolver) if params else None)
        # the trace always has one column per GLOBAL dataset
        self.total_loss.poisson_loss.names_all_global = names_all
        self.priors = list(self.total_loss.prior_loss.priors.values())
        self.n_d, self.n_c = len(names_all), len(components)
        val = self.total_loss.poisson_loss_validation
        self.n_val = val.n_datasets if val else 0
        # ONE flat buffer = [flux gradients of all components | scalars of one epoch] so that the
        # joint step needs a single all-reduce.
        numels = [c._flux_upsampled.numel() for c in components.values()]
        n_scalars = self.n_d + self.n_c + self.n_val
        self.comm = torch.zeros(sum(numels) + n_scalars, dtype=torch.float32, device=device)
        offsets = np.concatenate([[0], np.cumsum(numels)])
        self.states = [
            _ComponentState(name, comp, self.comm[offsets[i] : offsets[i + 1]])
            for i, (name, comp) in enumerate(components.items())
        ]
        self.scalars = self.comm[offsets[-1] :]
        slot_d = {name: i for i, name in enumerate(names_all)}
        self.local_idx = [(slot_d[name], i) for i, name in enumerate(local_names)]  # (global slot, local index)
        self.step = 0
        # comm_events = []: a sharded step brackets its collectives with event pairs on the compute stream (bench.py reads
        # them after an untimed phase: how long the stream waited for the all-reduce it overlapped, and for the all-gather)
        self.comm_events = None

        # The optimizer step inside the last kernel of a component's gradient (the prior's gather kernel, or the band sum of
        # a sharded prior): decided ONCE here -- the deconvolver's own `_optimizer_step` (a subclass or an instance that
        # overrides it is stepped through its hook and sees every gradient), no JOLIDECO_NO_FUSED_STEP, and the tiled
        # gather kernel the fused form lives in (option JD_GMM_GATHER_TILED=0 selects the per-pixel kernel: two calls).
        from . import _hip as _hip_mod

        own_step = getattr(type(deconvolver), "_optimizer_step", None) is MAPDeconvolver._optimizer_step
        tiled = _hip_mod.get_option("JD_GMM_GATHER_TILED") in (None, 1)
        self.fuse_optimizer_step = bool(own_step and "_optimizer_step" not in vars(deconvolver) and tiled
                                        and not os.environ.get("JOLIDECO_NO_FUSED_STEP"))
        self._setup_sharded_prior(os.environ.get("JOLIDECO_DIST_OVERLAP", "1") != "0")
        batchable = not os.environ.get("JOLIDECO_NO_BATCH") and self.total_loss.poisson_loss.batchable(
            [li for _, li in self.local_idx]
        )
        self.batch_joint = self.joint and batchable
        # every flux is exp(theta) [x a non-negative mask]: components that share one forward operator are then evaluated as
        # ONE image, their sum (`PoissonLoss.fwd_bwd_batch`, models/npred.py:241-261)
        self.flux_nonneg = all(
            st.use_log_flux and (st.mask is None or bool((st.mask >= 0).all())) for st in self.states
        )
        # calibrated / up-sampled datasets of one flux component on the native FFT path: their own batched step
        self.batch_joint_calibrated = bool(
            self.joint and not batchable and not os.environ.get("JOLIDECO_NO_BATCH") and self.n_c == 1
            and self.total_loss.poisson_loss.batchable_calibrated([li for _, li in self.local_idx])
        )
        # sequential mode: the per-epoch trace evaluates every dataset on the same stale flux -- one batched launch
        self.batch_trace = (not self.joint) and batchable
        # Planned epochs (single process, the session's own optimizer step): the per-step scalars live in device memory
        # (`StepScalars`), so an epoch's launch arguments never change -- and after `GRAPH_WARMUP` epochs the epoch is
        # captured in a hipGraph per flux-buffer parity and replayed (JOLIDECO_GRAPH=0: planned epochs without capture;
        # JOLIDECO_STEP_SCALARS=host: the by-value form of rounds 1-4 throughout)
        # JOLIDECO_GRAPH: "1" capture always, "0" never (planned epochs enqueued eagerly), unset / "auto": the first
        # AUTO_PROBE epochs run by value and are timed -- host time to enqueue an epoch against the device's time for it.
        # Where the host needs less than AUTO_CLEAR of the device's time the fit stays by value (a replayed epoch costs the
        # device a few microseconds more per step than eagerly launched kernels -- a fetch of the step scalars, the graph's
        # own hand-overs: +1.3 % on the 0.62 ms step of the benchmark); otherwise the epoch is captured, AUTO_PROBE replays
        # are timed the same way, and the faster form stays (tools/gpu/small_fits.py, profiles/r05/small_fits.txt: the
        # host/device ratio alone misjudges fits in between -- 1024^2 x 8 with the GMM prior, ratio 0.64: 343 us by value,
        # 311 replayed; with the uniform prior, ratio 0.75: 194 by value, 205 replayed)
        mode = os.environ.get("JOLIDECO_GRAPH", "auto").lower()
        self.graph_mode = {"1": "always", "on": "always", "0": "never", "off": "never"}.get(mode, "auto")
        if getattr(deconvolver, "use_graph", None) is not None:
            self.graph_mode = "always" if deconvolver.use_graph else "never"
        self.use_graph = self.graph_mode == "always"
        self.graph_policy = {"always": "captured epochs (forced)", "never": "no capture (forced)"}.get(self.graph_mode, "undecided")
        self._probe = []  # (host seconds, start event, end event) of the by-value probe epochs
        self._trial = None  # the replayed epochs being timed against them
        # the GMM prior's first phase on a second stream beside the likelihood launches (`_start_priors`)
        # JOLIDECO_PRIOR_OVERLAP: "1" / "0" forced; unset: the "auto" policy times both and keeps the faster (the two sides
        # compete for the same CUs: +6-9 % at 2048^2 x 8 and 4096^2, -10 % at 1024^2 x 4 -- profiles/r05/ab_prior_overlap.txt)
        self.overlap_mode = {"1": "on", "0": "off"}.get(os.environ.get("JOLIDECO_PRIOR_OVERLAP", ""), "auto")
        self.overlap_prior = self.overlap_mode != "off"
        # JOLIDECO_PRIOR_STREAM_PRIORITY (tuning): priority of the side stream (-1: its launches are dispatched first)
        side_priority = int(os.environ.get("JOLIDECO_PRIOR_STREAM_PRIORITY", "0"))
        self._side_stream = (torch.cuda.Stream(device=device, priority=side_priority)
                             if torch.device(device).type == "cuda" else None)
        self.step_scalars = None
        self._graphs = {}
        self._epochs_done = 0
        self._total_epochs = 0
        self._planned_ok = os.environ.get("JOLIDECO_STEP_SCALARS", "device") != "host"

    def gather_calibrations(self):
        """Sharded joint fit: every rank receives the current calibration parameters of the datasets the other ranks
        own (one small all-reduce: each owner contributes its (shift_x, shift_y, log background norm), the others
        zeros).  A collective: every rank must call it.  No-op for a single process or without calibrations."""
        cals, dist = self.calibrations, self.dist
        if cals is None or not (self.joint and dist.sharded) or dist.dry_run:
            return
        names = self.total_loss.poisson_loss.names_all_global
        mine = {names[gslot] for gslot, _ in self.local_idx}
        flat = torch.zeros(3 * len(names), dtype=torch.float32, device=self.comm.device)
        for i, name in enumerate(names):
            if name in mine and name in cals:
                cal = cals[name]
                flat[3 * i : 3 * i + 2] = cal.shift_xy.detach().reshape(-1).to(flat.device)
                flat[3 * i + 2] = cal._background_norm.detach().reshape(-1)[0].to(flat.device)
        dist.all_reduce_sum(flat)
        host = flat.cpu()
        for i, name in enumerate(names):
            if name not in mine and name in cals:
                cal = cals[name]
                with torch.no_grad():
                    cal.shift_xy.copy_(host[3 * i : 3 * i + 2].reshape(1, 2))
                    cal._background_norm.copy_(host[3 * i + 2 : 3 * i + 3])

    def _setup_sharded_prior(self, overlap):
        """Sharded joint fit: every rank evaluates the GMM prior on its band of patch rows.  With ``overlap`` (default)
        the band of the prior gradient travels in ONE all-gather of compact pieces [bands of the shardable priors |
        their values] while the all-reduce of the likelihood gradient is in flight; otherwise the prior gradient is
        accumulated into the flat buffer before its (single, blocking) all-reduce."""
        from .ops import band_rows

        self.band_plan = None
        dist = self.dist
        if not (self.joint and dist.sharded):
            return
        # ranks must draw identical cycle-spin shifts: identical generator states at the start
        state = []
        for prior in self.priors:
            generator = getattr(prior, "generator", None)
            if prior.shardable and generator is not None:
                state.append(int(np.frombuffer(generator.get_state().numpy().tobytes(), dtype=np.uint8).astype(np.int64).sum()))
                state.append(int(generator.initial_seed() % (1 << 62)))
        dist.assert_same_on_all_ranks(state or [0], "the state of the cycle-spin generators")
        # the schedule itself must agree too: a rank with another JOLIDECO_DIST_OVERLAP or another set of frozen
        # components would issue different collectives (a hang, or a size mismatch deep inside the backend)
        frozen = [int(st.frozen) for st in self.states]
        dist.assert_same_on_all_ranks([int(bool(overlap)), len(self.states)] + frozen, "the collective schedule (JOLIDECO_DIST_OVERLAP, "
                                      "frozen components)")
        if not overlap:
            return
        plan, offset = [], 0
        for ci, (st, prior) in enumerate(zip(self.states, self.priors)):
            if not prior.shardable or st.frozen:
                continue
            H, W = st.shape
            n_rows = prior.n_patch_rows(st.shape)
            y_ranges = []
            for r in range(dist.world_size):
                rows = DistContext(r, dist.world_size).shard_range(n_rows, self.prior_shares)
                y_ranges.append(band_rows(rows, prior.stride, H))
            size = max(y1 - y0 for y0, y1 in y_ranges) * W
            plan.append({"ci": ci, "offset": offset, "size": size, "y_ranges": y_ranges,
                         "rows": dist.shard_range(n_rows, self.prior_shares)})
            offset += size
        if not plan:
            return
        for i, item in enumerate(plan):
            item["value"] = offset + i  # the shard's prior value rides behind the bands
        chunk = -(-(offset + len(plan)) // 4) * 4  # multiple of 4 floats: 16-byte aligned pieces
        device = self.comm.device
        self.band_plan = plan
        self.band_chunk = chunk
        self.band_send = torch.zeros(chunk, dtype=torch.float32, device=device)
        self.band_recv = torch.zeros(chunk * dist.world_size, dtype=torch.float32, device=device)
        dist.assert_same_on_all_ranks([len(plan), chunk], "the band plan of the sharded prior")

    def _slot(self, i):
        return self.scalars[i : i + 1]

    def _cal_zero_grad(self, li):
        opt = self.cal_optimizers[li]
        if opt is not None:
            opt.zero_grad(set_to_none=True)

    def _cal_step(self, li):
        opt = self.cal_optimizers[li]
        if opt is not None:
            opt.step()

    def _cal_step_all(self):
        """The calibration steps of all local datasets (a joint step updates them together): with Adam ONE launch for every
        parameter that has a gradient (jd_adam_step_multi: the update of `_CalibrationStepper.step`, each parameter with
        its own step count); otherwise the per-dataset steps."""
        steppers = [self.cal_optimizers[li] for _, li in self.local_idx if self.cal_optimizers[li] is not None]
        if not steppers:
            return
        cfg = self.cfg
        if cfg.optimizer_type != "adam":
            for opt in steppers:
                opt.step()
            return
        items = [(p, st) for opt in steppers for p, st in zip(opt.params, opt.state)]
        _adam_step_many(cfg, items, self.__dict__.setdefault("_cal_step_cache", {}))

    def _prior_rows(self, prior, state):
        if self.joint and self.dist.sharded and prior.shardable:
            return self.dist.shard_range(prior.n_patch_rows(state.shape), self.prior_shares)
        return None

    def _apply_step(self, states, stepped):
        """The optimizer step of the components the prior's gather kernel has not stepped already.  The hook keeps its
        two-argument form `(states, step)` for a replacement installed by a test (which also switches the fusion off)."""
        if stepped:
            self.cfg._optimizer_step(states, self.step, stepped)
        else:
            self.cfg._optimizer_step(states, self.step)

    def _fuse_step(self, st, prior):
        """The prior of this component applies the optimizer step itself (`device_fwd_bwd_step`): single process (the
        gradient buffer is complete when the prior runs), not frozen, a prior that supports it, stride >= 4, and the
        session's own `_optimizer_step` (tests that replace it to record or suppress the step see every gradient)."""
        return (
            self.fuse_optimizer_step and not self.dist.sharded and not st.frozen
            and getattr(prior, "supports_fused_step", False) and getattr(prior, "stride", 0) >= 4
            and "_optimizer_step" not in vars(self.cfg)  # (a hook installed on the instance AFTER the session was built)
        )

    def _fuse_band_step(self, st):
        """Sharded fits: the bands of the prior's gradient are added and the optimizer step applied in one launch
        (jd_add_rolled_bands_step) -- not frozen, width a multiple of 4 and 16-byte aligned images, the session's own
        `_optimizer_step`, no JOLIDECO_NO_FUSED_STEP."""
        if st.frozen or not self.fuse_optimizer_step or "_optimizer_step" in vars(self.cfg):
            return False
        images = [st.theta, st.flux[0], st.flux[1], st.grad, getattr(st, "exp_avg", None), getattr(st, "exp_avg_sq", None), st.mask]
        return st.grad.shape[-1] % 4 == 0 and all(t is None or t.data_ptr() % 16 == 0 for t in images)

    def _timed(self, name, fn):
        """Run fn(); with `comm_events` set, bracketed by an event pair on the current stream."""
        if self.comm_events is None:
            return fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn()
        e1.record()
        self.comm_events.append((name, e0, e1))
        return out

    def comm_times_ms(self):
        """{collective: mean ms per step} of the event pairs collected since `comm_events = []` (synchronises)."""
        torch.cuda.synchronize()
        sums, counts = {}, {}
        for name, e0, e1 in self.comm_events or []:
            sums[name] = sums.get(name, 0.0) + e0.elapsed_time(e1)
            counts[name] = counts.get(name, 0) + 1
        return {name: sums[name] / counts[name] for name in sums}

    # ---- planned epochs: device-resident step scalars, hipGraph capture -------------------------------------------
    GRAPH_WARMUP = 3  # eager epochs before an epoch is captured (lazy allocations, table uploads, kernel attributes)

    def reset_graphs(self):
        """Forget the captured epochs (a library option that changes what an epoch launches was set: `_hip.set_option`
        bumps `_hip.OPTION_GENERATION`, and `epoch` compares); the "auto" policy decides anew."""
        self._graphs = {}
        self._epochs_done = 0
        self._option_generation = _hip.OPTION_GENERATION
        self._trial = None
        if self.graph_mode == "auto":
            self.use_graph, self.graph_policy, self._probe = False, "undecided", []
            if self.overlap_mode == "auto":
                self.overlap_prior = True

    def _planned_capable(self):
        """Planned epochs apply: one process, the session's own optimizer step (a hook sees every gradient through the
        by-value path), no event brackets around collectives, no kernel timers."""
        cfg = self.cfg
        own = getattr(type(cfg), "_optimizer_step", None) is MAPDeconvolver._optimizer_step and "_optimizer_step" not in vars(cfg)
        return (self._planned_ok and own and not self.dist.sharded and self.comm_events is None and not self.n_val
                and cfg.optimizer_type in ("adam", "sgd"))

    def _plan_slots(self):
        """Shift / bias slots of an epoch (fixed for the session) and the calibration steppers of every optimizer step."""
        cached = getattr(self, "_plan_slots_cache", None)
        if cached is not None:
            return cached
        drawing = [ci for ci, prior in enumerate(self.priors) if hasattr(prior, "draw_shifts")]
        n_local = len(self.local_idx)
        if self.joint:
            n_shift = len(drawing)
            cal_groups = [[self.cal_optimizers[li] for _, li in self.local_idx if self.cal_optimizers[li] is not None]]
            n_flux = 1
        else:
            n_shift = len(drawing) * (n_local + 1)  # every step + the trace
            cal_groups = [[self.cal_optimizers[li]] if self.cal_optimizers[li] is not None else [] for _, li in self.local_idx]
            n_flux = n_local
        n_cal = sum(len(opt.params) for group in cal_groups for opt in group)
        self._plan_slots_cache = (drawing, n_shift, n_flux, cal_groups, n_cal)
        return self._plan_slots_cache

    def _plan_epoch(self):
        """Host side of an epoch: draw the cycle-spin shifts of every prior evaluation IN THE ORDER the evaluations run
        (the generators advance exactly as in the by-value path), compute the bias terms of every optimizer step, and
        send both to the device.  Returns the plan the launches read: DeviceShifts per (step, prior), bias slots, and per
        optimizer step the calibration parameters that take it -- those that had a gradient after the previous epoch (a
        shift that is exactly zero never gets one: jolideco/models/npred.py:225-232; torch.optim.Adam skips it)."""
        from .ops import DeviceShifts

        cfg = self.cfg
        drawing, n_shift, n_flux, cal_groups, n_cal = self._plan_slots()
        if self.step_scalars is None:
            self.step_scalars = StepScalars(self.comm.device, n_shift, n_flux + n_cal)
        sc = self.step_scalars
        lr = cfg.optimizer_kwargs["lr"]
        beta1, beta2 = cfg.optimizer_kwargs.get("betas", (0.9, 0.999))
        adam = cfg.optimizer_type == "adam"
        n_eval = 1 if self.joint else len(self.local_idx) + 1
        shifts_host, shifts = [], []
        if len(drawing) == 1:  # one drawing prior: all draws of the epoch in one call (same numbers, same generator state)
            ci = drawing[0]
            prior, (H, W) = self.priors[ci], self.states[ci].shape
            for drawn in prior.draw_shifts_many(n_eval):
                if drawn is None:
                    shifts.append({ci: None})
                    shifts_host.append((0, 0))
                else:
                    shifts.append({ci: DeviceShifts(sc.shift_slots[len(shifts_host)], drawn)})
                    shifts_host.append((drawn[0] % H, drawn[1] % W))
        else:
            for _ in range(n_eval):
                row = {}
                for ci in drawing:
                    prior, (H, W) = self.priors[ci], self.states[ci].shape
                    drawn = prior.draw_shifts()
                    if drawn is None:
                        row[ci] = None
                        shifts_host.append((0, 0))
                    else:
                        row[ci] = DeviceShifts(sc.shift_slots[len(shifts_host)], drawn)
                        shifts_host.append((drawn[0] % H, drawn[1] % W))
                shifts.append(row)
        biases = [adam_bias_terms(self.step + j + 1, lr, beta1, beta2) if adam else (0.0, 1.0) for j in range(n_flux)]
        flux_bias = [sc.bias_slots[j] if adam else None for j in range(n_flux)]
        cal_bias, cal_items, k = [], [], n_flux
        for group in cal_groups:
            items = [(p, st) for opt in group for p, st in zip(opt.params, opt.state) if p.grad is not None]
            for _, st in items:
                biases.append(adam_bias_terms(st["step"] + 1, lr, beta1, beta2) if adam else (0.0, 1.0))
            cal_bias.append(sc.bias_range(k, len(items)) if (adam and items) else None)
            cal_items.append(items)
            k += len(items)
        sc.upload(shifts_host, biases)
        return {"shifts": shifts, "flux_bias": flux_bias, "cal_bias": cal_bias, "cal_items": cal_items, "n_steps": n_flux,
                "signature": tuple(len(items) for items in cal_items)}

    def _plan_by_value(self):
        """The same plan with HOST scalars: the shifts as drawn (in evaluation order), no device slots -- the kernels take
        them by value.  The single-process epochs of the default policy (device bound fits) run through `_enqueue_epoch`
        with this plan: one launch sequence for both forms."""
        drawing, _, n_flux, cal_groups, _ = self._plan_slots()
        n_eval = 1 if self.joint else len(self.local_idx) + 1
        shifts = [{ci: self.priors[ci].draw_shifts() for ci in drawing} for _ in range(n_eval)]
        cal_items = [[(p, st) for opt in group for p, st in zip(opt.params, opt.state)] for group in cal_groups]
        return {"shifts": shifts, "flux_bias": [None] * n_flux, "cal_bias": [None] * len(cal_groups), "cal_items": cal_items,
                "n_steps": n_flux, "signature": ()}

    # ---- the prior beside the likelihood ------------------------------------------------------------------------------
    def _overlap_active(self):
        """Phase 1 of a GMM prior (value + gradient rows: it reads the flux only) runs on a second stream while the main
        stream runs the likelihood launches of the step; the streams join in front of the gather (+ optimizer step).  Not
        while the kernel timers run (a kernel timed beside another one is not the kernel's time); JOLIDECO_PRIOR_OVERLAP=0:
        one stream."""
        return self.overlap_prior and self._side_stream is not None and not self.dist.sharded and not _hip.profile_active()

    def _prior_calls(self, coef, shifts, bias):
        """The prior evaluations of ONE optimizer step as callables `call(phases)` (3: the whole pass): [(ci, call, steps)],
        `steps`: the prior's gather kernel applies the component's optimizer step."""
        cfg, n_d, slot, step_no = self.cfg, self.n_d, self._slot, self.step + 1
        calls = []
        for ci, (st, prior) in enumerate(zip(self.states, self.priors)):
            if getattr(prior, "value_is_zero", False):
                continue  # (a uniform prior: its slot holds the 0 the first epoch wrote -- no fill launch per step)
            kwargs = {"shifts": shifts[ci]} if ci in shifts else {}
            flux, value = st.flux_cur, slot(n_d + ci)
            if self._fuse_step(st, prior):
                args = cfg._step_args(st, step_no, bias)

                def call(phases=3, prior=prior, flux=flux, value=value, args=args, kwargs=kwargs):
                    prior.device_fwd_bwd_step(flux, value, coef, args, **(dict(kwargs, phases=phases) if phases != 3 else kwargs))

                calls.append((ci, call, True))
            else:

                def call(phases=3, prior=prior, flux=flux, value=value, grad=st.grad, kwargs=kwargs):
                    prior.device_fwd_bwd(flux, value, grad=grad, coef=coef, **(dict(kwargs, phases=phases) if phases != 3 else kwargs))

                calls.append((ci, call, False))
        return calls

    def _start_priors(self, calls):
        """Phase 1 of every prior that has one (at most one per GMM handle: a handle holds one pass at a time) on the side
        stream, behind everything the main stream has been given so far.  Returns the components started."""
        if not self._overlap_active():
            return set()
        eligible, seen = [], set()
        for ci, call, _ in calls:
            prior = self.priors[ci]
            key = id(getattr(prior, "gmm", prior))
            if getattr(prior, "supports_phases", False) and key not in seen:
                seen.add(key)
                eligible.append((ci, call))
        if not eligible:
            return set()
        main = torch.cuda.current_stream(self.comm.device)
        self._side_stream.wait_stream(main)
        with torch.cuda.stream(self._side_stream):
            for _, call in eligible:
                call(1)
        return {ci for ci, _ in eligible}

    def _finish_priors(self, calls, early):
        """Join the side stream, then phase 2 of the priors started early and the whole pass of the others, in component
        order.  Returns the components whose optimizer step the prior applied."""
        if early:
            torch.cuda.current_stream(self.comm.device).wait_stream(self._side_stream)
        stepped = set()
        for ci, call, steps in calls:
            call(2 if ci in early else 3)
            if steps:
                stepped.add(ci)
        return stepped

    def _commit_replay(self, plan):
        """The host state an eagerly enqueued epoch leaves behind, after a REPLAYED one: step counts, flux buffer parity."""
        self.step += plan["n_steps"]
        if plan["n_steps"] % 2:
            for st in self.states:
                st.cur = 1 - st.cur
        for items in plan["cal_items"]:
            for _, st in items:
                st["step"] += 1

    def _epoch_planned(self):
        plan = self._plan_epoch()
        parity = tuple(st.cur for st in self.states) + plan["signature"]
        graph = self._graphs.get(parity) if self.use_graph else None
        if graph is not None:
            if self._trial is not None:
                self._trial_replay(graph)
            else:
                graph.replay()
            self._commit_replay(plan)
        elif self.use_graph and self._epochs_done >= self.GRAPH_WARMUP and not _hip.profile_active():
            # capture this epoch's launches (nothing runs during the capture), then run them
            snapshot = (self.step, [st.cur for st in self.states], [[st["step"] for _, st in items] for items in plan["cal_items"]])
            graph = torch.cuda.CUDAGraph()
            try:
                with torch.cuda.graph(graph):
                    self._enqueue_epoch(plan)
            except Exception:
                # (something in this fit cannot be captured: planned epochs without graphs from here on; the host state the
                # aborted enqueue changed is put back and the epoch runs eagerly)
                log.warning("hipGraph capture of an epoch failed; continuing without graphs", exc_info=True)
                self.use_graph = False
                self.step = snapshot[0]
                for st, cur in zip(self.states, snapshot[1]):
                    st.cur = cur
                for items, values in zip(plan["cal_items"], snapshot[2]):
                    for (_, st), value in zip(items, values):
                        st["step"] = value
                torch.cuda.synchronize()
                self._enqueue_epoch(plan)
            else:
                self._graphs[parity] = graph
                graph.replay()
        else:
            self._enqueue_epoch(plan)
        self._epochs_done += 1

    def _enqueue_epoch(self, plan):
        """The launches of one epoch of a single-process fit, reading the step scalars of `plan` from device memory: the
        body of `_epoch_by_value` without its sharded branches (same kernels, same order, same results bit for bit)."""
        cfg, states, priors, total_loss = self.cfg, self.states, self.priors, self.total_loss
        n_d, n_c = self.n_d, self.n_c
        slot = self._slot
        beta = -float(cfg.beta)

        def flux_step(stepped, bias):
            self.step += 1
            for st in states:
                st.bias_dev = bias
            try:
                self._apply_step(states, stepped)
            finally:
                for st in states:
                    st.bias_dev = None

        def cal_steps(items, bias):
            if not items:
                return
            if cfg.optimizer_type != "adam":
                lr = cfg.optimizer_kwargs["lr"]
                for p, st in items:
                    if p.grad is None:
                        continue
                    st["step"] += 1
                    check(_hip.lib().jd_sgd_step(ptr(p.data), ptr(p.data), ptr(p.data), ptr(p.grad), None, p.numel(), lr, 0, 0,
                                                 stream_ptr(p.device)))
                return
            _adam_step_many(cfg, items, self.__dict__.setdefault("_cal_step_cache", {}), bias)

        if self.joint:
            fluxes = [st.flux_cur for st in states]
            grads = [st.grad for st in states]
            calls = self._prior_calls(beta, plan["shifts"][0], plan["flux_bias"][0])
            early = self._start_priors(calls)  # (the priors' first phase beside the likelihood launches below)
            first = True
            if self.batch_joint:
                total_loss.poisson_loss.fwd_bwd_batch(
                    [li for _, li in self.local_idx], fluxes if n_c > 1 else fluxes[0],
                    [slot(gslot) for gslot, _ in self.local_idx], grad=grads if n_c > 1 else grads[0], accumulate=False,
                    flux_nonneg=self.flux_nonneg,
                )
                first = False
            elif self.batch_joint_calibrated:
                for _, li in self.local_idx:
                    self._cal_zero_grad(li)
                total_loss.poisson_loss.fwd_bwd_batch_calibrated(
                    [li for _, li in self.local_idx], fluxes[0], [slot(gslot) for gslot, _ in self.local_idx], grad=grads[0],
                    accumulate=False,
                )
                first = False
            else:
                for gslot, li in self.local_idx:
                    self._cal_zero_grad(li)
                    total_loss.poisson_loss.fwd_bwd(li, fluxes, slot(gslot), grads=grads, accumulate=not first,
                                                    flux_nonneg=self.flux_nonneg)
                    first = False
            if first:
                for g in grads:
                    g.zero_()
            flux_step(self._finish_priors(calls, early), plan["flux_bias"][0])
            cal_steps(plan["cal_items"][0], plan["cal_bias"][0])
        else:
            coef = beta / total_loss.prior_weight
            for j, (gslot, li) in enumerate(self.local_idx):
                fluxes = [st.flux_cur for st in states]
                grads = [st.grad for st in states]
                calls = self._prior_calls(coef, plan["shifts"][j], plan["flux_bias"][j])
                early = self._start_priors(calls)
                self._cal_zero_grad(li)
                total_loss.poisson_loss.fwd_bwd(li, fluxes, slot(gslot), grads=grads, accumulate=False,
                                                flux_nonneg=self.flux_nonneg)
                flux_step(self._finish_priors(calls, early), plan["flux_bias"][j])
                cal_steps(plan["cal_items"][j], plan["cal_bias"][j])
            stale = [st.flux_trace for st in states]
            if self.batch_trace:
                total_loss.poisson_loss.fwd_bwd_batch(
                    [li for _, li in self.local_idx], stale if n_c > 1 else stale[0],
                    [slot(gslot) for gslot, _ in self.local_idx], flux_nonneg=self.flux_nonneg,
                )
            else:
                for gslot, li in self.local_idx:
                    total_loss.poisson_loss.fwd_bwd(li, stale, slot(gslot), flux_nonneg=self.flux_nonneg)
            trace_shifts = plan["shifts"][-1]
            for ci, (st, prior) in enumerate(zip(states, priors)):
                if getattr(prior, "value_is_zero", False):
                    continue
                kwargs = {"shifts": trace_shifts[ci]} if ci in trace_shifts else {}
                prior.device_fwd_bwd(st.flux_trace, slot(n_d + ci), **kwargs)

    def epoch(self):
        """One epoch of the fit, enqueued without any host synchronisation: planned (device-resident step scalars, replayed
        from a captured hipGraph once warm) where `_planned_capable` holds, else the by-value form."""
        if getattr(self, "_option_generation", None) != _hip.OPTION_GENERATION:
            self.reset_graphs()
        # (the first epoch runs by value: it shows which calibration parameters receive a gradient at all)
        if self._total_epochs > 0 and self._planned_capable() and not _hip.profile_active():
            if self.graph_mode != "auto" or self.use_graph:
                self._total_epochs += 1
                return self._epoch_planned()
            if self.graph_policy == "undecided":
                return self._epoch_probe()
            self._total_epochs += 1
            return self._enqueue_epoch(self._plan_by_value())
        self._total_epochs += 1
        if self._graphs:
            self.reset_graphs()
        return self._epoch_by_value()

    AUTO_PROBE = 8  # by-value epochs timed before the "auto" policy decides
    AUTO_CLEAR = 0.4  # host enqueue time / device time of an epoch below which the device bounds the fit for certain
    AUTO_MARGIN = 0.95  # replayed epochs stay if they take at most this fraction of the by-value epochs' time

    def _epoch_probe(self):
        """A by-value epoch of the "auto" policy's probe phase: timed on the host and, by an event pair, on the device.
        Where the prior can run beside the likelihood the probe is twice as long and ALTERNATES between two streams and one
        (early epochs run on clocks and caches that are still settling: a drift must hit both forms alike).  After the
        last one (one wait for its event) the faster stream form stays, and the session goes on by value (the device
        clearly bounds the fit: the events are as far apart as the device needs) or captures the epoch and times replays
        (`_trial_replay`)."""
        import time

        two_forms = (self.overlap_mode == "auto" and not self.dist.sharded
                     and any(getattr(p, "supports_phases", False) for p in self.priors))
        if two_forms:
            self.overlap_prior = len(self._probe) % 2 == 0
        start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        start.record()
        t0 = time.perf_counter()
        self._enqueue_epoch(self._plan_by_value())
        host = time.perf_counter() - t0
        end.record()
        self._total_epochs += 1
        self._probe.append((host, start, end))
        if len(self._probe) < self.AUTO_PROBE * (2 if two_forms else 1):
            return
        end.synchronize()

        def medians(probe):  # (the first two of a form still pay for lazy allocations and table uploads)
            return (float(np.median([h for h, _, _ in probe[2:]])),
                    1e-3 * float(np.median([a.elapsed_time(b) for _, a, b in probe[2:]])))

        streams = ""
        if two_forms:
            # (an epoch enqueued by value takes the longer of the host's and the device's time for it)
            both, single = medians(self._probe[0::2]), medians(self._probe[1::2])
            self.overlap_prior = max(both) <= max(single)
            host_s, device_s = both if self.overlap_prior else single
            streams = (f"; prior {'beside the likelihood' if self.overlap_prior else 'behind the likelihood'}: "
                       f"{1e6 * max(both):.0f} us on two streams / {1e6 * max(single):.0f} us on one")
        else:
            host_s, device_s = medians(self._probe)
        self._probe = []
        ratio = host_s / max(device_s, 1e-9)
        if ratio < self.AUTO_CLEAR:
            self.graph_policy = (f"by value (device bound: enqueue {1e6 * host_s:.0f} us / device {1e6 * device_s:.0f} us per epoch"
                                 f"{streams})")
        else:
            # the host takes a good part of the epoch's time: capture, time AUTO_PROBE replays the same way, keep the faster
            self.use_graph = True
            # (replayed, the host's share of the stream forms' cost is gone: both forms are captured and timed in turn)
            forms = [True, False] if two_forms else [self.overlap_prior]
            self._trial = {"by_value": max(device_s, host_s), "by_value_form": self.overlap_prior, "host": host_s, "events": [],
                           "streams": streams, "forms": forms, "times": {}}
            self.overlap_prior = forms[0]
            self.graph_policy = (f"captured epochs (on trial: by value enqueue {1e6 * host_s:.0f} us / device {1e6 * device_s:.0f} us "
                                 f"per epoch{streams})")

    def _trial_replay(self, graph):
        """A replayed epoch of the trial: timed like the probe epochs; after AUTO_PROBE of them per stream form the fastest
        of {by value, replayed on two streams, replayed on one} stays."""
        trial = self._trial
        start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        start.record()
        graph.replay()
        end.record()
        trial["events"].append((start, end))
        if len(trial["events"]) < self.AUTO_PROBE:
            return
        end.synchronize()
        trial["times"][self.overlap_prior] = 1e-3 * float(np.median([a.elapsed_time(b) for a, b in trial["events"][2:]]))
        trial["events"] = []
        remaining = [form for form in trial["forms"] if form not in trial["times"]]
        if remaining:  # the other stream form: captured anew (after its eager epochs) and timed
            self.overlap_prior, self._graphs, self._epochs_done = remaining[0], {}, 0
            return
        times, by_value_s = trial["times"], trial["by_value"]
        best = min(times, key=times.get)
        self._trial = None
        replays = " / ".join(f"{1e6 * t:.0f} us {'on two streams' if form else 'on one stream'}" for form, t in times.items())
        numbers = (f"replayed {replays} / by value {1e6 * by_value_s:.0f} us per epoch, enqueue {1e6 * trial['host']:.0f} us"
                   f"{trial['streams']}")
        # (the by-value epochs were timed first, on a device that was still settling: a replay has to win by a margin)
        if times[best] <= self.AUTO_MARGIN * by_value_s:
            self.graph_policy = f"captured epochs (measured: {numbers})"
            if best != self.overlap_prior:  # (the form captured last lost: capture the other one again)
                self.overlap_prior, self._graphs, self._epochs_done = best, {}, 0
        else:
            self.use_graph = False
            self._graphs = {}
            self.overlap_prior = trial["by_value_form"]
            self.graph_policy = f"by value (measured: {numbers})"

    def _epoch_by_value(self):
        cfg, dist, states, priors, total_loss = self.cfg, self.dist, self.states, self.priors, self.total_loss
        n_d, n_c = self.n_d, self.n_c
        slot = self._slot
        if self.joint:
            # ---- one step on sum_d L_d - beta * logprior ------------------------------------
            fluxes = [st.flux_cur for st in states]
            grads = [st.grad for st in states]
            if dist.sharded:
                self.scalars.zero_()
            first = True
            if self.batch_joint:
                # all local datasets in three launches (forward + Poisson, losses, adjoint): same numbers as the loop
                total_loss.poisson_loss.fwd_bwd_batch(
                    [li for _, li in self.local_idx], fluxes if n_c > 1 else fluxes[0],
                    [slot(gslot) for gslot, _ in self.local_idx], grad=grads if n_c > 1 else grads[0], accumulate=False,
                    flux_nonneg=self.flux_nonneg,
                )
                first = False
            elif self.batch_joint_calibrated:
                for _, li in self.local_idx:
                    self._cal_zero_grad(li)
                total_loss.poisson_loss.fwd_bwd_batch_calibrated(
                    [li for _, li in self.local_idx], fluxes[0], [slot(gslot) for gslot, _ in self.local_idx], grad=grads[0],
                    accumulate=False,
                )
                first = False
            else:
                for gslot, li in self.local_idx:
                    self._cal_zero_grad(li)
                    total_loss.poisson_loss.fwd_bwd(li, fluxes, slot(gslot), grads=grads, accumulate=not first,
                                                    flux_nonneg=self.flux_nonneg)
                    first = False
            if first:
                for g in grads:
                    g.zero_()
            banded = {item["ci"]: item for item in (self.band_plan or [])}
            stepped = set()
            for ci, (st, prior) in enumerate(zip(states, priors)):
                if ci in banded:
                    continue
                if dist.sharded and not prior.shardable and dist.rank != 0:
                    continue  # cheap element-wise priors: rank 0 only, summed by the all-reduce
                if self._fuse_step(st, prior):
                    # single process: the prior is the last gradient term of its component -- its gather kernel applies
                    # the optimizer step (one pass over the gradient image and one launch less)
                    prior.device_fwd_bwd_step(st.flux_cur, slot(n_d + ci), -float(cfg.beta), cfg._step_args(st, self.step + 1))
                    stepped.add(ci)
                    continue
                prior.device_fwd_bwd(
                    st.flux_cur, slot(n_d + ci), grad=st.grad, coef=-float(cfg.beta),
                    patch_rows=self._prior_rows(prior, st),
                )
            if banded:
                # likelihood gradient (+ element-wise priors, + dataset losses) on its way while the prior is evaluated
                from .ops import add_rolled_bands

                pending = dist.all_reduce_sum_async(self.comm)
                for ci, item in banded.items():
                    st, prior = states[ci], priors[ci]
                    prior.device_fwd_bwd(
                        st.flux_cur, self.band_send[item["value"] : item["value"] + 1], coef=-float(cfg.beta),
                        patch_rows=item["rows"], band_out=self.band_send[item["offset"] : item["offset"] + item["size"]],
                    )
                    item["shifts"] = prior.last_shifts
                self._timed("all_gather_bands", lambda: dist.all_gather_flat(self.band_recv, self.band_send))
                if pending is not None:
                    self._timed("all_reduce_wait", pending.wait)
                pieces = self.band_recv.view(dist.world_size, self.band_chunk)
                for ci, item in banded.items():
                    if self._fuse_band_step(states[ci]):
                        # the bands are the last term of this component's gradient: their sum and the optimizer step in
                        # one launch (the gradient image is read once, never written)
                        from .ops import add_rolled_bands_step

                        add_rolled_bands_step(states[ci].grad.shape, item["shifts"], self.band_recv[item["offset"] :],
                                              self.band_chunk, item["y_ranges"], cfg._step_args(states[ci], self.step + 1))
                        stepped.add(ci)
                    else:
                        add_rolled_bands(states[ci].grad, item["shifts"], self.band_recv[item["offset"] :], self.band_chunk,
                                         item["y_ranges"])
                    # every rank sums the shard values in rank order: identical replicas
                    if dist.dry_run:
                        slot(n_d + ci).copy_(pieces[dist.rank, item["value"] : item["value"] + 1])
                    else:
                        torch.sum(pieces[:, item["value"]], dim=0, keepdim=True, out=slot(n_d + ci))
            elif dist.sharded:
                self._timed("all_reduce_blocking", lambda: dist.all_reduce_sum(self.comm))
            self.step += 1
            self._apply_step(states, stepped)
            self._cal_step_all()
        else:
            # ---- the reference loop: one step per dataset (core.py:214-229) -------------------
            for gslot, li in self.local_idx:
                fluxes = [st.flux_cur for st in states]
                grads = [st.grad for st in states]
                self._cal_zero_grad(li)
                total_loss.poisson_loss.fwd_bwd(li, fluxes, slot(gslot), grads=grads, accumulate=False,
                                                flux_nonneg=self.flux_nonneg)
                coef = -float(cfg.beta) / total_loss.prior_weight
                stepped = set()
                for ci, (st, prior) in enumerate(zip(states, priors)):
                    if self._fuse_step(st, prior):
                        prior.device_fwd_bwd_step(st.flux_cur, slot(n_d + ci), coef, cfg._step_args(st, self.step + 1))
                        stepped.add(ci)
                    else:
                        prior.device_fwd_bwd(st.flux_cur, slot(n_d + ci), grad=st.grad, coef=coef)
                self.step += 1
                self._apply_step(states, stepped)
                self._cal_step(li)
            # ---- trace on the STALE fluxes of the last step (core.py:247) ---------------------
            stale = [st.flux_trace for st in states]
            if self.batch_trace:
                total_loss.poisson_loss.fwd_bwd_batch(
                    [li for _, li in self.local_idx], stale if n_c > 1 else stale[0],
                    [slot(gslot) for gslot, _ in self.local_idx], flux_nonneg=self.flux_nonneg,
                )
            else:
                for gslot, li in self.local_idx:
                    total_loss.poisson_loss.fwd_bwd(li, stale, slot(gslot), flux_nonneg=self.flux_nonneg)
            for ci, (st, prior) in enumerate(zip(states, priors)):
                prior.device_fwd_bwd(st.flux_trace, slot(n_d + ci))
        if self.n_val:
            # validation losses on the same fluxes the trace row refers to
            vfl = [st.flux_prev if self.joint else st.flux_trace for st in states]
            for vi in range(self.n_val):
                total_loss.poisson_loss_validation.fwd_bwd(vi, vfl, slot(n_d + n_c + vi))


class MAPDeconvolverResult:
    """MAP deconvolver result (reference: jolideco/core.py:285-378)."""

    def __init__(
        self, config, components, trace_loss, components_init=None, calibrations=None, calibrations_init=None,
        wcs=None,
    ):
        self._components = components
        self._components_init = components_init
        self.trace_loss = trace_loss
        self._calibrations = calibrations
        self._calibrations_init = calibrations_init
        self._config = config
        self._wcs = wcs

    @property
    def components(self):
        return self._components

    @property
    def components_init(self):
        return self._components_init

    @property
    def calibrations(self):
        return self._calibrations

    @property
    def calibrations_init(self):
        return self._calibrations_init

    @property
    def flux_total(self):
        return self.components.flux_total_numpy

    @property
    def flux_upsampled_total(self):
        return self.components.flux_upsampled_total_numpy

    @property
    def config(self):
        return self._config

    @property
    def checkpoint_path(self):
        """Path to checkpoints"""
        return Path(self.config.get("checkpoint_path", None))

    def read_checkpoint(self, epoch):
        """Read the checkpoint of an epoch (reference: core.py:329-343) -> `MAPDeconvolverResult`."""
        filename = self.checkpoint_path / str(self.trace_loss["filename"][epoch])
        return self.__class__.read(filename=filename)

    @property
    def config_table(self):
        """Configuration as a one-row table (reference: core.py:425-433)."""
        from .utils.io.fits import config_to_table

        return config_to_table(self.config)

    def write(self, filename, overwrite=False, format=None):
        """Write the result to file.

        format : {"fits", "asdf", "npz"}
            "fits" and "asdf" are the reference's layouts (jolideco/utils/io/fits.py:421-459,
            asdf.py:112-142); "npz" a compact numpy archive (fluxes, trace, calibrations).  Default: from the
            file suffix.
        """
        from .utils.io import IO_FORMATS_MAP_RESULT_WRITE, get_writer

        registry = dict(IO_FORMATS_MAP_RESULT_WRITE, npz=_write_map_result_to_npz)
        if format is None and Path(filename).suffix == ".npz":
            format = "npz"
        writer = get_writer(filename=filename, format=format, registry=registry)
        writer(result=self, filename=filename, overwrite=overwrite)

    @classmethod
    def read(cls, filename, format=None):
        """Read a result written by `write` (or, for "fits", by the reference)."""
        from .utils.io import IO_FORMATS_MAP_RESULT_READ, get_reader

        registry = dict(IO_FORMATS_MAP_RESULT_READ, npz=_read_map_result_from_npz)
        if format is None and Path(filename).suffix == ".npz":
            format = "npz"
        reader = get_reader(filename=filename, format=format, registry=registry)
        return reader(filename=filename)


def _write_map_result_to_npz(result, filename, overwrite):
    """Fluxes, loss trace and calibrations as a numpy ``.npz`` archive."""
    filename = Path(filename)
    if filename.exists() and not overwrite:
        raise OSError(f"{filename} exists")
    arrays = {f"flux/{name}": flux for name, flux in result.components.to_numpy().items()}
    for name, component in result.components.items():
        arrays[f"meta/{name}"] = np.array(
            [component.upsampling_factor or 0, int(component.use_log_flux), int(component.frozen)]
        )
    for name in result.trace_loss.colnames:
        if name != "filename":
            arrays[f"trace/{name}"] = result.trace_loss[name]
    if result.calibrations is not None:
        for name, cal in result.calibrations.to_dict().items():
            arrays[f"calibration/{name}"] = np.array(
                [cal["shift_x"], cal["shift_y"], cal["background_norm"], cal["psf_scale"], float(cal["frozen"])]
            )
    np.savez_compressed(filename, **arrays)


def _read_map_result_from_npz(filename):
    """Read an ``.npz`` result: fluxes (priors are not stored and default to uniform), loss trace and
    calibrations."""
    from .models import NPredCalibration, NPredCalibrations
    from .utils.table import TraceTable

    data = dict(np.load(filename))
    components = FluxComponents()
    for key, flux in data.items():
        if key.startswith("flux/"):
            name = key[len("flux/"):]
            up, use_log, frozen = (int(v) for v in data.get(f"meta/{name}", np.array([0, 1, 0])))
            component = SpatialFluxComponent(
                flux_upsampled=torch.from_numpy(np.asarray(flux, dtype=np.float32))[None, None],
                use_log_flux=bool(use_log), upsampling_factor=up or None, frozen=bool(frozen),
            )
            components[name] = component
    names = [key[len("trace/"):] for key in data if key.startswith("trace/")]
    trace = TraceTable(names=names + ["filename"])
    n_rows = len(data[f"trace/{names[0]}"]) if names else 0
    for i in range(n_rows):
        row = {name: float(data[f"trace/{name}"][i]) for name in names}
        row["filename"] = ""
        trace.add_row(row)
    calibrations = None
    cal_keys = [key for key in data if key.startswith("calibration/")]
    if cal_keys:
        calibrations = NPredCalibrations()
        for key in cal_keys:
            sx, sy, norm, psf_scale, frozen = (float(v) for v in data[key])
            calibrations[key[len("calibration/"):]] = NPredCalibration(sx, sy, norm, psf_scale, bool(frozen))
    return MAPDeconvolverResult(config={}, components=components, trace_loss=trace, calibrations=calibrations)
