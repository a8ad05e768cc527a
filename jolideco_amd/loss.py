"""Loss assembly (reference: jolideco/loss.py).

`PoissonLoss`, `PriorLoss` and `TotalLoss` keep the reference's attributes and call signatures.
Each has the autograd-visible methods of the reference (`evaluate`, `__call__`, `loss_function`)
and a fused device path (`*_fwd_bwd`) that the fit loop uses: values land in device scalars, the
gradients are accumulated by the HIP kernels without autograd.
"""
import os

import numpy as np
import torch

from .models import NPredModels
from .models.npred import common_kernel_shape
from .ops import PoissonNLLFunction, stirling_mean
from .utils.table import TraceTable
from .utils.torch import TORCH_DEFAULT_DEVICE

__all__ = ["PoissonLoss", "PriorLoss", "TotalLoss"]


class _PoissonNLLLoss:
    """Callable equivalent of nn.PoissonNLLLoss(log_input=False, reduction="mean", eps=1e-25,
    full=True) (jolideco/loss.py:35-37) running on the HIP kernel; the flux independent Stirling
    mean is cached per counts tensor."""

    def __init__(self):
        self._stirling = {}

    def stirling(self, counts):
        key = (counts.data_ptr(), counts.numel())
        if key not in self._stirling:
            self._stirling[key] = stirling_mean(counts.detach().cpu().numpy())
        return self._stirling[key]

    def __call__(self, npred, counts):
        return PoissonNLLFunction.apply(npred, counts, self.stirling(counts))


class PoissonLoss:
    """Poisson loss of all datasets.

    Attributes
    ----------
    counts_all : list of `~torch.Tensor`  (1, 1, H, W) on the device
    npred_models_all : list of `NPredModels`
    names_all : list of str
    """

    def __init__(self, counts_all, npred_models_all, names_all):
        if len(counts_all) != len(npred_models_all):
            raise ValueError("counts_all and npred_models_all must have the same length")
        self.counts_all = counts_all
        self.npred_models_all = npred_models_all
        self.names_all = names_all
        self.loss_function = _PoissonNLLLoss()
        self.stirling_all = [self.loss_function.stirling(c) for c in counts_all]

    @property
    def n_datasets(self):
        return len(self.counts_all)

    @property
    def iter_by_dataset(self):
        for data in zip(self.counts_all, self.npred_models_all):
            yield data

    def evaluate(self, fluxes):
        """Per-dataset losses as a detached tensor (jolideco/loss.py:56-71)."""
        out = torch.empty(self.n_datasets, dtype=torch.float32, device=self.counts_all[0].device)
        with torch.no_grad():
            for idx in range(self.n_datasets):
                self.fwd_bwd(idx, [f.detach().reshape(f.shape[-2:]) for f in fluxes], out[idx : idx + 1])
        return out

    def __call__(self, fluxes):
        return torch.sum(self.evaluate(fluxes=fluxes))

    def fwd_bwd(self, idx, fluxes, loss_out, grads=None, accumulate=False, grad_scale=1.0, npred_out=None,
                flux_nonneg=False):
        """Fused forward model + Poisson NLL (+ gradient) of dataset ``idx``."""
        self.npred_models_all[idx].fwd_bwd(
            fluxes, self.counts_all[idx], self.stirling_all[idx], loss_out, grads=grads, accumulate=accumulate,
            grad_scale=grad_scale, npred_out=npred_out, flux_nonneg=flux_nonneg,
        )

    def batchable(self, indices):
        """True if the datasets `indices` can take the batched joint step: at most 4 flux components, no up-sampling,
        no calibration, ONE plan shared by every (dataset, component) model -- a separable plan, or (one flux
        component) a plan of the native FFT convolution."""
        models_all = [self.npred_models_all[i] for i in indices]
        if len(models_all) < 2:
            return False
        plans, methods = set(), set()
        for models in models_all:
            if not 1 <= len(models) <= 4 or models.calibration is not None:
                return False
            for model in models.values():
                if (model.upsampling_factor or 1) != 1:
                    return False
                plans.add(id(model.plan))
                methods.add(model.plan.method)
        if len(plans) != 1:
            return False
        if methods == {"separable"}:
            return True
        plan = next(iter(models_all[0].values())).plan
        return methods == {"fft"} and bool(plan.native_fft) and all(len(models) == 1 for models in models_all)

    def batchable_calibrated(self, indices):
        """True if the datasets `indices` can take the batched CALIBRATED / UP-SAMPLED joint step
        (`fwd_bwd_batch_calibrated`): one flux component, ONE plan of the native FFT convolution shared by all datasets,
        one up-sampling factor (2, 3 or 4), each dataset with or without a calibration."""
        models_all = [self.npred_models_all[i] for i in indices]
        if len(models_all) < 2 or any(len(models) != 1 for models in models_all):
            return False
        first = next(iter(models_all[0].values()))
        u = first.upsampling_factor or 1
        if u not in (2, 3, 4) or first.plan.method != "fft" or not first.plan.native_fft:
            return False
        return all(
            (m.upsampling_factor or 1) == u and m.plan is first.plan for models in models_all for m in models.values()
        )

    def fwd_bwd_batch_calibrated(self, indices, flux, loss_outs, grad=None, accumulate=False, grad_scale=1.0):
        """The batched joint step of calibrated / up-sampled datasets (requires `batchable_calibrated(indices)`): the
        numbers of `fwd_bwd` per dataset with ``accumulate`` from the second dataset on."""
        per_dataset = [self.npred_models_all[i] for i in indices]
        models = [next(iter(mm.values())) for mm in per_dataset]
        models[0].plan.npred_poisson_calibrated_batch_fwd_bwd(
            flux=flux, exposures=[m.exposure for m in models], khats=[m.khat for m in models],
            backgrounds=[mm.background for mm in per_dataset], counts=[self.counts_all[i] for i in indices],
            stirlings=[self.stirling_all[i] for i in indices], loss_outs=loss_outs,
            calibrations=[mm.calibration_pointers(grad is not None) for mm in per_dataset],
            upsampling=models[0].upsampling_factor or 1, grad=grad, accumulate=accumulate, grad_scale=grad_scale,
        )

    def mergeable(self, indices):
        """True if every dataset of `indices` evaluates all its flux components through ONE non-negative forward
        operator (`NPredModels.shared_operator`): the batched step may then run on the SUM of the component fluxes.
        JOLIDECO_MERGE_COMPONENTS=0: never (testing / A-B)."""
        key = tuple(indices)
        cache = self.__dict__.setdefault("_mergeable", {})
        if key not in cache:
            cache[key] = os.environ.get("JOLIDECO_MERGE_COMPONENTS", "1") != "0" and all(
                getattr(self.npred_models_all[i], "shared_operator", False) for i in indices
            )
        return cache[key]

    def fwd_bwd_batch(self, indices, flux, loss_outs, grad=None, accumulate=False, grad_scale=1.0, flux_nonneg=False):
        """Forward model + Poisson NLL (+ gradient, summed over the datasets in order) of the datasets `indices`
        in three launches (+ one adjoint launch per further component); requires `batchable(indices)`.
        ``flux`` / ``grad``: a tensor (one component) or lists with one tensor per component, in component order.
        ``flux_nonneg``: the caller guarantees flux >= 0 everywhere (exp(theta) [x mask]); together with
        `mergeable(indices)` the components are then evaluated as ONE flux image, their sum: one forward model and one
        adjoint per dataset whatever the number of components (models/npred.py:241-261: no term is clipped)."""
        per_dataset = [list(self.npred_models_all[i].values()) for i in indices]
        single = torch.is_tensor(flux)
        if not single and len(flux) > 1 and flux_nonneg and not accumulate and self.mergeable(indices):
            from .ops import copy_image_to, sum_images

            total = self.__dict__.get("_merged_flux")
            if total is None or total.shape != flux[0].shape or total.device != flux[0].device:
                total = self._merged_flux = torch.empty_like(flux[0])
            sum_images(total, list(flux))
            per_dataset[0][0].plan.npred_poisson_batch_fwd_bwd(
                flux=total, exposures=[models[0].exposure for models in per_dataset],
                khats=[models[0].khat for models in per_dataset],
                backgrounds=[self.npred_models_all[i].background for i in indices],
                counts=[self.counts_all[i] for i in indices], stirlings=[self.stirling_all[i] for i in indices],
                loss_outs=loss_outs, grad=None if grad is None else grad[0], accumulate=False, grad_scale=grad_scale,
            )
            if grad is not None:
                copy_image_to(grad[0], list(grad[1:]))
            return
        exposures = [[m.exposure for m in models] for models in per_dataset]
        khats = [[m.khat for m in models] for models in per_dataset]
        if single:
            exposures, khats = [e[0] for e in exposures], [k[0] for k in khats]
        per_dataset[0][0].plan.npred_poisson_batch_fwd_bwd(
            flux=flux, exposures=exposures, khats=khats,
            backgrounds=[self.npred_models_all[i].background for i in indices],
            counts=[self.counts_all[i] for i in indices], stirlings=[self.stirling_all[i] for i in indices],
            loss_outs=loss_outs, grad=grad, accumulate=accumulate, grad_scale=grad_scale,
        )

    @classmethod
    def from_datasets(cls, datasets, components, calibrations=None, device=TORCH_DEFAULT_DEVICE):
        npred_models_all, counts_all = [], []
        kernel_shape = common_kernel_shape(datasets, components, calibrations)
        for name, dataset in datasets.items():
            calibration = calibrations[name] if calibrations else None  # KeyError for an unknown name, like the reference
            models = NPredModels.from_dataset_numpy(
                dataset=dataset, components=components, calibration=calibration, device=device, kernel_shape=kernel_shape
            )
            npred_models_all.append(models)
            counts = torch.from_numpy(np.ascontiguousarray(dataset["counts"], dtype=np.float32)[None, None])
            counts_all.append(counts.to(device))
        return cls(counts_all=counts_all, npred_models_all=npred_models_all, names_all=list(datasets))


class PriorLoss:
    """Prior loss: one prior per flux component (jolideco/loss.py:136-168)."""

    def __init__(self, priors):
        self.priors = priors

    def evaluate(self, fluxes):
        return [prior(flux=flux) for flux, prior in zip(fluxes, self.priors.values())]

    def __call__(self, fluxes):
        return sum(self.evaluate(fluxes=fluxes))


class TotalLoss:
    """Total loss = sum of dataset losses - beta * sum of log-priors."""

    def __init__(self, poisson_loss, prior_loss, poisson_loss_validation=None, beta=1):
        self.poisson_loss = poisson_loss
        self.poisson_loss_validation = poisson_loss_validation
        self.prior_loss = prior_loss
        self.beta = beta
        self._trace = None

    @property
    def trace_names(self):
        """Column layout of the reference's trace table (jolideco/loss.py:192-210)."""
        names = ["total", "datasets-total", "priors-total"]
        names += [f"prior-{name}" for name in self.prior_loss.priors]
        # in a sharded joint fit a rank holds only its own datasets but the trace lists all of them
        names_d = getattr(self.poisson_loss, "names_all_global", None) or self.poisson_loss.names_all
        names += [f"dataset-{name}" for name in names_d]
        if self.poisson_loss_validation:
            names += ["datasets-validation-total"]
        names += ["filename"]
        return names

    @property
    def trace(self):
        if self._trace is None:
            self._trace = TraceTable(names=self.trace_names)
        return self._trace

    @property
    def prior_weight(self):
        """Number of datasets (jolideco/loss.py:252-255)."""
        return len(self.poisson_loss.counts_all)

    def make_row(self, loss_datasets, loss_priors, filename="", loss_validation=None):
        """Trace row with the reference's names and signs (jolideco/loss.py:226-250)."""
        loss_datasets_total = sum(loss_datasets)
        loss_priors_total = self.beta * sum(loss_priors)
        row = {
            "total": loss_datasets_total - loss_priors_total,
            "datasets-total": loss_datasets_total,
            "priors-total": -loss_priors_total,
            "filename": filename,
        }
        for name, value in zip(self.prior_loss.priors, loss_priors):
            row[f"prior-{name}"] = -self.beta * value
        for name, value in zip(self.poisson_loss.names_all, loss_datasets):
            row[f"dataset-{name}"] = value
        if loss_validation is not None:
            row["datasets-validation-total"] = sum(loss_validation)
        return row

    @torch.no_grad()
    def append_trace(self, fluxes, filename=""):
        """Re-evaluate every dataset loss and every prior on ``fluxes`` and append one row
        (jolideco/loss.py:212-250).  Consumes one cycle-spin draw per GMM prior, like the reference."""
        loss_datasets = [float(v) for v in self.poisson_loss.evaluate(fluxes=fluxes).cpu()]
        loss_priors = [float(torch.as_tensor(v)) for v in self.prior_loss.evaluate(fluxes=fluxes)]
        loss_validation = None
        if self.poisson_loss_validation:
            loss_validation = [float(v) for v in self.poisson_loss_validation.evaluate(fluxes=fluxes).cpu()]
        self.trace.add_row(self.make_row(loss_datasets, loss_priors, filename, loss_validation))

    def __call__(self, fluxes):
        loss_datasets = self.poisson_loss.evaluate(fluxes=fluxes)
        loss_priors = self.prior_loss.evaluate(fluxes=fluxes)
        return sum(loss_datasets) - self.beta * sum(loss_priors)

    def hessian_diagonals(self, fluxes):
        """What the reference's `hessian_diagonals` returns (jolideco/loss.py:263-279): the Hessian-vector product of
        the total loss with vectors of ones.  There the dataset losses pass through `torch.tensor(...)`
        (loss.py:71) and carry no graph, so ONLY the prior terms contribute: -beta * H_prior x ones per component."""
        # "+ 0.0" turns the -0.0 of a curvature-free prior into the +0.0 autograd fills in there (error = +inf)
        return tuple(
            -self.beta * prior.hessian_ones(flux) + 0.0 for flux, prior in zip(fluxes, self.prior_loss.priors.values())
        )

    def fluxes_error(self, fluxes):
        """Flux errors sqrt(1 / hessian) per component name (jolideco/loss.py:281-300): inf where the prior has no
        curvature, nan where the curvature is negative -- as in the reference."""
        hessians = self.hessian_diagonals(fluxes=fluxes)
        return {name: torch.sqrt(1 / hessian) for name, hessian in zip(self.prior_loss.priors, hessians)}

    @classmethod
    def from_datasets_and_components(
        cls, datasets, components, datasets_validation=None, beta=1, calibrations=None, device=TORCH_DEFAULT_DEVICE
    ):
        poisson_loss = PoissonLoss.from_datasets(
            datasets=datasets, components=components, device=device, calibrations=calibrations
        )
        poisson_loss_validation = None
        if datasets_validation:
            poisson_loss_validation = PoissonLoss.from_datasets(
                datasets=datasets_validation, components=components, calibrations=calibrations, device=device
            )
        return cls(
            poisson_loss=poisson_loss,
            poisson_loss_validation=poisson_loss_validation,
            prior_loss=PriorLoss(priors=components.priors),
            beta=beta,
        )
