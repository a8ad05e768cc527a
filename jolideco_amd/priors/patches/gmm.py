"""Gaussian mixture model of image patches (reference: jolideco/priors/patches/gmm.py:64-299).

Constants are prepared on the host in float64/float32 exactly as the reference does (scipy
Cholesky -> precision Cholesky -> fp32), then handed to the HIP library which lays them out in
MFMA fragment order.  The trained GMM libraries of the reference ("zoran-weiss", ...) are external
data files that are not part of this repository: use `from_numpy`.
"""
from dataclasses import dataclass, field
from typing import Optional

import numpy as np
import torch

from ...utils.norms import PatchNorm, SubtractMeanPatchNorm
from ...utils.numpy import compute_precision_cholesky, get_pixel_weights

__all__ = ["GaussianMixtureModel", "GaussianMixtureModelMeta"]


@dataclass
class GaussianMixtureModelMeta:
    """Meta data: ``stride`` selects the overlap pixel weights, ``patch_norm`` the patch
    normalisation (jolideco/priors/patches/gmm.py:24-61)."""

    stride: Optional[int] = None
    patch_norm: PatchNorm = field(default_factory=SubtractMeanPatchNorm)


class GaussianMixtureModel:
    """Gaussian mixture model with full covariances.

    Parameters: fp32 numpy arrays ``means`` (K, D), ``covariances`` (K, D, D), ``weights`` (K,),
    ``precisions_cholesky`` (K, D, D).
    """

    def __init__(self, means, covariances, weights, precisions_cholesky, meta=None):
        self.means_numpy = np.asarray(means, dtype=np.float32)
        self.covariances_numpy = np.asarray(covariances, dtype=np.float32)
        self.weights_numpy = np.asarray(weights, dtype=np.float32)
        self.precisions_cholesky_numpy = np.asarray(precisions_cholesky, dtype=np.float32)
        self.meta = meta or GaussianMixtureModelMeta()
        self._handles = {}

    @classmethod
    def from_numpy(cls, means, covariances, weights, meta=None):
        """Create from float64 numpy arrays (jolideco/priors/patches/gmm.py:119-149)."""
        precisions_cholesky = compute_precision_cholesky(covariances=np.asarray(covariances))
        return cls(
            means=np.asarray(means).astype(np.float32),
            covariances=np.asarray(covariances).astype(np.float32),
            weights=np.asarray(weights).astype(np.float32),
            precisions_cholesky=precisions_cholesky.astype(np.float32),
            meta=meta,
        )

    @classmethod
    def from_sklearn_gmm(cls, gmm):
        return cls.from_numpy(means=gmm.means_, covariances=gmm.covariances_, weights=gmm.weights_)

    @classmethod
    def from_registry(cls, name, **kwargs):
        raise NotImplementedError(
            "the trained GMM library files of the reference are external data that is not available here; "
            "construct the model with GaussianMixtureModel.from_numpy(means, covariances, weights)"
        )

    @property
    def n_components(self):
        return self.covariances_numpy.shape[0]

    @property
    def n_features(self):
        return self.covariances_numpy.shape[1]

    @property
    def patch_shape(self):
        npix = int(self.means_numpy.shape[-1] ** 0.5)
        return npix, npix

    # fp32 constants, computed with the same fp32 torch ops as the reference ------------------
    @property
    def means_precisions_cholesky_numpy(self):
        """mu_k @ P_k in fp32 (gmm.py:217-228)."""
        mu = torch.from_numpy(self.means_numpy)
        pc = torch.from_numpy(self.precisions_cholesky_numpy)
        return torch.stack([torch.matmul(m, p) for m, p in zip(mu, pc)]).numpy()

    @property
    def log_det_cholesky_numpy(self):
        """sum_i log P_k[i, i] in fp32 (gmm.py:235-240)."""
        pc = torch.from_numpy(self.precisions_cholesky_numpy)
        diag = pc.reshape(self.n_components, -1)[:, :: self.n_features + 1]
        return torch.sum(torch.log(diag), axis=1).numpy()

    @property
    def log_weights_numpy(self):
        return torch.log(torch.from_numpy(self.weights_numpy)).numpy()

    @property
    def pixel_weights_numpy(self):
        """(1, D) overlap weights; all ones when ``meta.stride`` is None (gmm.py:290-299)."""
        if self.meta.stride is None:
            weights = np.ones(self.patch_shape)
        else:
            weights = get_pixel_weights(patch_shape=self.patch_shape, stride=self.meta.stride)
        return weights.reshape((1, -1))

    # device side ---------------------------------------------------------------------------
    def handle(self, device):
        """Native handle for ``device`` (created on first use, cached)."""
        from ...ops import GmmHandle

        device = torch.device(device)
        if device.type == "cuda" and device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        key = str(device)
        if key not in self._handles:
            self._handles[key] = GmmHandle(
                precisions_cholesky=self.precisions_cholesky_numpy,
                means_precisions_cholesky=self.means_precisions_cholesky_numpy,
                log_det_cholesky=self.log_det_cholesky_numpy,
                log_weights=self.log_weights_numpy,
                pixel_weights=self.pixel_weights_numpy.astype(np.float32),
                device=device,
            )
        return self._handles[key]

    def estimate_log_prob(self, x):
        """(n, K) weighted log-probabilities of already normalised patches ``x`` (n, D) on the GPU
        (jolideco/priors/patches/gmm.py:262-281)."""
        if not (isinstance(x, torch.Tensor) and x.is_cuda):
            raise RuntimeError("x must be a HIP tensor: jolideco_amd has no CPU path")
        return self.handle(x.device).estimate_log_prob(x.contiguous())

    def __deepcopy__(self, memo):
        # constants are immutable: share them (and the native handles) between copies
        return self

    def to_dict(self):
        return {"type": "custom", "n_components": int(self.n_components), "n_features": int(self.n_features)}
