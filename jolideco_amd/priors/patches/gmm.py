"""Gaussian mixture model of image patches (reference: jolideco/priors/patches/gmm.py:64-299).

Constants are prepared on the host in float64/float32 exactly as the reference does (scipy
Cholesky -> precision Cholesky -> fp32), then handed to the HIP library which lays them out in
MFMA fragment order.  The trained GMM libraries of the reference ("zoran-weiss", ...) are external
data files that are not part of this repository: `from_registry` / `read` load them from
``$JOLIDECO_GMM_LIBRARY`` exactly as the reference does (gmm.py:301-391,493-508) when the user has them;
`from_numpy` takes explicit arrays.
"""
import json
import os
from dataclasses import dataclass, field
from pathlib import Path
from typing import Optional

import numpy as np
import torch

from ...utils.norms import PatchNorm, SubtractMeanPatchNorm
from ...utils.numpy import compute_precision_cholesky, get_pixel_weights

__all__ = ["GaussianMixtureModel", "GaussianMixtureModelMeta", "GMMNotAvailableError"]


class GMMNotAvailableError(ValueError):
    """The named GMM is not in the user's GMM library (or there is no library)."""


@dataclass
class GaussianMixtureModelMeta:
    """Meta data: ``stride`` selects the overlap pixel weights, ``patch_norm`` the patch
    normalisation (jolideco/priors/patches/gmm.py:24-61)."""

    stride: Optional[int] = None
    patch_norm: PatchNorm = field(default_factory=SubtractMeanPatchNorm)

    @classmethod
    def from_table(cls, table):
        """Meta data of a GMM table file: patch norm from the ``PNPTYPE`` keyword, stride = half the
        patch edge (jolideco/priors/patches/gmm.py:38-61)."""
        patch_norm = PatchNorm.from_dict({"type": table.meta.get("PNPTYPE", "subtract-mean")})
        npix = int(table["means"].shape[-1] ** 0.5)
        return cls(stride=npix // 2, patch_norm=patch_norm)


def get_gmm_registry():
    """Index of the user's GMM library: ``$JOLIDECO_GMM_LIBRARY/jolideco-gmm-library-index.json``
    (jolideco/priors/patches/gmm.py:493-508; read on demand here, at import time there)."""
    path = Path(os.path.expandvars("$JOLIDECO_GMM_LIBRARY/jolideco-gmm-library-index.json"))
    if not path.exists():
        return {}
    with path.open() as f:
        return json.load(f)


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
        self.registry_name = None
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
        """Load a trained GMM by its name in the user's library index (gmm.py:301-335)."""
        registry = get_gmm_registry()
        if name not in registry:
            raise GMMNotAvailableError(
                f"Not a supported GMM {name}, choose from {list(registry)} (the index is "
                "$JOLIDECO_GMM_LIBRARY/jolideco-gmm-library-index.json; the library files are external data)"
            )
        kwargs.update(registry[name])
        gmm = cls.read(**kwargs)
        gmm.registry_name = name
        return gmm

    @classmethod
    def read(cls, filename, format="epll-matlab", **kwargs):
        """Read a trained GMM (gmm.py:336-391).

        format : {"epll-matlab", "epll-matlab-16x16", "table"}
            Zoran & Weiss EPLL ``.mat`` files (through scipy.io) or a FITS table with ``means`` (K, D),
            ``weights`` (K,) and ``covariances`` (K, D, D) columns.
        """
        filename = str(Path(os.path.expandvars(str(filename))))
        if format in ("epll-matlab", "epll-matlab-16x16"):
            import scipy.io as sio

            record = sio.loadmat(filename)["GS" if format == "epll-matlab" else "GMM"]
            covariances = record["covs"][0][0].T
            weights = record["mixweights"][0][0][:, 0]
            if format == "epll-matlab":
                means = record["means"][0][0].T
                meta = GaussianMixtureModelMeta(stride=4, patch_norm=SubtractMeanPatchNorm())
            else:
                means = np.zeros((200, 256))
                meta = GaussianMixtureModelMeta(stride=8, patch_norm=SubtractMeanPatchNorm())
        elif format == "table":
            from ...utils.io._fitsfile import read_fits

            tables = [hdu.data for hdu in read_fits(filename) if hdu.kind == "bintable"]
            if not tables:
                raise ValueError(f"{filename} holds no table")
            table = tables[0]
            means, weights, covariances = table["means"], table["weights"], table["covariances"]
            meta = GaussianMixtureModelMeta.from_table(table)
        else:
            raise ValueError(f"Not a supported format {format}")
        return cls.from_numpy(means=means, covariances=covariances, weights=weights, meta=meta, **kwargs)

    def write(self, filename, overwrite=False):
        """Write the model as a FITS table that `read(format="table")` -- of this package and of the
        reference -- loads back."""
        from ...utils.io._fitsfile import HDU, FitsTable, write_fits

        table = FitsTable(
            {
                "means": self.means_numpy.astype(np.float64),
                "weights": self.weights_numpy.astype(np.float64),
                "covariances": self.covariances_numpy.astype(np.float64),
            },
            meta={"PNPTYPE": self.meta.patch_norm.to_dict()["type"]},
        )
        write_fits(filename, [HDU(kind="primary"), HDU(table, name="GMM")], overwrite=overwrite)

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
        """``{"type": <name in the GMM library>}`` like the reference (gmm.py:458-471); a model built from
        explicit arrays has no name and is recorded as "custom" with its size."""
        if self.registry_name is not None:
            return {"type": self.registry_name}
        return {"type": "custom", "n_components": int(self.n_components), "n_features": int(self.n_features)}

    @classmethod
    def from_dict(cls, data):
        return cls.from_registry(name=data["type"])
