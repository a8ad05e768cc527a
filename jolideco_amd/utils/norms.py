"""Image / patch normalisations used by the GMM patch prior.

Only the normalisations on the accelerated hot path are implemented: `IdentityImageNorm`
(reference default, jolideco/utils/norms.py:225-232) and `SubtractMeanPatchNorm`
(:97-103, fused into the HIP kernel).  Requesting any other norm raises NotImplementedError.
"""

__all__ = [
    "PatchNorm",
    "SubtractMeanPatchNorm",
    "ImageNorm",
    "IdentityImageNorm",
    "NORMS_REGISTRY",
    "NORMS_PATCH_REGISTRY",
]


class PatchNorm:
    """Patch normalisation base class"""

    def to_dict(self):
        for name, cls in NORMS_PATCH_REGISTRY.items():
            if isinstance(self, cls):
                return {"type": name}
        return {}

    @classmethod
    def from_dict(cls, data):
        kwargs = dict(data)
        if "type" in kwargs:
            type_ = kwargs.pop("type")
            if type_ not in NORMS_PATCH_REGISTRY:
                raise NotImplementedError(f"patch norm {type_!r} is not implemented in jolideco_amd")
            return NORMS_PATCH_REGISTRY[type_](**kwargs)
        return cls(**kwargs)


class SubtractMeanPatchNorm(PatchNorm):
    """Subtract the patch mean (Zoran & Weiss).  On device this is fused into the GMM kernel
    (csrc/gmm.hip); this host version works on torch tensors for explicit patch arrays."""

    def __call__(self, patches):
        return patches - patches.nanmean(dim=1, keepdim=True)


class ImageNorm:
    """Image normalisation base class"""

    def __init__(self, frozen=False):
        self.frozen = frozen

    def to_dict(self):
        for name, cls in NORMS_REGISTRY.items():
            if isinstance(self, cls):
                return {"type": name}
        return {}

    @classmethod
    def from_dict(cls, data):
        kwargs = dict(data)
        if "type" in kwargs:
            type_ = kwargs.pop("type")
            if type_ not in NORMS_REGISTRY:
                raise NotImplementedError(f"image norm {type_!r} is not implemented in jolideco_amd")
            return NORMS_REGISTRY[type_](**kwargs)
        return cls(**kwargs)


class IdentityImageNorm(ImageNorm):
    """Identity image norm"""

    def __call__(self, image):
        return image

    def inverse(self, image):
        return image


NORMS_REGISTRY = {"identity": IdentityImageNorm}
NORMS_PATCH_REGISTRY = {"subtract-mean": SubtractMeanPatchNorm}
