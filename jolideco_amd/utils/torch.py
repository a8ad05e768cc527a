"""Tensor utilities with the reference's names (jolideco/utils/torch.py), backed by the HIP library."""
import logging

import torch

__all__ = [
    "TORCH_DEFAULT_DEVICE",
    "convolve_fft_torch",
    "cycle_spin_shifts",
    "cycle_spin",
    "get_default_generator",
    "view_as_overlapping_patches_torch",
]

log = logging.getLogger(__name__)

# The reference defaults to "cpu" (utils/torch.py:21); this package only runs on the accelerator.
TORCH_DEFAULT_DEVICE = "cuda"


def convolve_fft_torch(image, kernel):
    """'same' FFT convolution of a (1, 1, H, W) image with a (1, 1, kh, kw) kernel on the GPU
    (rocFFT R2C/C2R + HIP k-space multiply); differentiable w.r.t. ``image``.

    Same contract as jolideco/utils/torch.py:347-370.
    """
    from ..ops import ConvPlan, ConvSameFunction, require_hip_tensor

    image_c = require_hip_tensor(image, "image")
    kernel_c = require_hip_tensor(kernel, "kernel")
    if image_c.numel() != image_c.shape[-2] * image_c.shape[-1]:
        raise NotImplementedError("only single 2-D images (1, 1, H, W) are supported")
    H, W = image_c.shape[-2:]
    kh, kw = kernel_c.shape[-2:]
    plan = ConvPlan.get(H, W, kh, kw, image_c.device)
    khat = plan.psf_spectrum(kernel_c.reshape(kh, kw))
    return ConvSameFunction.apply(image_c, None, khat, plan)


def cycle_spin_shifts(patch_shape, generator):
    """Draw the cycle-spin shifts exactly like jolideco/utils/torch.py:108-116: two `randint`
    draws from a HOST generator in [-p//4, p//4]; the first rolls rows (dim -2), the second
    columns (dim -1).  Drawing on the host keeps runs comparable with the reference CPU path and
    avoids the device->host sync of `int(shift)` in the reference."""
    wy, wx = patch_shape[0] // 4, patch_shape[1] // 4
    first = torch.randint(-wy, wy + 1, (1,), generator=generator)
    second = torch.randint(-wx, wx + 1, (1,), generator=generator)
    return int(first), int(second)


def cycle_spin_shifts_many(patch_shape, generator, n):
    """`n` consecutive draws of `cycle_spin_shifts`, in one `randint` call where both directions share a range (square
    patches): the CPU generator hands out one number per element in order, so `randint(size=(2 n,))` returns the numbers of
    2 n single draws and leaves the generator in the same state (tests/test_host_logic.py holds torch to that)."""
    wy, wx = patch_shape[0] // 4, patch_shape[1] // 4
    if wy != wx or n > 4096:
        return [cycle_spin_shifts(patch_shape, generator) for _ in range(n)]
    values = torch.randint(-wy, wy + 1, (2 * n,), generator=generator).tolist()
    return [(values[2 * i], values[2 * i + 1]) for i in range(n)]


def cycle_spin(image, patch_shape, generator):
    """Rolled copy of ``image`` (jolideco/utils/torch.py:91-119).  The accelerated prior never
    materialises this: the roll is folded into the kernel's addressing."""
    shifts = cycle_spin_shifts(patch_shape, generator)
    return torch.roll(image, shifts=shifts, dims=(image.ndim - 2, image.ndim - 1))


def get_default_generator(device="cpu"):
    """Host generator with torch's default seed (jolideco/utils/torch.py:393-414).  The shifts
    are always drawn on the host, so the device argument is accepted and ignored."""
    return torch.Generator(device="cpu")


def view_as_overlapping_patches_torch(image, shape, stride=None):
    """(n_patches, p*p) view of overlapping patches (jolideco/utils/torch.py:251-275); a torch
    view helper for inspection -- the prior kernel builds patches in registers instead."""
    if stride is None:
        stride = shape[0] // 2
    win = image.unfold(image.ndim - 2, shape[0], stride).unfold(image.ndim - 1, shape[0], stride)
    return win.reshape(-1, shape[0] * shape[1])
