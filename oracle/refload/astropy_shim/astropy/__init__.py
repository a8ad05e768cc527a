"""Import-only stand-in for astropy (absent for python3.10 in this image).

TEST INFRASTRUCTURE ONLY. Lets `/root/reference/jolideco` be *imported* in the build
container so that golden vectors can be generated from the reference's own hot path
(oracle/refload/make_golden.py). None of the hot-path arithmetic goes through this shim:
`Table` is a row container for the loss trace and `lazyproperty` is functools.cached_property.
The convolution kernels below are used only by the reference's synthetic *data* helpers
(jolideco/data/core.py); they are pinned by the reference's known answers in
jolideco/data/tests/test_core.py (psf[7][7] = 0.015965 etc.).
"""
__version__ = "0.0-shim"
