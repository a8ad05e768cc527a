"""Loader that makes `/root/reference/jolideco` importable in the BUILD container only.

TEST INFRASTRUCTURE. Used by make_golden.py (fixture generation) and by
tests that cross-check oracle/cpu_ref.py against the live reference when /root/reference
exists. Nothing here travels as a dependency of the GPU tests, smoke() or bench.py.

Work-arounds:
  1. astropy is absent for python3.10 -> ./astropy_shim (import-only stand-in)
  2. jolideco/__init__.py:11 imports `.version` but the tree only has `_version.py`
  3. jolideco/priors/patches/gmm.py:493-508 opens $JOLIDECO_GMM_LIBRARY/...index.json at import
"""
import os
import sys
import tempfile
import types
from pathlib import Path

REFERENCE_PATH = Path("/root/reference")


def reference_available():
    return (REFERENCE_PATH / "jolideco" / "core.py").exists()


def load_reference():
    if "jolideco" in sys.modules and getattr(sys.modules["jolideco"], "__jd_ref__", False):
        return sys.modules["jolideco"]
    if not reference_available():
        raise RuntimeError("reference tree not present (expected only in the build container)")
    shim = str(Path(__file__).parent / "astropy_shim")
    try:
        import astropy  # noqa: F401
    except ImportError:
        sys.path.insert(0, shim)
    sys.path.insert(0, str(REFERENCE_PATH))
    libdir = Path(tempfile.mkdtemp(prefix="jd-gmm-lib-"))
    (libdir / "jolideco-gmm-library-index.json").write_text("{}")
    os.environ.setdefault("JOLIDECO_GMM_LIBRARY", str(libdir))
    os.environ.setdefault("MPLBACKEND", "agg")
    mod = types.ModuleType("jolideco.version")
    mod.version = "0.3.dev0-ref"
    sys.modules["jolideco.version"] = mod
    import jolideco

    jolideco.__jd_ref__ = True
    return jolideco
