"""CPU: libjolideco_hip.so loads, exports every symbol include/jolideco_hip.h declares, and its
host-side argument validation works (only calls that return before touching a device)."""
import ctypes
import re
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parent.parent
HEADER = REPO / "include" / "jolideco_hip.h"


@pytest.fixture(scope="module")
def lib():
    from jolideco_amd import _hip

    if not _hip.library_path().exists():
        import __graft_entry__

        __graft_entry__.build()
    return _hip.lib()


def declared_symbols():
    text = HEADER.read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(jd_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_are_exported_and_bound(lib):
    from jolideco_amd import _hip

    names = declared_symbols()
    assert len(names) >= 20
    for name in names:
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
    # the ctypes table binds exactly the declared functions (no stale or missing prototypes)
    assert sorted(_hip.EXPORTS) == names


def test_no_cxx_or_torch_types_in_the_abi():
    """The boundary is plain C: no C++ / torch types in the header."""
    text = re.sub(r"/\*.*?\*/", "", HEADER.read_text(), flags=re.S)  # comments cite torch files
    for forbidden in ("std::", "at::", "torch", "Tensor", "template", "class "):
        assert forbidden not in text, forbidden
    assert 'extern "C"' in text
    assert "#include <stddef.h>" in text and "#include <stdint.h>" in text


def test_library_identity(lib):
    assert lib.jd_version() >= 100
    assert lib.jd_target_arch() == b"gfx950"
    assert lib.jd_kernel_name(0) == b"poisson_fused_kernel"
    assert lib.jd_kernel_name(1) == b"gmm_fwd_kernel"


def test_argument_validation_reports_errors(lib):
    """Bad arguments return JD_ERR_INVALID (-1) with a message and never reach the device."""
    handle = ctypes.c_void_p()
    assert lib.jd_conv_plan_create(0, 16, 3, 3, 0, ctypes.byref(handle)) == -1
    assert b"non-positive shape" in lib.jd_last_error()
    assert lib.jd_conv_plan_create(16, 16, 3, 3, 0, None) == -1
    assert lib.jd_gmm_create(4, 64, None, None, None, None, ctypes.byref(handle)) == -1
    assert b"null argument" in lib.jd_last_error()
    arr = (ctypes.c_float * 4)()
    fp = ctypes.cast(arr, ctypes.POINTER(ctypes.c_float))
    assert lib.jd_gmm_create(4, 16, fp, fp, fp, fp, ctypes.byref(handle)) == -1
    assert b"D = 64" in lib.jd_last_error()
    assert lib.jd_poisson_nll(None, None, 0, 0.0, 1e-25, None, None, None) == -1
    assert lib.jd_adam_step(None, None, None, None, None, None, None, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, None, None) == -1
    assert lib.jd_elementwise_prior_fwd_bwd(7, None, 0, 0, 0, 0, None, 0, None, None) == -1
    assert lib.jd_profile_enable(0) == -1
    total, count = ctypes.c_double(), ctypes.c_longlong()
    assert lib.jd_profile_read(99, ctypes.byref(total), ctypes.byref(count)) == -1
    # destroying a null handle is a no-op
    assert lib.jd_conv_plan_destroy(None) == 0 and lib.jd_gmm_destroy(None) == 0


def test_check_raises_runtime_error(lib):
    from jolideco_amd import _hip

    with pytest.raises(RuntimeError, match="libjolideco_hip error -1"):
        _hip.check(lib.jd_conv_plan_create(-1, -1, 0, 0, 0, None))


def test_product_path_has_no_cpu_fallback():
    """Without a HIP device the product raises; it never computes on the CPU."""
    import numpy as np
    import torch

    from jolideco_amd import MAPDeconvolver, SpatialFluxComponent
    from jolideco_amd.ops import ConvPlan, require_hip_tensor

    with pytest.raises(RuntimeError, match="no CPU fallback"):
        MAPDeconvolver(n_epochs=1, device="cpu")
    with pytest.raises(RuntimeError):
        require_hip_tensor(torch.zeros(4, 4))
    with pytest.raises(RuntimeError, match="HIP device"):
        ConvPlan(8, 8, 3, 3, "cpu")
    if not torch.cuda.is_available():
        comp = SpatialFluxComponent.from_numpy(flux=np.ones((16, 16)))
        data = {"counts": np.ones((16, 16), np.float32), "psf": np.ones((3, 3), np.float32) / 9,
                "exposure": np.ones((16, 16), np.float32), "background": np.ones((16, 16), np.float32)}
        with pytest.raises((RuntimeError, AssertionError)):
            MAPDeconvolver(n_epochs=1, device="cuda", display_progress=False).run({"d": data}, components=comp)


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under jolideco_amd/ may reference it."""
    for path in (REPO / "jolideco_amd").rglob("*.py"):
        text = path.read_text()
        assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), path
    for path in (REPO / "jolideco_amd" / "csrc").glob("*"):
        if path.suffix in (".hip", ".h"):
            assert "oracle" not in path.read_text(), path
