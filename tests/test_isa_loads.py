"""Build-time guard for a property the compiler can silently take away again (DESIGN_LOG.md, "What the compiler made of the
loads"; no GPU needed: hipcc -S).  Pointers read from a per-dataset table are GENERIC to the compiler; an access through
one is a flat_load / flat_store with a full `s_waitcnt vmcnt(0) lgkmcnt(0)` behind it, which serialises the loads of every
kernel that also works in LDS.  The native FFT kernels and the transposed shift go through the address-space-1 accessors
of csrc/jd_common.h (gld4 / gst4 / issue_row5 ...): their assembly must hold no flat accesses beyond the handful of
pointer fetches at a kernel's entry, and the column kernel's spectrum loads must not come back one per wait."""
import shutil
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tools"))

pytestmark = pytest.mark.skipif(shutil.which("hipcc") is None and not Path("/opt/rocm/bin/hipcc").exists(),
                                reason="needs hipcc (cross-compiles gfx950 without a GPU)")


def _demangled(rows):
    out = {}
    for name, loads, full, flat_ld, flat_st in rows:
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        out[dem.replace("jd::(anonymous namespace)::", "").replace("jd::", "")] = (loads, full, flat_ld, flat_st)
    return out


def test_native_fft_kernels_hold_no_flat_accesses_in_their_loops():
    import isa_loads

    kernels = _demangled(isa_loads.scan(ROOT / "jolideco_amd" / "csrc" / "fftnative.hip"))
    assert any("fftn_cols_kernel<128, 4, 16, 16, 9>" in k for k in kernels), sorted(kernels)[:5]
    for name, (loads, full, flat_ld, flat_st) in kernels.items():
        if not name.startswith("void fftn_"):
            continue
        # (entry: up to five per-dataset pointers fetched through a selected address; exits: the loss scalars of two blocks)
        assert flat_ld <= 5 and flat_st <= 2, (name, flat_ld, flat_st)
        static = "0, 0, 0, " not in name  # compile-time radix schedules: what c3fft / c6 run
        if static and "fftn_rows_inv_batch" not in name:
            assert flat_ld == 0, (name, flat_ld)
    loads, full, _, _ = next(v for k, v in kernels.items() if "fftn_cols_kernel<128, 4, 16, 16, 9>" in k)
    assert full <= loads // 2, ("the column kernel's loads wait one by one again", loads, full)


def test_transposed_shift_issues_its_row_loads_together():
    import isa_loads

    kernels = _demangled(isa_loads.scan(ROOT / "jolideco_amd" / "csrc" / "shift.hip"))
    loads, full, flat_ld, flat_st = next(v for k, v in kernels.items() if "shift_bwd4_kernel<4>" in k)
    assert flat_ld <= 1 and flat_st == 0, (flat_ld, flat_st)
    assert loads >= 24 and full <= loads // 2, (loads, full)
