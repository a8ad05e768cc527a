"""CPU: the FFT core of the native FFT convolution (csrc/jd_fftcore.h: Stockham passes, in-register radix 2 / 3 / 4 / 8 /
9 / 16 butterflies, radix schedules of the lengths 2^a * {1, 3, 9}) compiled for the host and run butterfly by
butterfly against a float64 DFT (tests/native/fftcore_check.cpp)."""
import shutil
import subprocess
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parent.parent


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_fft_core_matches_a_float64_dft(tmp_path):
    exe = tmp_path / "fftcore_check"
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", str(REPO / "jolideco_amd" / "csrc"), str(REPO / "tests" / "native" / "fftcore_check.cpp"),
                    "-o", str(exe)], check=True)
    done = subprocess.run([str(exe)], capture_output=True, text=True)
    assert done.returncode == 0, done.stdout[-3000:]
    assert "BAD" not in done.stdout and done.stdout.count("rel err") == 34
    assert "2064 -> 2304, 1056 -> 1152" in done.stdout
