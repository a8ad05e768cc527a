#!/usr/bin/env python3
"""Register / LDS / scratch use of every kernel of one .hip file (hipcc -Rpass-analysis=kernel-resource-usage).

    python tools/kernel_resources.py jolideco_amd/csrc/walkconv.hip [extra hipcc flags]
"""
import re
import subprocess
import sys
from pathlib import Path

FLAGS = "-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form".split()


def main():
    src = Path(sys.argv[1]).resolve()
    cmd = ["/opt/rocm/bin/hipcc", *FLAGS, *sys.argv[2:], "-Rpass-analysis=kernel-resource-usage", "-c", str(src), "-o", "/dev/null"]
    txt = subprocess.run(cmd, capture_output=True, text=True, cwd=src.parent).stderr
    keys = {"VGPR": "VGPRs", "AGPR": "AGPRs", "SGPR": "SGPRs", "scratch": r"ScratchSize \[bytes/lane\]",
            "occ": r"Occupancy \[waves/SIMD\]", "LDS": r"LDS Size \[bytes/block\]"}
    for block in re.split(r"remark: [^\n]*Function Name: ", txt)[1:]:
        name = block.split("\n")[0].split(" [-R")[0].strip()
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        dem = dem.replace("jd::(anonymous namespace)::", "").replace("void ", "")
        vals = []
        for label, key in keys.items():
            m = re.search(r" " + key + r": (\d+)", block)
            vals.append(f"{label} {m.group(1) if m else '?':>5s}")
        print(f"{dem[:84]:86s}" + "  ".join(vals))


if __name__ == "__main__":
    main()
