#!/usr/bin/env python3
"""Two checks of every kernel's device assembly that found the round-5 load problems (DESIGN_LOG.md, "What the compiler made
of the loads"):
  * `vmcnt(0)` waits against loads: about as many full waits as loads means load, wait, use -- one at a time;
  * flat_load / flat_store: accesses through a pointer the compiler takes for generic (read from a per-dataset table):
    each needs `s_waitcnt vmcnt(0) lgkmcnt(0)`.
    python tools/isa_loads.py [file.hip ...]        (default: every .hip of jolideco_amd/csrc; hipcc -S, no GPU needed)
"""
import re
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
FLAGS = "-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form --cuda-device-only -S".split()


def scan(src):
    out = Path("/tmp") / (src.stem + ".isa_loads.s")
    subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, str(src), "-o", str(out)], cwd=src.parent, capture_output=True, check=True)
    name, rows = None, []
    for line in out.read_text().split("\n"):
        m = re.match(r"^(_Z\S+):", line)
        if m:
            name, loads, full, flat_ld, flat_st = m.group(1), 0, 0, 0, 0
            continue
        if name is None:
            continue
        loads += "global_load" in line or "flat_load" in line or "buffer_load" in line
        flat_ld += "flat_load" in line
        flat_st += "flat_store" in line
        full += bool(re.search(r"s_waitcnt vmcnt\(0\)", line))
        if "s_endpgm" in line:
            rows.append((name, loads, full, flat_ld, flat_st))
            name = None
    return rows


def main():
    files = [Path(f).resolve() for f in sys.argv[1:]] or sorted((ROOT / "jolideco_amd" / "csrc").glob("*.hip"))
    for src in files:
        for name, loads, full, flat_ld, flat_st in scan(src):
            if loads >= 6 and (full >= 0.5 * loads or flat_ld + flat_st > 2):
                dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
                dem = dem.replace("jd::(anonymous namespace)::", "").replace("jd::", "")
                print(f"{src.name:16s} loads {loads:4d}  full waits {full:4d}  flat ld/st {flat_ld:3d}/{flat_st:3d}  {dem[:96]}")


if __name__ == "__main__":
    main()
