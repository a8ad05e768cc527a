#!/usr/bin/env python3
"""Instruction mix of the kernels of one .hip file whose mangled name contains a pattern:
    python tools/isa_mix.py jolideco_amd/csrc/walkconv.hip walk_kernelILi17ELi4ELi2ELb1ELb1ELi0
Counts are over the whole kernel body (an unrolled walk loop of WS rows dominates it)."""
import re
import subprocess
import sys
from collections import Counter
from pathlib import Path

FLAGS = "-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form".split()
src = Path(sys.argv[1]).resolve()
pattern = sys.argv[2]
extra = sys.argv[3:]
out = Path("/tmp") / (src.stem + ".s")
subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, *extra, "-S", "--cuda-device-only", str(src), "-o", str(out)], cwd=src.parent,
               capture_output=True, text=True, check=True)
lines = out.read_text().split("\n")
start = None
for n, line in enumerate(lines):
    if start is None and re.match(r"^_Z\S*:", line) and pattern in line:
        start, name = n, line.split(":")[0]
    elif start is not None and "s_endpgm" in line:
        c = Counter()
        for l in lines[start:n]:
            m = re.match(r"\s+([a-z]\w+)", l)
            if m:
                c[m.group(1)] += 1
        valu = sum(v for k, v in c.items() if k.startswith("v_"))
        salu = sum(v for k, v in c.items() if k.startswith("s_") and not k.startswith("s_waitcnt"))
        print(name[:100])
        print(f"  VALU {valu}  SALU {salu}  s_waitcnt {c['s_waitcnt']}  lines {n - start}")
        for k in ("v_pk_fma_f32", "v_pk_mul_f32", "v_fma_f32", "v_fmac_f32", "v_mul_f32", "v_mov_b32", "v_accvgpr_write_b32", "v_accvgpr_read_b32"):
            print(f"    {k:22s}{c[k]}")
        print("    ds  :", {k: v for k, v in c.items() if k.startswith("ds_")})
        print("    vmem:", {k: v for k, v in c.items() if k.startswith("global_") or k.startswith("buffer_") or k.startswith("flat_")})
        sample = [l.strip() for l in lines[start:n] if "v_pk_fma_f32" in l and " s[" in l][:2]
        print("    e.g.", sample)
        start = None
