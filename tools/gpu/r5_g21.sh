#!/bin/bash
O=gpurun_out/r5y; mkdir -p $O
python bench.py > $O/c3_n1_bench.json 2> $O/c3_n1_bench.err; echo "rc=$?"
python bench.py --steps 20 --warmup 5 > $O/c3_n1_bench_driver_flags.json 2>> $O/c3_n1_bench.err; echo "rc=$?"
python bench.py --config c6 > $O/c6_n1_bench.json 2> $O/c6_n1_bench.err; echo "rc=$?"
python bench.py --config e0102 > $O/e0102_bench.json 2> $O/e0102_bench.err; echo "rc=$?"
python - <<'PY'
import json
for c in ("c3_n1_bench","c3_n1_bench_driver_flags","c6_n1_bench","e0102_bench"):
    d=json.load(open(f"gpurun_out/r5y/{c}.json")); print(c, round(d['value'],2), round(d['ms_per_step'],4), d.get('graph_policy'))
    for k in ("fft_psf","general_psf","dense_fp32_gmm","graph_replay","c6_chandra_like","sequential_mode"):
        if k in d: print("   ", k, {kk: (round(v,3) if isinstance(v,float) else v) for kk,v in d[k].items() if kk in ("value","ms_per_step","epochs_per_s")})
    if "roofline_c6" in d: print("   roofline_c6", round(d["roofline_c6"]["frac"],3), {k: round(v["frac"],3) for k,v in d["roofline_c6"]["launches"].items()})
PY
