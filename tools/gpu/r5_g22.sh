#!/bin/bash
# strip walks with a prefetch depth of 1 (211 registers: a walk wave fits beside a screen wave on a SIMD) against the default (231)
O=gpurun_out/r5z; mkdir -p $O
run() {  # label cfg env...
  label=$1; cfg=$2; shift; shift
  env "$@" python bench.py --config $cfg --steps 100 --warmup 10 --repeats 5 --no-cpu-baseline --no-general-psf --no-c6 > $O/$label.json 2> $O/$label.err
  python - $O/$label.json $label <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); print(f"{sys.argv[2]:20s} {d['value']:8.1f} it/s {d['ms_per_step']:.4f} ms  {d.get('kernel_ms_per_step')}  {d.get('graph_policy')}")
except Exception as e:
    print(sys.argv[2], 'failed', e)
PY
}
P1=JOLIDECO_HIP_LIBRARY=$PWD/jolideco_amd/libjolideco_hip_p1.so
for cfg in c3 c5 c4 c2; do
run ${cfg}_base $cfg A=1
run ${cfg}_p1 $cfg $P1
run ${cfg}_base2 $cfg A=1
run ${cfg}_p1b $cfg $P1
done
