#!/bin/bash
O=gpurun_out/r5v; mkdir -p $O
run() {  # label cfg env...
  label=$1; cfg=$2; shift; shift
  env "$@" python bench.py --config $cfg --steps 100 --warmup 10 --repeats 5 --no-cpu-baseline > $O/$label.json 2> $O/$label.err
  python - $O/$label.json $label <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); print(f"{sys.argv[2]:28s} {d['value']:8.1f} it/s {d['ms_per_step']:.4f} ms  host {d.get('host_enqueue_ms_per_step')}  {d.get('graph_policy')}")
except Exception as e:
    print(sys.argv[2], 'failed', e)
PY
}
for cfg in c2 c5; do
run ${cfg}_ov1_auto $cfg A=1
run ${cfg}_ov0_auto $cfg JOLIDECO_PRIOR_OVERLAP=0
run ${cfg}_ov1_g0 $cfg JOLIDECO_GRAPH=0
run ${cfg}_ov0_g0 $cfg JOLIDECO_PRIOR_OVERLAP=0 JOLIDECO_GRAPH=0
run ${cfg}_ov1_g1 $cfg JOLIDECO_GRAPH=1
run ${cfg}_ov0_g1 $cfg JOLIDECO_PRIOR_OVERLAP=0 JOLIDECO_GRAPH=1
run ${cfg}_legacy $cfg JOLIDECO_STEP_SCALARS=host
done
