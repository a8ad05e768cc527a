#!/bin/bash
# tile heights of the walk launches / screen kernel form with the prior running beside the likelihood (c3)
O=gpurun_out/r5u; mkdir -p $O
run() {  # label env...
  label=$1; shift
  env "$@" python bench.py --config c3 --repeats 5 > $O/$label.json 2> $O/$label.err
  python - $O/$label.json $label <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); print(f"{sys.argv[2]:28s} {d['value']:8.1f} it/s {d['ms_per_step']:.4f} ms")
except Exception as e:
    print(sys.argv[2], 'failed', e)
PY
}
run base A=1
run rows56 JD_SEP_WALK_ROWS=56
run rows92 JD_SEP_WALK_ROWS=92
run rows128 JD_SEP_WALK_ROWS=128
run rows182 JD_SEP_WALK_ROWS=182
run adj48 JD_SEP_WALK_ADJ_ROWS=48
run adj72 JD_SEP_WALK_ADJ_ROWS=72
run adj108 JD_SEP_WALK_ADJ_ROWS=108
run adj33_72 JD_SEP_WALK_ADJ_ROWS33=72
run adj33_144 JD_SEP_WALK_ADJ_ROWS33=144
run np1 JD_GMM_SCREEN_NP=1
run base2 A=1
