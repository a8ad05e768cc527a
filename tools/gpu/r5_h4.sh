#!/bin/bash
O=gpurun_out/r5h4; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_edge_cases.py tests/test_gpu_fit.py tests/test_gpu_graph.py tests/test_gpu_random_parity.py -x -q -m gpu > $O/t.log 2>&1; echo "rc=$?" >> $O/t.log; tail -n 3 $O/t.log
timeout 600 python tools/ab.py c3 5 30 -- base: nopre:JD_GMM_GATHER_PRELOAD=0 > $O/c3.txt 2>&1; grep -v amdgpu.ids $O/c3.txt | cut -c1-420
timeout 600 python tools/ab.py c4 4 20 -- base: nopre:JD_GMM_GATHER_PRELOAD=0 > $O/c4.txt 2>&1; grep -v amdgpu.ids $O/c4.txt | cut -c1-420
timeout 600 python tools/ab.py c2 5 60 -- base: nopre:JD_GMM_GATHER_PRELOAD=0 > $O/c2.txt 2>&1; grep -v amdgpu.ids $O/c2.txt | cut -c1-420
cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o c3 -- python3 $GRAFT_REPO_ROOT/bench.py --config c3 --no-c6 --no-cpu-baseline --no-general-psf > $GRAFT_REPO_ROOT/$O/bench_c3.json 2> $GRAFT_REPO_ROOT/$O/bench_c3.err
cd $GRAFT_REPO_ROOT; python - <<'PY'
import csv, glob
for f in glob.glob('gpurun_out/r5h4/prof/**/*kernel_stats.csv', recursive=True):
    for r in list(csv.DictReader(open(f)))[:16]:
        print(r['Name'][:90], r['Calls'], r['AverageNs'])
PY
