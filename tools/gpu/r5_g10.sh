#!/bin/bash
# kernel trace of the smallest calibrated fit (512^2 flux pixels, 8 observations, uniform prior): where its 253 us per step go
O=gpurun_out/r5n; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in 256:uniform:by-value 256:gmm:planned; do
  n=$(echo $c | tr ':' '_')
  SMALL_FITS_ONLY=$c JOLIDECO_GRAPH=0 rocprofv3 --kernel-trace --stats -d $R/$O/prof_$n -o small -- python3 $R/tools/gpu/small_fits.py > $R/$O/run_$n.log 2>&1
  f=$(ls $R/$O/prof_$n/*/*kernel_stats.csv $R/$O/prof_$n/*kernel_stats.csv 2>/dev/null | head -n 1)
  echo "== $c"; python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:24]:
    print(f"{r['Name'][:110]:110s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:8.1f} us  {r['Percentage']}%")
PY
  rm -rf $R/$O/prof_$n/*/*.db
done
