#!/bin/bash
# kernel trace of the smallest calibrated fit (512^2 flux pixels, 8 observations): where its step goes
O=gpurun_out/r5n; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in 256:uniform:graph 256:gmm:graph; do
  n=$(echo $c | tr ':' '_')
  SMALL_FITS_ONLY=$c rocprofv3 --kernel-trace -d $R/$O/prof_$n -o small -- python3 $R/tools/gpu/small_fits.py > $R/$O/run_$n.log 2>&1
  grep flux $R/$O/run_$n.log
done
