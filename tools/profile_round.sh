#!/bin/bash
# Everything profiles/<round>/ is made of, in one call on a GPU box (run from the repository root):
#   tools/profile_round.sh r03      ->  gpurun_out/profile_r03/{c3,c2,c4,c5}_n1_bench.json, *_kernel_stats.csv, pmc_*/
# The PMC passes collect FETCH_SIZE and WRITE_SIZE in SEPARATE runs (MI355X_MICROARCH.md: they do not fit one pass).
set -euo pipefail  # a failed bench or profiler run stops the script: no partial profile round gets copied
R=${1:-r04}
OUT=gpurun_out/profile_$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
python3 bench.py > $OUT/c3_n1_bench.json 2> $OUT/c3_n1_bench.err
python3 bench.py --steps 20 --warmup 5 > $OUT/c3_n1_bench_driver_flags.json 2>> $OUT/c3_n1_bench.err
for cfg in c2 c4 c5; do
  python3 bench.py --config $cfg --steps 100 --warmup 10 > $OUT/${cfg}_n1_bench.json 2> $OUT/${cfg}_n1_bench.err
done
for cfg in c3 c4 c5; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$cfg -o $cfg -- \
      python3 bench.py --config $cfg --steps 20 --warmup 5 --no-cpu-baseline --no-general-psf > $OUT/prof_$cfg.log 2>&1
  stats=$(find $OUT/prof_$cfg -name "*kernel_stats.csv" | sort | sed -n 1p)
  [ -n "$stats" ] && [ -s "$stats" ] || { echo "no kernel_stats.csv for $cfg" >&2; exit 1; }
  cp "$stats" $OUT/${cfg}_n1_kernel_stats.csv
  if [ $cfg = c5 ]; then continue; fi  # (counters: the two configurations the roofline objects quote)
  for counter in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $counter --output-format csv -d $OUT/pmc_${cfg}_$counter -o pmc -- \
        python3 bench.py --config $cfg --steps 5 --warmup 2 --repeats 1 --settle-seconds 0 --no-cpu-baseline --no-general-psf \
        > $OUT/pmc_${cfg}_$counter.log 2>&1
  done
  python3 tools/pmc_traffic_csv.py ${cfg}h $OUT/pmc_${cfg}_FETCH_SIZE $OUT/pmc_${cfg}_WRITE_SIZE > $OUT/pmc_${cfg}.csv
done
find $OUT -name "*.csv" -size +3M -delete
find $OUT -name "*kernel_trace.csv" -delete
ls -la $OUT
