#!/bin/bash
# Everything profiles/<round>/ is made of, in one call on a GPU box (run from the repository root):
#   tools/profile_round.sh r04   ->  gpurun_out/profile_r04/{c3,c2,c4,c5,c6}_n1_bench.json, *_kernel_stats.csv, pmc_*, sq_*
# The PMC passes collect FETCH_SIZE and WRITE_SIZE in SEPARATE runs (MI355X_MICROARCH.md: they do not fit one pass), the
# SQ counters of the GMM screen kernel in two more; every rocprofv3 call has python3 right behind "--".
set -euo pipefail  # a failed bench or profiler run stops the script: no partial profile round gets copied
R=${1:-r05}
OUT=gpurun_out/profile_${R}_lite
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
python3 bench.py > $OUT/c3_n1_bench.json 2> $OUT/c3_n1_bench.err
python3 bench.py --steps 20 --warmup 5 > $OUT/c3_n1_bench_driver_flags.json 2>> $OUT/c3_n1_bench.err
for cfg in c2 c4 c5; do
  python3 bench.py --config $cfg --steps 100 --warmup 10 > $OUT/${cfg}_n1_bench.json 2> $OUT/${cfg}_n1_bench.err
done
python3 bench.py --config c6 > $OUT/c6_n1_bench.json 2> $OUT/c6_n1_bench.err
python3 bench.py --config c1 --steps 200 --warmup 20 > $OUT/c1_n1_bench.json 2> $OUT/c1_n1_bench.err
python3 bench.py --config e0102 > $OUT/e0102_bench.json 2> $OUT/e0102_bench.err
# (the profiler runs below time the by-value epochs: the same kernels, no graph replay under the tracer)
export JOLIDECO_GRAPH=0 JOLIDECO_STEP_SCALARS=host
kernel_stats() {  # <label> <bench args...>: rocprofv3 kernel statistics of one bench command
  local label=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$label -o $label -- \
      python3 bench.py "$@" --steps 20 --warmup 5 --no-cpu-baseline --no-general-psf > $OUT/prof_$label.log 2>&1
  local stats
  stats=$(find $OUT/prof_$label -name "*kernel_stats.csv" | sort | sed -n 1p)
  [ -n "$stats" ] && [ -s "$stats" ] || { echo "no kernel_stats.csv for $label" >&2; exit 1; }
  cp "$stats" $OUT/${label}_n1_kernel_stats.csv
}
traffic() {  # <label> <bench args...>: FETCH_SIZE and WRITE_SIZE per kernel, separate passes
  local label=$1; shift
  for counter in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $counter --output-format csv -d $OUT/pmc_${label}_$counter -o pmc -- \
        python3 bench.py "$@" --steps 5 --warmup 2 --repeats 1 --settle-seconds 0 --no-cpu-baseline --no-general-psf \
        > $OUT/pmc_${label}_$counter.log 2>&1
  done
  python3 tools/pmc_traffic_csv.py ${label} $OUT/pmc_${label}_FETCH_SIZE $OUT/pmc_${label}_WRITE_SIZE > $OUT/pmc_${label}.csv
}
for cfg in c3 c4 c5 c6; do kernel_stats $cfg --config $cfg; done
traffic c3 --config c3
traffic c4 --config c4
traffic c6 --config c6
# the same fit through the FFT path (the native FFT convolution on these sizes)
export JOLIDECO_CONV_METHOD=fft
kernel_stats c3fft --config c3
traffic c3fft --config c3
unset JOLIDECO_CONV_METHOD
# SQ counters of the default c3 step (the GMM screen kernel is the dominant launch): five passes of four counters
sq_pass() {
  local name=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d $OUT/sq_$name -o pmc -- \
      python3 bench.py --steps 5 --warmup 2 --repeats 1 --settle-seconds 0 --no-cpu-baseline --no-general-psf > $OUT/sq_$name.log 2>&1
}
# MFMA Toeplitz convolution against the native FFT convolution by PSF size (the rule of the method "auto")
find $OUT -name "*.csv" -size +3M -delete
find $OUT -name "*kernel_trace.csv" -delete
ls -la $OUT
