#!/bin/bash
# one-wave generic row kernels (rows <= 1024 points): parity tests, then small fits with and without
O=gpurun_out/r5p; mkdir -p $O
python -m pytest tests/test_gpu_fft_native.py tests/test_gpu_fft_batch.py tests/test_gpu_graph.py -x -q -m gpu > $O/t1.log 2>&1; echo "rc=$?" >> $O/t1.log
tail -n 4 $O/t1.log
python -m pytest tests/test_gpu_baseline_parity.py tests/test_gpu_fit.py tests/test_gpu_edge_cases.py -x -q -m gpu -k "calib or c6 or fft or upsampl or pooled" > $O/t2.log 2>&1; echo "rc=$?" >> $O/t2.log
tail -n 4 $O/t2.log
for tiny in 1024 0 1024 0; do
  echo "JD_FFT_TINY=$tiny"
  JD_FFT_TINY=$tiny SMALL_FITS_ONLY=256:uniform:by-value python tools/gpu/small_fits.py 2>&1 | grep flux
  JD_FFT_TINY=$tiny SMALL_FITS_ONLY=256:uniform:graph python tools/gpu/small_fits.py 2>&1 | grep flux
  JD_FFT_TINY=$tiny SMALL_FITS_ONLY=256:gmm:graph python tools/gpu/small_fits.py 2>&1 | grep flux
done
for tiny in 1024 0; do JD_FFT_TINY=$tiny python bench.py --config e0102 > $O/e0102_$tiny.json 2> $O/e0102_$tiny.err; python - $O/e0102_$tiny.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); print('e0102', d['value'], d['unit'], d.get('ms_per_step'))
PY
done
