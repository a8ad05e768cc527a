#!/bin/bash
mkdir -p gpurun_out/r5h
python -m pytest tests/test_gpu_graph.py tests/test_cabi.py -x -q -m gpu -s > gpurun_out/r5h/t1.log 2>&1; echo "rc=$?" >> gpurun_out/r5h/t1.log
python tools/ab.py c6 2 20 -- cb4: cb8:JD_FFT_NATIVE=8 cb2:JD_FFT_NATIVE=2 > gpurun_out/r5h/ab_c6_cb.txt 2>&1
python tools/ab.py c3fft 2 40 -- cb4: cb8:JD_FFT_NATIVE=8 > gpurun_out/r5h/ab_c3fft_cb.txt 2>&1
python tools/gpu/small_fits.py > gpurun_out/r5h/small_fits.txt 2>&1
python bench.py --shard-of 8 --rank 2 --steps 50 --warmup 10 --no-cpu-baseline --no-general-psf > gpurun_out/r5h/c3_rank2_of_8.json 2> gpurun_out/r5h/c3_rank2_of_8.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r5h/c3_quick.json 2> gpurun_out/r5h/c3_quick.err
tail -n 6 gpurun_out/r5h/t1.log | cut -c1-300
grep " step " gpurun_out/r5h/ab_c6_cb.txt gpurun_out/r5h/ab_c3fft_cb.txt | cut -c1-420
grep "flux grid" gpurun_out/r5h/small_fits.txt
python - <<'PY'
import json
d=json.load(open('gpurun_out/r5h/c3_rank2_of_8.json')); print('rank2of8', d['ms_per_step'], d['kernel_ms_per_step'], d.get('graph_policy'))
d=json.load(open('gpurun_out/r5h/c3_quick.json')); print('c3', d['value'], d['ms_per_step'], d['host_enqueue_ms_per_step'], d.get('graph_policy'), d.get('graph_replay'), d.get('odd_size_fft'), d['c6_chandra_like']['value'])
PY
