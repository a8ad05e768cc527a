#!/bin/bash
mkdir -p gpurun_out/r5j
python -m pytest tests/test_gpu_distributed.py tests/test_gpu_fft_batch.py tests/test_gpu_fft_native.py -x -q -m gpu > gpurun_out/r5j/t1.log 2>&1; echo "rc=$?" >> gpurun_out/r5j/t1.log
python -m pytest tests/test_gpu_baseline_parity.py tests/test_gpu_kernels.py -x -q -m gpu -k "c6_shaped or calib or shift or band" > gpurun_out/r5j/t2.log 2>&1; echo "rc=$?" >> gpurun_out/r5j/t2.log
python tools/ab.py c6 2 20 -- rows: old:JD_FFT_BATCH=5 > gpurun_out/r5j/ab_c6_shift.txt 2>&1
python bench.py --shard-of 8 --rank 2 --steps 50 --warmup 10 --no-cpu-baseline --no-general-psf > gpurun_out/r5j/c3_rank2_of_8.json 2> gpurun_out/r5j/c3_rank2_of_8.err
python tools/shard_table.py c3 60 cost 8 > gpurun_out/r5j/shard_c3_cost8.txt 2>&1
python tools/shard_table.py c3 60 round-robin 8 > gpurun_out/r5j/shard_c3_rr8.txt 2>&1
tail -n 3 gpurun_out/r5j/t1.log gpurun_out/r5j/t2.log
grep " step " gpurun_out/r5j/ab_c6_shift.txt | cut -c1-420
grep "N=8" gpurun_out/r5j/shard_c3_cost8.txt gpurun_out/r5j/shard_c3_rr8.txt | cut -c1-200
python - <<'PY'
import json
d=json.load(open('gpurun_out/r5j/c3_rank2_of_8.json')); print('rank2of8', d['ms_per_step'], d['kernel_ms_per_step'])
PY
