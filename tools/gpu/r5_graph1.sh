#!/bin/bash
mkdir -p gpurun_out/r5e
python -m pytest tests/test_gpu_graph.py -x -q -m gpu > gpurun_out/r5e/t_graph.log 2>&1; echo "rc=$?" >> gpurun_out/r5e/t_graph.log
python -m pytest tests/test_gpu_fit.py tests/test_gpu_fft_batch.py tests/test_gpu_distributed.py -x -q -m gpu > gpurun_out/r5e/t2.log 2>&1; echo "rc=$?" >> gpurun_out/r5e/t2.log
python bench.py --config c6 > gpurun_out/r5e/c6.json 2> gpurun_out/r5e/c6.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-general-psf > gpurun_out/r5e/c3_quick.json 2> gpurun_out/r5e/c3_quick.err
JOLIDECO_GRAPH=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-general-psf > gpurun_out/r5e/c3_quick_nograph.json 2> gpurun_out/r5e/c3_quick_nograph.err
python bench.py --config c1 --steps 200 --warmup 20 > gpurun_out/r5e/c1.json 2> gpurun_out/r5e/c1.err
python bench.py --config e0102 --epochs 50 > gpurun_out/r5e/e0102_50.json 2> gpurun_out/r5e/e0102_50.err
tail -n 15 gpurun_out/r5e/t_graph.log | cut -c1-300; tail -n 3 gpurun_out/r5e/t2.log
for f in c6 c3_quick c3_quick_nograph c1 e0102_50; do echo "== $f"; cut -c1-900 gpurun_out/r5e/$f.json; tail -n 2 gpurun_out/r5e/$f.err | cut -c1-300; done
