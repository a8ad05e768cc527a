#!/bin/bash
mkdir -p gpurun_out/r5g
python -m pytest tests/test_gpu_fft_native.py tests/test_gpu_edge_cases.py tests/test_gpu_fft_batch.py -x -q -m gpu > gpurun_out/r5g/t1.log 2>&1; echo "rc=$?" >> gpurun_out/r5g/t1.log
python -m pytest tests/test_gpu_baseline_parity.py -x -q -m gpu -k "c6_shaped" > gpurun_out/r5g/t2.log 2>&1; echo "rc=$?" >> gpurun_out/r5g/t2.log
python -m pytest tests/test_gpu_fit.py tests/test_gpu_kernels.py tests/test_gpu_graph.py -x -q -m gpu > gpurun_out/r5g/t3.log 2>&1; echo "rc=$?" >> gpurun_out/r5g/t3.log
python tools/gpu/small_fits.py > gpurun_out/r5g/small_fits.txt 2>&1
python tools/rccl_probe.py > gpurun_out/r5g/rccl_probe.json 2> gpurun_out/r5g/rccl_probe.err
python bench.py --config e0102 > gpurun_out/r5g/e0102.json 2> gpurun_out/r5g/e0102.err
tail -n 3 gpurun_out/r5g/t1.log gpurun_out/r5g/t2.log gpurun_out/r5g/t3.log
grep "flux grid" gpurun_out/r5g/small_fits.txt; tail -n 2 gpurun_out/r5g/small_fits.txt | cut -c1-300
cat gpurun_out/r5g/rccl_probe.json | cut -c1-900; tail -n 2 gpurun_out/r5g/rccl_probe.err | cut -c1-300
cut -c1-1500 gpurun_out/r5g/e0102.json
