#!/bin/bash
mkdir -p gpurun_out/r5f
python -m pytest tests/test_gpu_graph.py tests/test_gpu_fit.py tests/test_gpu_fft_batch.py tests/test_gpu_distributed.py -x -q -m gpu > gpurun_out/r5f/t1.log 2>&1; echo "rc=$?" >> gpurun_out/r5f/t1.log
python tools/shard_table.py c3 60 cost 2 4 8 > gpurun_out/r5f/shard_c3_cost.txt 2>&1
python tools/shard_table.py c3 60 round-robin 4 8 > gpurun_out/r5f/shard_c3_rr.txt 2>&1
python tools/shard_table.py c5 40 cost 2 4 8 > gpurun_out/r5f/shard_c5_cost.txt 2>&1
python tools/shard_table.py c5 40 round-robin 8 > gpurun_out/r5f/shard_c5_rr.txt 2>&1
python bench.py > gpurun_out/r5f/c3_bench.json 2> gpurun_out/r5f/c3_bench.err
tail -n 4 gpurun_out/r5f/t1.log
grep "N=" gpurun_out/r5f/shard_c3_cost.txt gpurun_out/r5f/shard_c3_rr.txt gpurun_out/r5f/shard_c5_cost.txt gpurun_out/r5f/shard_c5_rr.txt | grep "max/min"
cut -c1-1200 gpurun_out/r5f/c3_bench.json; tail -n 3 gpurun_out/r5f/c3_bench.err | cut -c1-200
