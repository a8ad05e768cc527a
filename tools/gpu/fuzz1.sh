mkdir -p gpurun_out/fuzz1
timeout 900 python tools/fuzz_batch.py 90 4 walk > gpurun_out/fuzz1/walk.txt 2>&1; tail -3 gpurun_out/fuzz1/walk.txt
timeout 900 python tools/fuzz_batch.py 90 5 > gpurun_out/fuzz1/tile.txt 2>&1; tail -3 gpurun_out/fuzz1/tile.txt
timeout 600 python tools/fuzz_conv.py 40 3 > gpurun_out/fuzz1/conv.txt 2>&1; tail -3 gpurun_out/fuzz1/conv.txt
