#!/bin/bash
O=gpurun_out/r5h3; mkdir -p $O
timeout 600 python tools/ab.py c3 5 30 -- base: nopre:JD_GMM_GATHER_PRELOAD=0 sb1024:JD_GMM_SORT_BLOCKS=1024 sb512:JD_GMM_SORT_BLOCKS=512 sb256:JD_GMM_SORT_BLOCKS=256 > $O/c3.txt 2>&1; grep -v amdgpu.ids $O/c3.txt | cut -c1-420
timeout 600 python tools/ab.py c4 4 20 -- base: nopre:JD_GMM_GATHER_PRELOAD=0 sb1024:JD_GMM_SORT_BLOCKS=1024 sb512:JD_GMM_SORT_BLOCKS=512 > $O/c4.txt 2>&1; grep -v amdgpu.ids $O/c4.txt | cut -c1-420
timeout 600 python tools/ab.py c2 5 60 -- base: nopre:JD_GMM_GATHER_PRELOAD=0 sb256:JD_GMM_SORT_BLOCKS=256 sb128:JD_GMM_SORT_BLOCKS=128 > $O/c2.txt 2>&1; grep -v amdgpu.ids $O/c2.txt | cut -c1-420
timeout 900 python -m pytest tests/test_gpu_fit.py tests/test_gpu_graph.py -x -q -m gpu > $O/t.log 2>&1; echo "rc=$?" >> $O/t.log; tail -n 3 $O/t.log
