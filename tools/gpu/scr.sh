mkdir -p gpurun_out/scr; rm -f gpurun_out/scr/ab.txt
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_baseline_parity.py -x -q -m gpu -k "gmm or prior or c2 or c3" > gpurun_out/scr/tests.log 2>&1; tail -3 gpurun_out/scr/tests.log
for r in 1 2 3; do
for lib in old default; do
  if [ $lib = default ]; then unset JOLIDECO_HIP_LIBRARY; else export JOLIDECO_HIP_LIBRARY=jolideco_amd/libjolideco_hip_$lib.so; fi
  timeout 300 python tools/ab.py c3 1 40 -- $lib: 2>&1 | grep step >> gpurun_out/scr/ab.txt
done; done
cut -c1-200 gpurun_out/scr/ab.txt
