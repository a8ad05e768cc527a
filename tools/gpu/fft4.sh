mkdir -p gpurun_out/fft4
timeout 900 python -m pytest tests/test_gpu_fft_native.py tests/test_gpu_kernels.py tests/test_gpu_edge_cases.py tests/test_gpu_fit.py -x -q -k "fft or native" > gpurun_out/fft4/tests.txt 2>&1
tail -4 gpurun_out/fft4/tests.txt
JOLIDECO_CONV_METHOD=fft timeout 600 python3 tools/ab.py c3 3 20 -- fused: unfused:JD_SEP_NO_FUSION=1 > gpurun_out/fft4/ab.txt 2>&1
grep -v amdgpu.ids gpurun_out/fft4/ab.txt | cut -c1-330
