mkdir -p gpurun_out/c5a
timeout 1200 python -m pytest tests/test_gpu_mixed_psf.py tests/test_gpu_fit.py -x -q -k "two_components or mixed or four_components or batched" > gpurun_out/c5a/tests.txt 2>&1
tail -4 gpurun_out/c5a/tests.txt
python3 tools/ab.py c5 3 20 -- base: > gpurun_out/c5a/ab.txt 2>&1
grep -v amdgpu.ids gpurun_out/c5a/ab.txt | cut -c1-300
