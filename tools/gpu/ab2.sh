# packed row pass: correctness, then A/B against the scalar-row build in the same call (same box)
mkdir -p gpurun_out/ab2
python -m pytest tests/test_gpu_mixed_psf.py tests/test_gpu_distributed.py -x -q -k "not bench_two and not two_rank" > gpurun_out/ab2/tests1.txt 2>&1
tail -4 gpurun_out/ab2/tests1.txt
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fit.py tests/test_gpu_edge_cases.py -x -q -k "walk or separable or batched or strip or operator or likelihood or aliasing or poisson_epilogue or gather or optimizer_step" > gpurun_out/ab2/tests2.txt 2>&1
tail -4 gpurun_out/ab2/tests2.txt
for cfg in c3 c4 c5; do
  for lib in packed scalar; do
    if [ $lib = scalar ]; then export JOLIDECO_HIP_LIBRARY=$PWD/jolideco_amd/libjolideco_hip_scalar_row.so; else unset JOLIDECO_HIP_LIBRARY; fi
    echo "== $cfg $lib" >> gpurun_out/ab2/ab.txt
    python3 tools/ab.py $cfg 5 30 -- base: >> gpurun_out/ab2/ab.txt 2>&1
  done
done
unset JOLIDECO_HIP_LIBRARY
python3 tools/ab.py c3 5 30 -- base: cost150:JD_SEP_WALK_COST33=150 cost230:JD_SEP_WALK_COST33=230 cost280:JD_SEP_WALK_COST33=280 adjrows66:JD_SEP_WALK_ADJ_ROWS=72 >> gpurun_out/ab2/ab.txt 2>&1
grep -v "amdgpu.ids" gpurun_out/ab2/ab.txt
