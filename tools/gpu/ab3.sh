mkdir -p gpurun_out/ab3
python -m pytest tests/test_gpu_distributed.py -x -q -k "rccl" > gpurun_out/ab3/rccl.txt 2>&1
tail -3 gpurun_out/ab3/rccl.txt
for lib in scalar pkrow scalar pkrow; do
  if [ $lib = pkrow ]; then export JOLIDECO_HIP_LIBRARY=$PWD/jolideco_amd/libjolideco_hip_pkrow.so; else unset JOLIDECO_HIP_LIBRARY; fi
  echo "== c3 $lib" >> gpurun_out/ab3/ab.txt
  python3 tools/ab.py c3 4 30 -- c190: c230:JD_SEP_WALK_COST33=230 c260:JD_SEP_WALK_COST33=260 >> gpurun_out/ab3/ab.txt 2>&1
done
export JOLIDECO_HIP_LIBRARY=$PWD/jolideco_amd/libjolideco_hip_pkrow.so
python -m pytest tests/test_gpu_mixed_psf.py -x -q > gpurun_out/ab3/tests_pk.txt 2>&1
tail -2 gpurun_out/ab3/tests_pk.txt
grep -v "amdgpu.ids" gpurun_out/ab3/ab.txt
