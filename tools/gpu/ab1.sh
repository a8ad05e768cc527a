cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
mkdir -p gpurun_out/ab1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ab1/prof -o c3 -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-general-psf > gpurun_out/ab1/prof.log 2>&1
cp $(find gpurun_out/ab1/prof -name "*kernel_stats.csv" | head -1) gpurun_out/ab1/c3_kernel_stats.csv
find gpurun_out/ab1/prof -name "*.csv" -size +1M -delete
python3 tools/ab.py c3 5 30 -- base: adjrows33_72:JD_SEP_WALK_ADJ_ROWS33=72 adjrows33_108:JD_SEP_WALK_ADJ_ROWS33=108 adjrows33_144:JD_SEP_WALK_ADJ_ROWS33=144 plain33:JD_SEP_WALK_ADJ33=1 plain33_76:JD_SEP_WALK_ADJ33=1,JD_SEP_WALK_ADJ_ROWS33=76 plain33_112:JD_SEP_WALK_ADJ33=1,JD_SEP_WALK_ADJ_ROWS33=112 cost150:JD_SEP_WALK_COST33=150 cost230:JD_SEP_WALK_COST33=230 cost280:JD_SEP_WALK_COST33=280 > gpurun_out/ab1/ab.txt 2>&1
python -m pytest tests/test_gpu_distributed.py -x -q -k rccl > gpurun_out/ab1/rccl.txt 2>&1
tail -5 gpurun_out/ab1/rccl.txt
cat gpurun_out/ab1/ab.txt
