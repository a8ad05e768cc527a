mkdir -p gpurun_out/c6prof
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/c6prof/prof -o c6 -- python3 bench.py --config c6 --steps 50 --warmup 10 > gpurun_out/c6prof/log.txt 2>&1
cp $(find gpurun_out/c6prof/prof -name "*kernel_stats.csv" | sort | sed -n 1p) gpurun_out/c6prof/c6_n1_kernel_stats.csv
find gpurun_out/c6prof/prof -name "*.csv" -size +1M -delete
head -25 gpurun_out/c6prof/c6_n1_kernel_stats.csv | cut -c1-200
