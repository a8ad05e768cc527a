#!/bin/bash
mkdir -p gpurun_out/r5i
python -m pytest tests -x -q -m gpu > gpurun_out/r5i/t_all.log 2>&1; echo "rc=$?" >> gpurun_out/r5i/t_all.log
python tools/shard_table.py c3 60 cost 2 4 8 > gpurun_out/r5i/shard_c3_cost.txt 2>&1
python tools/shard_table.py c3 60 round-robin 4 8 > gpurun_out/r5i/shard_c3_rr.txt 2>&1
python tools/shard_table.py c5 40 cost 2 4 8 > gpurun_out/r5i/shard_c5_cost.txt 2>&1
python tools/shard_table.py c5 40 round-robin 4 8 > gpurun_out/r5i/shard_c5_rr.txt 2>&1
python bench.py --shard-of 8 --rank 2 --steps 50 --warmup 10 --no-cpu-baseline --no-general-psf > gpurun_out/r5i/c3_rank2_of_8.json 2> gpurun_out/r5i/c3_rank2_of_8.err
tail -n 4 gpurun_out/r5i/t_all.log
grep "max/min" gpurun_out/r5i/shard_*.txt
python - <<'PY'
import json
d=json.load(open('gpurun_out/r5i/c3_rank2_of_8.json')); print('rank2of8', d['ms_per_step'], d['kernel_ms_per_step'])
PY
