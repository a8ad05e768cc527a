#!/bin/bash
# the rank-share tables of DESIGN section 4 on the final build + the bench lines under the final policy
O=gpurun_out/r5x; mkdir -p $O
python tools/shard_table.py c3 60 cost 2 4 8 > $O/shard_c3_cost.txt 2> $O/shard_c3_cost.err
python tools/shard_table.py c3 60 round-robin 4 8 > $O/shard_c3_rr.txt 2> $O/shard_c3_rr.err
python tools/shard_table.py c5 60 cost 4 8 > $O/shard_c5_cost.txt 2> $O/shard_c5_cost.err
python tools/shard_table.py c5 60 round-robin 8 > $O/shard_c5_rr.txt 2> $O/shard_c5_rr.err
grep -h "max" $O/shard_*.txt | tail -n 20
python bench.py > $O/c3_n1_bench.json 2> $O/c3_n1_bench.err
python bench.py --steps 20 --warmup 5 > $O/c3_n1_bench_driver_flags.json 2>> $O/c3_n1_bench.err
for cfg in c2 c4 c5; do python bench.py --config $cfg --steps 100 --warmup 10 > $O/${cfg}_n1_bench.json 2> $O/${cfg}_n1_bench.err; done
python bench.py --config c6 > $O/c6_n1_bench.json 2> $O/c6_n1_bench.err
python bench.py --config c1 --steps 200 --warmup 20 > $O/c1_n1_bench.json 2> $O/c1_n1_bench.err
python bench.py --config e0102 > $O/e0102_bench.json 2> $O/e0102_bench.err
python bench.py --shard-of 8 --rank 2 > $O/c3_rank2_of_8_bench.json 2> $O/c3_rank2_of_8_bench.err
python tools/gpu/small_fits.py > $O/small_fits.txt 2>&1
python - <<'PY'
import json
for c in ("c1","c2","c3","c4","c5","c6"):
    d=json.load(open(f"gpurun_out/r5x/{c}_n1_bench.json")); print(c, round(d['value'],1), round(d['ms_per_step'],4), d.get('clock_mhz'), d.get('graph_policy'))
PY
