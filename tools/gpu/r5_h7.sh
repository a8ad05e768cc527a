#!/bin/bash
O=gpurun_out/r5h7; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?"
python -m pytest tests -x -q -m gpu > $O/full.log 2>&1; echo "rc=$?" >> $O/full.log
tail -n 4 $O/full.log
for c in c3 c5 c6; do timeout 300 python tools/gpu/r5_h1.py $c 6 40 > $O/prio_$c.txt 2> $O/prio_$c.err; grep -v amdgpu.ids $O/prio_$c.txt; done
