#!/bin/bash
O=gpurun_out/r5final2; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?"
python -m pytest tests -x -q -m gpu > $O/full.log 2>&1; echo "rc=$?" >> $O/full.log
tail -n 4 $O/full.log
bash tools/gpu/r5_profile_lite.sh r05 > $O/profile.log 2>&1; echo "profile rc=$?"; tail -n 5 $O/profile.log
