mkdir -p gpurun_out/final
timeout 600 python tools/fuzz_gmm.py 80 21 > gpurun_out/final/fuzz_gmm.txt 2>&1; tail -1 gpurun_out/final/fuzz_gmm.txt
timeout 600 python tools/fuzz_batch.py 60 9 > gpurun_out/final/fuzz_batch.txt 2>&1; tail -1 gpurun_out/final/fuzz_batch.txt
timeout 600 python tools/fuzz_conv.py 120 4 > gpurun_out/final/fuzz_conv.txt 2>&1; tail -1 gpurun_out/final/fuzz_conv.txt
timeout 1700 python -m pytest tests -x -q -m gpu > gpurun_out/final/gputests.log 2>&1; tail -2 gpurun_out/final/gputests.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final/smoke.log 2>&1; tail -2 gpurun_out/final/smoke.log
