mkdir -p gpurun_out/fft6
timeout 900 python -m pytest tests/test_gpu_fft_native.py -x -q > gpurun_out/fft6/tests.txt 2>&1
tail -4 gpurun_out/fft6/tests.txt
python bench.py --config c6 > gpurun_out/fft6/c6.json 2> gpurun_out/fft6/c6.err
python3 -c "
import json; d=json.loads(open('gpurun_out/fft6/c6.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['host_enqueue_ms_per_step'], d['kernel_ms_per_step'])"
python bench.py --config c4 --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/fft6/c4.json 2> gpurun_out/fft6/c4.err
python3 -c "
import json; d=json.loads(open('gpurun_out/fft6/c4.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], {k:(v['value'],v['ms_per_step']) for k,v in d.items() if k in ('general_psf','fft_psf')})"
