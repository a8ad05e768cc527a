for e in 1024 2048 4096; do for k in 17 21 25 29 33; do python tools/conv_bench.py $e $k direct,fft 2>&1 | grep " us"; done; done
