mkdir -p gpurun_out/ilv
timeout 600 python tools/ab.py c3 4 30 -- base: ilv:JD_SEP_INTERLEAVE=1 > gpurun_out/ilv/ab.txt 2>&1
grep step gpurun_out/ilv/ab.txt | cut -c1-330
timeout 600 python tools/ab.py c4 3 20 -- base: ilv:JD_SEP_INTERLEAVE=1 > gpurun_out/ilv/ab4.txt 2>&1
grep step gpurun_out/ilv/ab4.txt | cut -c1-330
