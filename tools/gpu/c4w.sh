mkdir -p gpurun_out/c4w
timeout 900 python tools/ab.py c4 3 20 -- base: c2r76:JD_SEP_WALK_COLS=2,JD_SEP_WALK_ROWS=76 c2r58:JD_SEP_WALK_COLS=2,JD_SEP_WALK_ROWS=58 c2r112:JD_SEP_WALK_COLS=2,JD_SEP_WALK_ROWS=112 c4r56:JD_SEP_WALK_COLS=4,JD_SEP_WALK_ROWS=56 c4r74:JD_SEP_WALK_COLS=4,JD_SEP_WALK_ROWS=74 > gpurun_out/c4w/ab.txt 2>&1
grep step gpurun_out/c4w/ab.txt | cut -c1-200
