export TUNE_WARMUP=2 TUNE_REPEATS=4 TUNE_HIPCC=1
python tools/tune.py jacobi2d 16384 320 16,4,256,3 16,4,256,3,ringpk=1 16,4,256,3 16,4,256,3,ringpk=1 2>&1 | grep "Gcell\|FAILED" | cut -c1-200
python tools/tune.py jacobi2d 16384 240 12,4,256,3 12,4,256,3,ring=12,max_period=12,waves_per_eu=4 12,4,256,3,ringpk=1 12,4,256,3,wave_groups=0 2>&1 | grep "Gcell\|FAILED" | cut -c1-200
python tools/tune.py seidel2d 16384 320 16,4,256,3 16,4,256,3,ring=12,max_period=12 16,4,256,3,ringpk=1 2>&1 | grep "Gcell\|FAILED" | cut -c1-200
python tools/tune.py seidel2d 16384 240 12,4,256,3 12,4,256,3,ring=6,max_period=6,waves_per_eu=3 12,4,256,3,ringpk=1 2>&1 | grep "Gcell\|FAILED" | cut -c1-200
python tools/tune.py jacobi2d 16384 160 8,4,256,3 8,4,256,3,wave_groups=4,pairs=2,ring=6,vgpr_budget=400 8,4,256,3,wave_groups=4,pairs=2,ring=12,max_period=12,vgpr_budget=400 2>&1 | grep "Gcell\|FAILED" | cut -c1-200
