export TUNE_WARMUP=2 TUNE_REPEATS=4
python tools/tune.py seidel2d 16384 320 16,4,256,3 16,4,256,3,ring=12,max_period=12 16,4,256,3,waves_per_eu=3 16,4,256,3,ring=12,max_period=12,waves_per_eu=3 16,4,256,3,wave_groups=8,pairs=2,ring=6,vgpr_budget=400 16,4,256,3,wave_groups=8,pairs=2,ring=12,max_period=12,vgpr_budget=400,waves_per_eu=4 2>&1 | grep -v amdgpu
python tools/tune.py seidel2d 16384 240 12,4,256,3 12,4,256,3,wave_groups=6,pairs=2,ring=6,vgpr_budget=400 2>&1 | grep -v amdgpu
