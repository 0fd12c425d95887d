python -m pytest tests/test_gpu_parity.py -x -q -k "3d or heat or denormal" 2>&1 | tail -2
python bench.py --app heat3d --size 512 512 512 --iterate 20 --steps 30 --warmup 10 --cpu-seconds 0 2>&1 | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('heat3d blob ms', round(d['ms_per_step'],4), d['roofline']['kernel'], round(d['roofline']['kernel_avg_us'],1))"
python bench.py --app jacobi3d --size 512 512 512 --iterate 200 --steps 10 --warmup 3 --cpu-seconds 0 2>&1 | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('cfg5 blob ms', round(d['ms_per_step'],4), d['roofline']['kernel'], round(d['roofline']['kernel_avg_us'],1))"
TUNE_HIPCC=1 python tools/tune3d.py heat3d 512 20 "" wp_prio=0 wp_waves_per_eu=2 2>&1 | grep us/sweep
TUNE_HIPCC=1 python tools/tune3d.py jacobi3d 512 200 "" wp_prio=0 wp_prio=3/2/1/0 2>&1 | grep us/sweep
