#!/usr/bin/env python3
"""Runs bench.py's single-GPU protocol for every sample program and BASELINE
config; writes profiles/<tag>_all_samples.json and prints a markdown table."""
import sys as _sys
if len(_sys.argv) > 1 and _sys.argv[1] in ('-h', '--help'):   # usage = the text above
  print(__doc__)
  _sys.exit(0)
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = [
    ('cfg1 blur 2000x100 x1', ['--app', 'blur', '--size', '2000', '100', '--iterate', '1']),
    ('cfg2 jacobi2d 8192^2 x100', ['--app', 'jacobi2d', '--size', '8192', '8192', '--iterate', '100']),
    ('cfg3 blur 16384^2 x1', ['--app', 'blur', '--size', '16384', '16384', '--iterate', '1']),
    ('cfg4 jacobi2d 16384^2 x1000', ['--app', 'jacobi2d', '--size', '16384', '16384', '--iterate', '1000']),
    ('cfg5 jacobi3d 512^3 x200 (1 GPU)', ['--app', 'jacobi3d', '--size', '512', '512', '512', '--iterate', '200']),
    ('seidel2d 16384^2 x100', ['--app', 'seidel2d', '--size', '16384', '16384', '--iterate', '100']),
    ('sobel2d 16384^2 x1', ['--app', 'sobel2d', '--size', '16384', '16384', '--iterate', '1']),
    ('denoise2d 8192^2 x1', ['--app', 'denoise2d', '--size', '8192', '8192', '--iterate', '1']),
    ('heat3d 512^3 x20', ['--app', 'heat3d', '--size', '512', '512', '512', '--iterate', '20']),
    ('denoise3d 256^3 x1', ['--app', 'denoise3d', '--size', '256', '256', '256', '--iterate', '1']),
]
tag = sys.argv[1] if len(sys.argv) > 1 else 'r01'
rows = []
for name, args in CASES:
  # short steps are repeated until the timed region is ~0.1 s: a burst of a few
  # milliseconds runs at lower clocks (heat3d x20: 3.3 ms per step in a 3-step
  # run, 2.2 ms in a 30-step run)
  steps, warmup = ('3', '1') if 'cfg4' in name else ('30', '10')
  r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', steps,
                      '--warmup', warmup, '--cpu-seconds', '3', '--no-other-configs'] + args,
                     capture_output=True, text=True)
  line = [l for l in r.stdout.splitlines() if l.startswith('{"metric')]
  if not line:
    print(name, 'FAILED', r.stderr[-500:]); continue
  d = json.loads(line[-1]); d['case'] = name; rows.append(d)
  rf = d['roofline']
  print('| %s | %.3f | %.1f | %s | %s | %s %.2f | alg %.2f | hbm %s | valu %.2f | %.1f (%d thr) |' % (
      name, d['ms_per_step'], d['value'], d['config']['depth_schedule'], rf['kernel'],
      rf['bound'], rf['frac'], rf['frac_algorithmic'],
      '%.2f' % rf['hbm_measured_frac'] if rf.get('hbm_measured_frac') else 'n/a',
      rf['valu_frac'], d['cpu_baseline']['value'], d['cpu_baseline']['cores']), flush=True)
os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
with open(os.path.join(ROOT, 'gpurun_out', '%s_all_samples.json' % tag), 'w') as f:
  json.dump(rows, f, indent=1)
