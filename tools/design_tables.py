#!/usr/bin/env python3
"""Prints the two markdown tables of DESIGN.md section 6 from the committed
profiles (profiles/<tag>_all_samples.json, <tag>_traffic.json + kernel times)."""
import sys as _sys
if len(_sys.argv) > 1 and _sys.argv[1] in ('-h', '--help'):   # usage = the text above
  print(__doc__)
  _sys.exit(0)
import json
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else 'r02'
rows = json.load(open(os.path.join(ROOT, 'profiles', '%s_all_samples.json' % tag)))
print('| workload (1 × MI355X) | ms | G updates/s (valid) | launches | dominant kernel | bound: frac | alg | hbm (PMC) | valu | CPU G/s |')
print('|---|---|---|---|---|---|---|---|---|---|')
for r in rows:
  rf = r['roofline']
  print('| %s | %.3f | %.0f | %s | %s | %s %.2f | %.2f | %s | %.2f | %.1f |' % (
      r['case'], r['ms_per_step'], r['value'],
      r['config']['depth_schedule'].replace('denoise3d_stage_', 'stage '), rf['kernel'],
      rf['bound'], rf['frac'], rf['frac_algorithmic'],
      '%.2f' % rf['hbm_measured_frac'] if rf.get('hbm_measured_frac') else '—',
      rf['valu_frac'], r['cpu_baseline']['value']))
print()
traffic = json.load(open(os.path.join(ROOT, 'profiles', '%s_traffic.json' % tag)))['entries']
times = {}
for r in rows:
  rf = r['roofline']
  times[(rf['kernel'], tuple(r['config']['dims']), r['config']['iterate'])] = rf['kernel_avg_us']
print('| kernel (workload) | launches | read MB | write MB | read ÷ write | avg launch µs | HBM rate | of 8 TB/s |')
print('|---|---|---|---|---|---|---|---|')
for e in traffic:
  t = times.get((e['kernel'], tuple(e['dims']), e['iterate']))
  rate = (e['hbm_bytes_per_launch'] / t / 1e6) if t else None
  print('| %s (%s) | %d | %.0f | %.0f | %.2f | %s | %s | %s |' % (
      e['kernel'], e['workload'], e['launches'], e['read_bytes_per_launch'] / 1e6,
      e['write_bytes_per_launch'] / 1e6, e['read_over_write'],
      '%.0f' % t if t else '—', '%.2f TB/s' % rate if rate else '—',
      '%.2f' % (rate / 8.0) if rate else '—'))
