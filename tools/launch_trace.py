#!/usr/bin/env python3
"""Per-launch durations of one sweep from a rocprofv3 kernel trace:
  rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 bench.py --app A \\
      --size ... --iterate N --steps 1 --warmup 0 --cpu-seconds 0
  python3 tools/launch_trace.py DIR A
prints the launches of the LAST sweep in order (kernel, µs, workgroups).  This is
how the 2x anomalies of the XCD super-tile padding were found (docs/DESIGN_HISTORY.md 4.1b)."""
import sys as _sys
if len(_sys.argv) > 1 and _sys.argv[1] in ('-h', '--help'):   # usage = the text above
  print(__doc__)
  _sys.exit(0)
import csv
import glob
import os
import sys

folder, app = sys.argv[1], sys.argv[2]
rows = []
for path in glob.glob(os.path.join(folder, '**', '*kernel_trace.csv'), recursive=True):
  with open(path) as f:
    for r in csv.DictReader(f):
      name = r['Kernel_Name'].split('(')[0]
      if name.startswith(app + '_'):
        grid = int(r.get('Grid_Size_X') or r.get('Grid_Size') or 0)
        block = int(r.get('Workgroup_Size_X') or r.get('Workgroup_Size') or 1)
        rows.append((int(r['Start_Timestamp']), name,
                     (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3,
                     grid // max(1, block)))
rows.sort()
# bench.py runs the same sweep several times (warm-up, timed, per-kernel timing):
# the last sweep = the shortest tail whose sequence of (kernel, workgroups) repeats
keys = [(r[1], r[3]) for r in rows]
period = len(rows)
for p in range(1, len(rows) // 2 + 1):
  if keys[-p:] == keys[-2 * p:-p]:
    period = p
    break
sweep = rows[-period:]
total = 0.0
for i, (_, name, us, groups) in enumerate(sweep):
  total += us
  print('%3d  %-28s %9.1f us  %6d workgroups' % (i, name, us, groups))
print('sum of %d launches: %.1f us' % (len(sweep), total))
