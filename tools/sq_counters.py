#!/usr/bin/env python3
"""Per-launch averages of the SQ counters tools/sq_counters.sh collected, by kernel,
with the shares the DESIGN tables quote (VALU issue utilisation, where a wavefront
spends its life).  usage: sq_counters.py <directory of pass*/ outputs>"""
import sys as _sys
if len(_sys.argv) > 1 and _sys.argv[1] in ('-h', '--help'):   # usage = the text above
  print(__doc__)
  _sys.exit(0)
import csv
import glob
import json
import os
import sys

root = sys.argv[1]
sums, launches = {}, {}
for path in glob.glob(os.path.join(root, 'pass*', '**', '*counter_collection.csv'),
                      recursive=True):
  seen = {}
  with open(path) as f:
    for row in csv.DictReader(f):
      kernel = row['Kernel_Name'].split('(')[0]
      key = (kernel, row['Counter_Name'])
      sums[key] = sums.get(key, 0.0) + float(row['Counter_Value'])
      seen.setdefault(key, set()).add(row['Dispatch_Id'])
  for key, ids in seen.items():
    launches[key] = launches.get(key, 0) + len(ids)
out = {}
for (kernel, counter), total in sorted(sums.items()):
  out.setdefault(kernel, {})[counter] = total / launches[(kernel, counter)]
  out[kernel].setdefault('_launches', launches[(kernel, counter)])
for kernel, c in out.items():
  if 'GRBM_GUI_ACTIVE' in c and 'SQ_ACTIVE_INST_VALU' in c and 'SQ_WAVE_CYCLES' in c:
    shader = c['GRBM_GUI_ACTIVE'] / 8.0          # summed over the 8 XCDs
    simds = 1024.0
    c['_derived'] = dict(
        shader_cycles=shader,
        # SQ_ACTIVE_INST_VALU counts quad-cycles: x4 = cycles a SIMD spent issuing
        valu_issue_utilisation=c['SQ_ACTIVE_INST_VALU'] * 4 / simds / shader,
        valu_cycles_per_instruction=c['SQ_ACTIVE_INST_VALU'] * 4 / max(1.0, c.get(
            'SQ_INSTS_VALU', 0.0)),
        wave_issuing=c['SQ_ACTIVE_INST_ANY'] * 4 / c['SQ_WAVE_CYCLES'] / 4,
        wave_parked=(c['SQ_WAIT_ANY'] - 0.0) / c['SQ_WAVE_CYCLES'],
        wave_issue_stalled=c['SQ_WAIT_INST_ANY'] / c['SQ_WAVE_CYCLES'],
        lds_busy=c.get('SQ_LDS_IDX_ACTIVE', 0.0) / (shader * 256),
        lds_conflict_share=c.get('SQ_LDS_BANK_CONFLICT', 0.0) / max(
            1.0, c.get('SQ_LDS_IDX_ACTIVE', 0.0)))
json.dump(out, sys.stdout, indent=1, sort_keys=True)
