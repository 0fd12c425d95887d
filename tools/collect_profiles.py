#!/usr/bin/env python3
"""Turns the rocprofv3 output of one round (gpurun_out/) into the small files
committed under profiles/: kernel-trace stats, HBM traffic from the PMC passes.

The three passes (see profiles/README.md for the exact commands):
  rocprofv3 --kernel-trace --stats   -> <round>_kernel_stats.csv
  rocprofv3 --pmc FETCH_SIZE         -> read bytes   (x2 on gfx950, see below)
  rocprofv3 --pmc WRITE_SIZE         -> write bytes
FETCH_SIZE/WRITE_SIZE are in KiB.  MI355X_MICROARCH.md (HBM section): on gfx950
FETCH_SIZE reports exactly half the bytes of a wide coalesced streaming read, so
it is doubled; WRITE_SIZE is exact for 16-byte-per-lane streaming stores.
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
  tag = sys.argv[1]                       # e.g. r01
  src = os.path.join(ROOT, 'gpurun_out')
  dst = os.path.join(ROOT, 'profiles')
  os.makedirs(dst, exist_ok=True)
  newest = lambda pattern: max(glob.glob(pattern), key=os.path.getmtime)
  stats = newest(os.path.join(src, 'prof_%s' % tag, '*', '*_kernel_stats.csv'))
  shutil.copy(stats, os.path.join(dst, '%s_kernel_stats.csv' % tag))
  traffic = {}
  # pmc_fetch / pmc_write hold the passes of the default bench; pmc_fetch_<x> /
  # pmc_write_<x> those of other workloads (kernel names are unique per program)
  for counter, prefix in (('FETCH_SIZE', 'pmc_fetch'), ('WRITE_SIZE', 'pmc_write')):
    for folder in sorted(glob.glob(os.path.join(src, prefix + '*'))):
      if not os.path.isdir(folder):
        continue
      f = newest(os.path.join(folder, '*', '*_counter_collection.csv'))
      agg = collections.defaultdict(list)
      for r in csv.DictReader(open(f)):
        if r['Counter_Name'] == counter:
          agg[r['Kernel_Name']].append(float(r['Counter_Value']))
      for kernel, vals in agg.items():
        if kernel.startswith('__amd_rocclr') and folder != os.path.join(src, prefix):
          continue
        traffic.setdefault(kernel, {})[counter] = dict(
            launches=len(vals), mean_KiB=sum(vals) / len(vals), min_KiB=min(vals),
            max_KiB=max(vals))
  out = {}
  for kernel, c in traffic.items():
    if 'FETCH_SIZE' in c and 'WRITE_SIZE' in c:
      read = 2.0 * c['FETCH_SIZE']['mean_KiB'] * 1024
      write = c['WRITE_SIZE']['mean_KiB'] * 1024
      out[kernel] = dict(read_bytes_per_launch=read, write_bytes_per_launch=write,
                         hbm_bytes_per_launch=read + write,
                         fetch_size_raw_KiB=c['FETCH_SIZE']['mean_KiB'],
                         write_size_raw_KiB=c['WRITE_SIZE']['mean_KiB'],
                         launches=c['FETCH_SIZE']['launches'],
                         correction='FETCH_SIZE x2 (gfx950), WRITE_SIZE x1')
  with open(os.path.join(dst, '%s_traffic.json' % tag), 'w') as f:
    json.dump(out, f, indent=1, sort_keys=True)
  p = os.path.join(src, 'prof_%s_bench.log' % tag)
  if os.path.exists(p):
    with open(p) as f:
      lines = [l for l in f if l.startswith('{"metric')]
    with open(os.path.join(dst, '%s_bench_under_rocprof.json' % tag), 'w') as f:
      f.writelines(lines)
  print(json.dumps(out, indent=1))


if __name__ == '__main__':
  main()
