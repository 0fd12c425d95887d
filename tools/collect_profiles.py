#!/usr/bin/env python3
"""Turns the rocprofv3 output of one round (gpurun_out/, written by
tools/collect_on_gpu.sh) into the small files committed under profiles/:

  <round>_kernel_stats.csv          --kernel-trace --stats summary of bench.py
  <round>_bench_under_rocprof.json  the bench line printed during that pass
  <round>_traffic.json              HBM bytes per launch from the PMC passes
  <round>_sq_counters.json          SQ counters per kernel (tools/sq_counters.sh): VALU
                                    issue utilisation, wavefront time shares

The traffic file holds one entry per (kernel, grid, iteration count): a figure
measured on one grid says nothing about another, so bench.py only quotes an entry
whose kernel AND dims AND iterate match its own run, whose `kernel_digest` (program +
kernel shape, kernel.calibration_key) is that of the kernel it runs, and - a
per-launch AVERAGE depends on which launches of the sweep went to that kernel - whose
`launches` equals the number of launches the kernel has in its own sweep.  The
averages are over the ONE timed sweep of the pass (the last `launches_per_step`
dispatches), not over the sweeps the tuning step tried before it.  `commit` says which
tree the kernels were built from.  FETCH_SIZE / WRITE_SIZE are in KiB.
MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE reports half the bytes of a
coalesced streaming read, so it is doubled; WRITE_SIZE is exact.  The rule was
re-checked for 4-, 8- and 16-byte-per-lane copies with tools/pmc_calib.hip
(512 MiB read -> FETCH_SIZE 256 MiB in all three; profiles/r02_pmc_calibration.txt).
"""
import sys as _sys
if len(_sys.argv) > 1 and _sys.argv[1] in ('-h', '--help'):   # usage = the text above
  print(__doc__)
  _sys.exit(0)
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def counter_per_kernel(folder, counter, last=None):
  """Counter values per kernel, in dispatch order; `last` = N keeps only the last N
  dispatches of the run - the one timed sweep of `bench.py --steps 1`, not the sweeps
  soda_hip_plan_tune tried before it (other splits, other kernels per launch)."""
  files = glob.glob(os.path.join(folder, '*', '*_counter_collection.csv'))
  if not files:
    return {}
  rows = [r for r in csv.DictReader(open(max(files, key=os.path.getmtime)))
          if r['Counter_Name'] == counter and not r['Kernel_Name'].startswith('__amd_rocclr')]
  rows.sort(key=lambda r: int(r['Dispatch_Id']))
  if last:
    rows = rows[-last:]
  agg = collections.defaultdict(list)
  for r in rows:
    agg[r['Kernel_Name']].append(float(r['Counter_Value']))
  return agg


def launches_per_step(log_path):
  """config.launches_per_step of the bench line a PMC pass printed."""
  try:
    with open(log_path) as f:
      for line in f:
        if line.startswith('{"metric'):
          return int(json.loads(line)['config']['launches_per_step'])
  except (OSError, ValueError, KeyError):
    pass
  return None


def kernel_digests(app):
  """kernel name -> kernel.calibration_key of the kernels the shipped blob of `app`
  is generated from (program + kernel shape): bench.py quotes a profile entry only for
  the very kernel it was measured on."""
  sys.path[:0] = [ROOT, os.path.join(ROOT, 'soda-compiler_amd')]
  import __graft_entry__ as entry
  from soda_hip import frontend
  from soda_hip.codegen import kernel, spec as specmod
  st = frontend.load(entry.sample_path(app), iterate=entry.BLOB_ITERATE.get(app))
  spec = specmod.spec_from_stencil(st)
  return {e['name']: kernel.calibration_key(e, spec) for e in kernel.generate(spec)[1]}


def main():
  tag = sys.argv[1]                       # e.g. r02
  src = os.path.join(ROOT, 'gpurun_out')
  dst = os.path.join(ROOT, 'profiles')
  os.makedirs(dst, exist_ok=True)
  stats = glob.glob(os.path.join(src, 'prof_%s' % tag, '*', '*_kernel_stats.csv'))
  if stats:
    shutil.copy(max(stats, key=os.path.getmtime),
                os.path.join(dst, '%s_kernel_stats.csv' % tag))
  # the tree the GPU box ran: stamped by tools/collect.sh when the call was made (the
  # snapshot has no .git) - never this checkout's state at post-processing time
  try:
    with open(os.path.join(src, 'pmc_%s_commit.txt' % tag)) as f:
      commit = f.read().strip() or None
  except OSError:
    commit = None
  entries = []
  with open(os.path.join(src, 'pmc_%s_workloads.txt' % tag)) as f:
    # name|bench arguments[|given split] (tools/collect_on_gpu.sh)
    workloads = [tuple(line.strip().split('|')[:2]) for line in f if line.strip()]
  for name, args in workloads:
    words = args.split()
    app = words[words.index('--app') + 1]
    i = words.index('--size') + 1
    dims = []
    while i < len(words) and not words[i].startswith('--'):
      dims.append(int(words[i]))
      i += 1
    iterate = int(words[words.index('--iterate') + 1])
    n_f = launches_per_step(os.path.join(src, 'pmc_%s_%s_FETCH_SIZE.log' % (tag, name)))
    n_w = launches_per_step(os.path.join(src, 'pmc_%s_%s_WRITE_SIZE.log' % (tag, name)))
    fetch = counter_per_kernel(os.path.join(src, 'pmc_%s_%s_FETCH_SIZE' % (tag, name)),
                               'FETCH_SIZE', last=n_f)
    write = counter_per_kernel(os.path.join(src, 'pmc_%s_%s_WRITE_SIZE' % (tag, name)),
                               'WRITE_SIZE', last=n_w)
    digests = kernel_digests(app)
    for kernel in sorted(set(fetch) & set(write)):
      # (the tuning step may settle on a neighbouring split in the two passes - cfg4:
      # 41x24+1x16 in one, 40x24+2x20 in the other; launches of one kernel differ little
      # in size, so per-kernel averages from schedules one launch apart are kept)
      # Launches whose counters agree within 5 % (2-D sweeps: the box shrinks by a
      # few rows per launch) are `uniform`: their average does not depend on which
      # launches of the sweep the kernel got, so a pass that tuned itself to another
      # split still speaks for the kernel.
      spread = max((max(v) - min(v)) / max(v) for v in (fetch[kernel], write[kernel]))
      uniform = spread < 0.05
      if n_f is None or n_w is None or (not uniform and abs(
          len(fetch[kernel]) - len(write[kernel])) > max(1, len(fetch[kernel]) // 20)):
        print('%s %s: the two passes ran different schedules (%s / %s launches per step, '
              '%d / %d of this kernel): skipped' % (
                  name, kernel, n_f, n_w, len(fetch[kernel]), len(write[kernel])))
        continue
      read_b = 2.0 * sum(fetch[kernel]) / len(fetch[kernel]) * 1024
      write_b = sum(write[kernel]) / len(write[kernel]) * 1024
      entries.append(dict(
          workload=name, app=app, kernel=kernel, dims=dims, iterate=iterate,
          launches=len(fetch[kernel]), launches_write_pass=len(write[kernel]),
          launches_per_step=n_f, uniform=uniform,
          kernel_digest=digests.get(kernel), read_bytes_per_launch=read_b,
          write_bytes_per_launch=write_b, hbm_bytes_per_launch=read_b + write_b,
          read_over_write=read_b / write_b if write_b else None,
          fetch_size_raw_KiB=sum(fetch[kernel]) / len(fetch[kernel]),
          write_size_raw_KiB=sum(write[kernel]) / len(write[kernel]),
          correction='FETCH_SIZE x2 (gfx950), WRITE_SIZE x1', commit=commit))
  with open(os.path.join(dst, '%s_traffic.json' % tag), 'w') as f:
    json.dump(dict(entries=entries), f, indent=1, sort_keys=True)
  sq_entries = []
  for name, args in workloads:
    path = os.path.join(src, 'sq_%s_%s' % (tag, name), 'summary.json')
    if not os.path.exists(path):
      continue
    words = args.split()
    i = words.index('--size') + 1
    dims = []
    while i < len(words) and not words[i].startswith('--'):
      dims.append(int(words[i]))
      i += 1
    iterate = int(words[words.index('--iterate') + 1])
    with open(path) as f:
      summary = json.load(f)
    app = words[words.index('--app') + 1]
    digests = kernel_digests(app)
    for kernel, counters in sorted(summary.items()):
      derived = counters.get('_derived')
      if not derived or '_fused_' not in kernel and '_stage_' not in kernel:
        continue
      sq_entries.append(dict(
          workload=name, kernel=kernel, dims=dims, iterate=iterate, commit=commit,
          launches=counters.get('_launches'), kernel_digest=digests.get(kernel),
          valu_instructions=counters.get('SQ_INSTS_VALU'),
          **{k: derived[k] for k in ('valu_issue_utilisation', 'valu_cycles_per_instruction',
                                     'wave_issuing', 'wave_issue_stalled', 'wave_parked',
                                     'lds_busy', 'lds_conflict_share', 'shader_cycles')}))
  if sq_entries:
    with open(os.path.join(dst, '%s_sq_counters.json' % tag), 'w') as f:
      json.dump(dict(entries=sq_entries,
                     note='per-launch averages of rocprofv3 --pmc passes (tools/sq_counters.sh); '
                          'wave_* = shares of SQ_WAVE_CYCLES'), f, indent=1, sort_keys=True)
    for e in sq_entries:
      print('%-10s %-28s VALU busy %.2f  issuing %.2f  issue-stalled %.2f  parked %.2f' % (
          e['workload'], e['kernel'], e['valu_issue_utilisation'], e['wave_issuing'],
          e['wave_issue_stalled'], e['wave_parked']))
  p = os.path.join(src, 'prof_%s_bench.log' % tag)
  if os.path.exists(p):
    with open(p) as f:
      lines = [l for l in f if l.startswith('{"metric')]
    # the bench line of the --stats pass WITH the summary rows of its kernels and the
    # cross-check the two allow: sum over the sweep's launches of the summary's average
    # duration of each launch's kernel against the line's own ms_per_step
    record = dict(bench=json.loads(lines[-1]) if lines else None, commit=commit,
                  kernel_stats=[], check=None)
    if stats and lines:
      app = record['bench']['config']['app']
      with open(max(stats, key=os.path.getmtime)) as f:
        rows = {r['Name'].split('(')[0]: r for r in csv.DictReader(f)}
      for name, r in sorted(rows.items()):
        if name.startswith(app + '_'):
          record['kernel_stats'].append(dict(
              name=name, calls=int(r['Calls']), average_ns=float(r['AverageNs']),
              total_ns=float(r['TotalDurationNs']), min_ns=float(r['MinNs']),
              max_ns=float(r['MaxNs'])))
      total_ns, per_step = 0.0, {}
      for part in record['bench']['config']['depth_schedule'].split('+'):
        count, _, depth = part.partition('x')
        name = '%s_fused_k%s' % (app, depth)
        mine = [k for k in record['kernel_stats'] if k['name'] == name]
        if mine:
          total_ns += int(count) * mine[0]['average_ns']
          per_step[name] = per_step.get(name, 0) + int(count)
      sweeps = record['bench']['steps'] + record['bench']['warmup'] + 3
      record['check'] = dict(
          launches_per_step=per_step, sweeps_in_the_pass=sweeps,
          calls_expected={n: c * sweeps for n, c in per_step.items()},
          sum_of_average_ns_ms=total_ns / 1e6, ms_per_step=record['bench']['ms_per_step'],
          ratio=total_ns / 1e6 / record['bench']['ms_per_step'])
      print('--stats cross-check: sum(launches x AverageNs) = %.3f ms, ms_per_step = %.3f '
            '(ratio %.3f)' % (total_ns / 1e6, record['bench']['ms_per_step'],
                              record['check']['ratio']))
    with open(os.path.join(dst, '%s_bench_under_rocprof.json' % tag), 'w') as f:
      json.dump(record, f, indent=1, sort_keys=True)
      f.write('\n')
  for e in entries:
    print('%-10s %-28s %-18s read %8.1f MB  write %8.1f MB  read/write %.2f  (%d launches)' % (
        e['workload'], e['kernel'], 'x'.join(map(str, e['dims'])),
        e['read_bytes_per_launch'] / 1e6, e['write_bytes_per_launch'] / 1e6,
        e['read_over_write'], e['launches']))


if __name__ == '__main__':
  main()
