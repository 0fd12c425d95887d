#!/usr/bin/env python3
"""Headline benchmark: jacobi2d float 16384 x 16384, iterate 1000 (BASELINE.json
configs[3], the configuration the metric is quoted on; it fits one MI355X).

  python bench.py --gpus N --steps K --warmup W

A "step" is one whole 1000-iteration sweep of the grid, input already resident
in HBM.  N > 1 is launched by torch.distributed.run, one rank per GPU: the grid
is cut into N slabs along the outer dimension (strong scaling: the grid is fixed)
and neighbours exchange ghost rows over RCCL every E iterations; E and the order of
exchange and sweeps (serial, or overlapped on a side stream) are the fastest of a few
candidates timed during warm-up (`config.exchange_candidates_ms`), unless `--exchange`
/ `--overlap` give them.

Prints ONE JSON line (rank 0).  `value` = valid cell-updates / wall time, valid =
the cells the reference semantics define (the box shrinks by the stencil radius
every iteration).  `roofline` describes the dominant kernel
(roofline_block below): `bound` names its binding physical limit - HBM traffic
(PMC bytes of this kernel on this grid, from profiles/rNN_traffic.json) or VALU
issue (useful f32 lane-operations against the chip's 78.6 T/s) - and `frac` the
fraction of that limit's peak; `frac_algorithmic` is SURVEY 8(d)'s figure
(algorithmic bytes per update / time / 8 TB/s), which temporal blocking pushes
past 1 and which is therefore not a bound.  `cpu_baseline` is the CPU oracle (a
port of the loop nest the reference emits into `<app>_test`) timed on this host's
cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, 'soda-compiler_amd')):
  if p not in sys.path:
    sys.path.insert(0, p)

import numpy as np  # noqa: E402

SEED = 20240607
HBM_PEAK_GBPS = 8000.0   # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)


def parse_args():
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=3)
  ap.add_argument('--warmup', type=int, default=1)
  ap.add_argument('--app', default='jacobi2d')
  ap.add_argument('--size', type=int, nargs='+', default=[16384, 16384],
                  help='grid extents, fastest dimension first')
  ap.add_argument('--iterate', type=int, default=1000)
  ap.add_argument('--max-depth', type=int, default=0,
                  help='cap on fused depth (0 = deepest in the blob)')
  ap.add_argument('--exchange', type=int, default=0,
                  help='N > 1: iterations between halo exchanges (0 = the fastest of '
                       '1, 2, 4, 8 x the deepest fused kernel, timed during warm-up)')
  ap.add_argument('--cpu-seconds', type=float, default=12.0,
                  help='CPU baseline sample budget (0 = skip)')
  ap.add_argument('--cpu-baseline-at-all-n', action='store_true',
                  help='N > 1: rank 0 times the CPU baseline as well (by default it is on '
                       'the N = 1 line only: it is a property of the host, and 12 s of every '
                       'multi-GPU run would go into it)')
  ap.add_argument('--force-dist', action='store_true',
                  help='take the torch.distributed slab path even with 1 rank')
  ap.add_argument('--overlap', action='store_true',
                  help='N > 1: exchange on a side stream beside the interior '
                       'sweep (boundary bands first), with the default exchange '
                       'period unless --exchange gives one; without either flag both '
                       'orders are timed during warm-up and the faster one runs')
  ap.add_argument('--no-exchange-tune', action='store_true',
                  help='N > 1: keep the default exchange period and the serial order '
                       'instead of timing the candidate (period, order) pairs during '
                       'warm-up')
  ap.add_argument('--static-cut', action='store_true',
                  help='N > 1: every rank keeps the rows of the even cut of the whole grid '
                       'for the whole run; default: the rows each super-step defines are cut '
                       'evenly again (the valid box shrinks every iteration: a static cut '
                       'idles the first and last ranks)')
  ap.add_argument('--recut', action='store_true',
                  help='N > 1: re-cut slabs, without timing the chosen exchange period and '
                       'order under the static cut as well (the default keeps the faster cut)')
  ap.add_argument('--no-tune', action='store_true',
                  help='split `iterate` into fused depths by the calibrated model alone '
                       'instead of timing the candidate splits on this grid during '
                       'warm-up (soda_hip_plan_tune)')
  ap.add_argument('--split', default='',
                  help="the split of `iterate` into fused depths, fixed: '41x24+1x16' as "
                       'config.depth_schedule of an earlier run prints it (profiling '
                       'passes repeat the schedule of the timed run with it); implies '
                       '--no-tune (soda_hip_plan_set_split)')
  ap.add_argument('--no-other-configs', action='store_true',
                  help='N = 1, headline workload: do not time BASELINE configs 2, 3 and 5 '
                       'after it (config.other_configs)')
  ap.add_argument('--jit', action='store_true',
                  help='compile the kernels with hiprtc instead of loading the '
                       'code object built by __graft_entry__.build()')
  return ap.parse_args()


def open_program(app, iterate, jit):
  from soda_hip import frontend
  from soda_hip.codegen import kernel
  from soda_hip.codegen import spec as specmod
  from soda_hip.runtime import host
  st = frontend.load(os.path.join(ROOT, 'tests', 'samples', app + '.soda'),
                     iterate=iterate)
  spec = specmod.spec_from_stencil(st)
  blob = os.path.join(ROOT, 'soda-compiler_amd', 'blobs', app + '.hsaco')
  if jit or not os.path.exists(blob):
    text, _ = kernel.generate(spec)
    return host.open_program(source=text, spec=spec), spec
  return host.open_program(blob=blob, spec=spec), spec


def make_input(spec, dims, rows=None):
  """Seeded uniform [0,1) floats / full-range ints (SURVEY.md 8d pattern B);
  `rows` = (first, last) of the outer dimension to materialise."""
  from soda_hip.codegen import spec as specmod
  shape = tuple(reversed(dims))
  out = []
  for t in spec['inputs']:
    dt = np.dtype(specmod.NUMPY_NAME[t['c_type']])
    rng = np.random.default_rng(SEED)
    wide = dt.kind != 'f' and dt.itemsize > 4     # 64-bit draws: no cheap skip
    if rows is not None and not wide:
      # only this rank's rows are materialised: the generator is moved past the
      # rows before them (PCG64 makes two 32-bit draws of every 64-bit step, for
      # float32 and for integer ranges up to 2^32 alike; advance() also drops a
      # buffered half) - the values equal the whole grid's rows [first, last)
      # (tests/test_bench_input.py), without 1 GiB of host memory per rank
      row = int(np.prod(shape[1:], dtype=np.int64))
      skip = rows[0] * row
      rng.bit_generator.advance(skip // 2)
      if skip % 2:
        rng.random(1, dtype=np.float32)
      shape = (rows[1] - rows[0],) + shape[1:]
    if dt.kind == 'f':
      a = rng.random(shape, dtype=np.float32).astype(dt, copy=False)
    else:
      a = rng.integers(0, np.iinfo(dt).max + 1, size=shape).astype(dt)
    if rows is not None and wide:
      a = np.ascontiguousarray(a[rows[0]:rows[1]])
    out.append(a)
  return out


def cpu_baseline(spec, dims, budget_s):
  """The CPU oracle on this host: a bounded number of whole-grid sweeps, OpenMP
  team = the CPUs the cgroup grants (a box can show 256 logical CPUs and grant 16:
  more threads than the quota only time-slice), each thread PINNED to a physical core
  of its own (neighbouring cores; oracle/pin_threads.c) so that the pages the parallel
  first touch placed stay next to the threads that use them - rounds 2-5 let the
  threads wander and saw 13.6 .. 47 G/s on the same quota.  One discarded sample, then
  three: the MEDIAN is reported, the best and the spread beside it."""
  from oracle import soda_oracle
  orc = soda_oracle.Oracle(spec, flags=('-O3', '-march=native'))
  inputs = make_input(spec, dims)
  quota = orc.cpu_quota()
  logical = len(os.sched_getaffinity(0))
  orc.set_threads(quota)
  mask = os.sched_getaffinity(0)
  cpus = orc.team_cpus(quota)
  attempted = len(cpus) == quota
  pinned = attempted and orc.pin_threads(cpus) == 0
  try:
    t, u = orc.time_iterations(inputs, 2, warmup=1)      # cold: sizes the discarded sample
    n = int(max(3, min(5000, budget_s / 8.0 / max(t / 2, 1e-6))))
    t, u = orc.time_iterations(inputs, n, warmup=1)      # discarded: sizes the three kept
    samples, seconds = [], t
    n = int(max(3, min(20000, budget_s * 0.75 / 3.0 / max(t / n, 1e-6))))
    for _ in range(3):     # ~a quarter of the budget each: short samples are noisy ones
      t, u = orc.time_iterations(inputs, n, warmup=1)
      seconds += t
      samples.append(u / t / 1e9)
  finally:
    if attempted:        # (also after a partial failure: some threads may be pinned)
      orc.unpin_threads(mask, quota)
  samples.sort()
  median = samples[1]
  return dict(value=median, unit='Gcell-updates/s', cores=quota, kind='port',
              best=samples[-1], spread=(samples[-1] - samples[0]) / median,
              pinned=bool(pinned), sweeps_per_sample=n, seconds=seconds,
              logical_cpus=logical, flags='g++ -O3 -march=native -ffp-contract=off',
              # (at most 120 characters: what the driver's record keeps of a string)
              sample='median of 3 samples (1 discarded) x %d sweeps of %s, %d pinned '
                     'threads' % (n, 'x'.join(map(str, dims)), quota),
              samples=samples)


def profile_entry_matches(entry, kernel, dims, iterate, digest, launches):
  """A committed profile entry speaks for this run only if it was measured on the same
  kernel (name AND shape digest, kernel.calibration_key), the same grid and iteration
  count, and - per-launch averages depend on which launches of the sweep the kernel
  got - about the same number of launches per sweep (one apart is tolerated: the
  tuning step settles on neighbouring splits from run to run), unless the entry's
  launches were all of one size within 5 % (`uniform`: 2-D sweeps)."""
  return entry['kernel'] == kernel and list(entry['dims']) == list(dims) and \
      entry['iterate'] == iterate and entry.get('kernel_digest') == digest and \
      digest is not None and (
          entry.get('uniform') or     # launches of (nearly) one size: any share will do
          abs(entry.get('launches', -99) - launches) <= max(1, launches // 20))


def measured_traffic(kernel, dims, iterate, digest=None, launches=0):
  """HBM bytes per launch of `kernel` ON THIS GRID from the newest committed
  rocprofv3 PMC passes (profiles/rNN_traffic.json, written by
  tools/collect_profiles.py from separate --pmc runs of this same command: PMC
  counters cannot be collected from inside the timed process).  Entries are
  keyed by kernel, grid and iteration count; a shape that was not profiled gives
  None - a figure taken on another grid says nothing about this one."""
  import glob
  files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_traffic.json')))
  for path in reversed(files):
    with open(path) as f:
      data = json.load(f)
    for entry in data.get('entries', []):
      if profile_entry_matches(entry, kernel, dims, iterate, digest, launches):
        return entry, os.path.basename(path)
  return None, None


def measured_counters(kernel, dims, iterate, digest=None):
  """SQ counters of `kernel` on this grid from the newest committed PMC passes
  (profiles/rNN_sq_counters.json, tools/sq_counters.sh + collect_profiles.py):
  VALU issue utilisation and where a wavefront spends its life."""
  import glob
  for path in reversed(sorted(glob.glob(os.path.join(ROOT, 'profiles',
                                                     'r*_sq_counters.json')))):
    with open(path) as f:
      data = json.load(f)
    for entry in data.get('entries', []):
      if entry['kernel'] == kernel and list(entry['dims']) == list(dims) and \
          entry['iterate'] == iterate and digest is not None and \
          entry.get('kernel_digest') == digest:
        return entry, os.path.basename(path)
  return None, None


def per_iteration_updates(spec, dims, iterate, rows=None):
  """Valid cell-updates of each iteration; `rows` = (first, last) restricts the
  count to those rows of the outermost dimension (one rank's own rows)."""
  from soda_hip.codegen import spec as specmod
  out = []
  for lo, hi in specmod.iteration_margins(spec, iterate):
    cells = 1
    for d in range(spec['dim'] - 1):
      cells *= max(0, dims[d] - lo[d] - hi[d])
    first, last = lo[-1], dims[-1] - hi[-1]
    if rows is not None:
      first, last = max(first, rows[0]), min(last, rows[1])
    out.append(cells * max(0, last - first))
  return out


# MI355X_MICROARCH.md: 256 CUs x 4 SIMD-32 x 2.4 GHz = 78.6 T f32 lane-operations
# per second for adds and multiplies (a packed v_pk_*_f32 does two per lane in
# twice the cycles: the same rate); FMA would double it but the reference's
# arithmetic is not contracted.
VALU_PEAK_TLANEOPS = 256 * 4 * 32 * 2.4e9 / 1e12


# MI355X_MICROARCH.md: 6.29 TB/s measured for a float4 copy = what "all of HBM" looks
# like from a kernel; used only to compare how BUSY the two limits are
HBM_ACHIEVABLE_FRAC = 6.29 / 8.0


ROOFLINE_KEY_ORDER = ('kernel', 'bound', 'frac', 'unit', 'achieved', 'peak',
                      'frac_algorithmic', 'hbm_measured_frac', 'valu_frac', 'traffic',
                      'kernel_avg_us', 'kernel_launches', 'shader_clock_ghz',
                      'valu_issue_utilisation', 'useful_instruction_share',
                      'hbm_floor_frac', 'updates_per_launch',
                      'algorithmic_bytes_per_update', 'lane_ops_per_update',
                      'kernel_time_scale')


def roofline_block(spec, program, schedule, updates, timing, dims, iterate,
                   step_us=None, clock=None):
  """`schedule` = [(kernel entry, modelled us)] as issued, `updates` = valid
  cell-updates of each iteration of the sweep that `timing` timed; `step_us` = wall
  time of one un-instrumented sweep (the timed loop).

  Kernel time: every launch at the fastest of three event-bracketed repeats; events
  between launches still add gaps, so when those launches sum to more than the
  un-instrumented step took, they are scaled down to it (`kernel_time_scale` < 1):
  a kernel is never reported longer than the step that contains it.

  The figures: `frac_algorithmic` is SURVEY.md 8(d)'s (algorithmic bytes per
  update x updates per launch / launch duration / 8 TB/s; > 1 means temporal
  blocking at work, it is not a bound); `hbm_measured_frac` the PMC bytes of the
  same kernel on the same grid / duration / 8 TB/s; `valu_frac` the useful f32
  lane-operations / duration / the chip's 78.6 T/s.  `bound` names the larger of
  the two physical fractions and `frac`, `achieved`, `peak`, `unit` belong to it."""
  from soda_hip.codegen import kernel as kernelmod
  from soda_hip.codegen import spec as specmod
  abytes = specmod.algorithmic_bytes_per_update(spec)
  name = timing['dominant_name']
  per_launch, done = [], 0
  for entry, _ in schedule:
    d = max(1, entry['depth'])
    if entry['name'] == name:
      per_launch.append(sum(updates[done:done + d]))
    done += d if entry['kind'] == 'fused' else 0
  if not per_launch:      # per-stage kernels: one iteration per group of launches
    per_launch = list(updates)
  upd = sum(per_launch) / len(per_launch)
  scale = 1.0
  if step_us and timing.get('fastest_us', 0) > step_us:
    scale = step_us / timing['fastest_us']
  avg_s = timing['dominant_us'] / max(1, timing['dominant_launches']) * 1e-6 * scale
  alg = upd * abytes / avg_s / 1e9
  ops = kernelmod.arithmetic_weight(spec)
  valu = upd * ops / avg_s / 1e12
  digest = ([kernelmod.calibration_key(e, spec) for e, _ in schedule if e['name'] == name]
            or [None])[0]
  traffic, source = measured_traffic(name, dims, iterate, digest,
                                     timing['dominant_launches'])
  block = dict(kernel=name, kernel_avg_us=avg_s * 1e6, kernel_time_scale=scale,
               kernel_launches=timing['dominant_launches'],
               updates_per_launch=upd, algorithmic_bytes_per_update=abytes,
               algorithmic_GBps=alg, frac_algorithmic=alg / HBM_PEAK_GBPS,
               lane_ops_per_update=ops, valu_Tlaneops=valu,
               valu_frac=valu / VALU_PEAK_TLANEOPS, traffic=None,
               hbm_measured_frac=None)
  hbm_frac = None
  if traffic:
    block.update(traffic=traffic['hbm_bytes_per_launch'], traffic_source=source,
                 traffic_commit=traffic.get('commit'),
                 traffic_read_bytes=traffic['read_bytes_per_launch'],
                 traffic_write_bytes=traffic['write_bytes_per_launch'],
                 hbm_measured_GBps=traffic['hbm_bytes_per_launch'] / avg_s / 1e9)
    hbm_frac = block['hbm_measured_GBps'] / HBM_PEAK_GBPS
    block['hbm_measured_frac'] = hbm_frac
  if hbm_frac is None:
    # no PMC figure for this shape: the kernel cannot move less than one read
    # and one write of the box per launch
    depth = max([e['depth'] for e, _ in schedule if e['name'] == name] or [1])
    hbm_frac = alg / max(1, depth) / HBM_PEAK_GBPS
    block['hbm_floor_frac'] = hbm_frac
  # Which limit binds: with SQ counters for this kernel on this grid, the busier of
  # the two units (VALU issue utilisation against the share of the ACHIEVABLE HBM
  # rate); without, the larger fraction of peak.
  sq, sq_source = measured_counters(name, dims, iterate, digest)
  valu_binds = block['valu_frac'] > hbm_frac
  if sq:
    block.update(valu_issue_utilisation=sq['valu_issue_utilisation'],
                 wave_parked=sq.get('wave_parked'), counters_source=sq_source)
    valu_binds = sq['valu_issue_utilisation'] > hbm_frac / HBM_ACHIEVABLE_FRAC
    # what separates valu_frac from 1: valu_frac = issue utilisation x useful share of
    # the issued instructions x shader clock / 2.4 GHz.  Wave-level VALU instructions
    # and shader cycles per launch are the counters' (GRBM_GUI_ACTIVE / 8 XCDs); a
    # packed instruction covers 128 lane-operations, a scalar one 64.
    packed = any(e.get('pairs') for e, _ in schedule if e['name'] == name)
    if sq.get('valu_instructions'):
      block['useful_instruction_share'] = upd * ops / (128.0 if packed else 64.0) / \
          sq['valu_instructions']
    if sq.get('shader_cycles'):
      block['shader_cycles_per_launch'] = sq['shader_cycles']
  # the clock this run held under the load (measured beside the sweeps, not a counter
  # file): valu_frac = VALU issue utilisation x useful share x shader_clock_ghz / 2.4
  if clock:
    block['shader_clock_ghz'] = clock['ghz']
    block['shader_clock_probe_ms'] = clock['seconds'] * 1e3
  if valu_binds:
    block.update(bound='valu', achieved=valu, peak=VALU_PEAK_TLANEOPS,
                 unit='Tlane-op/s', frac=block['valu_frac'])
  else:
    block.update(bound='hbm', achieved=hbm_frac * HBM_PEAK_GBPS, peak=HBM_PEAK_GBPS,
                 unit='GB/s', frac=hbm_frac)
  # the keys the verdict rests on come first: records that keep only the first N
  # scalar keys of an object (the driver's BENCH_rNN.json keeps 24) keep these
  return {k: block[k] for k in ROOFLINE_KEY_ORDER if k in block} | \
      {k: v for k, v in block.items() if k not in ROOFLINE_KEY_ORDER}


def parse_split(text, iterate):
  """'41x24+1x16' (config.depth_schedule of a bench line) -> [24] * 41 + [16]."""
  depths = []
  for part in text.split('+'):
    count, x, depth = part.strip().partition('x')
    if not x or not count.isdigit() or not depth.isdigit() or int(count) < 1 or int(depth) < 1:
      raise SystemExit('--split: %r is not <count>x<depth>[+<count>x<depth>...]' % part)
    depths += [int(depth)] * int(count)
  if sum(depths) != iterate:
    raise SystemExit('--split: the depths add up to %d, --iterate is %d' % (sum(depths),
                                                                            iterate))
  return depths


def schedule_text(schedule):
  """'41x24+1x16' from the launch list."""
  runs = []
  for entry, _ in schedule:
    label = str(entry['depth']) if entry['kind'] == 'fused' else entry['name']
    if runs and runs[-1][0] == label:
      runs[-1][1] += 1
    else:
      runs.append([label, 1])
  return '+'.join('%dx%s' % (n, label) for label, n in runs)


# The other single-GPU BASELINE configs ("plus cfg 2, 3, 5 single-point
# numbers"): timed after the headline on the same device in the same process, 30 steps
# after 10 of warm-up each (a burst of a few milliseconds runs at lower clocks), and
# printed under config.other_configs.
OTHER_CONFIGS = (
    ('cfg2', 'jacobi2d', [8192, 8192], 100),
    ('cfg3', 'blur', [16384, 16384], 1),
    ('cfg5', 'jacobi3d', [512, 512, 512], 200),
)
OTHER_STEPS, OTHER_WARMUP = 30, 10


def measure(app, dims, iterate, steps, warmup, max_depth=0, split='', no_tune=False,
            jit=False):
  """One workload under the bench protocol on the current device: `warmup` untimed
  sweeps (+ the tuning step), `steps` timed ones, then the same sweep under per-launch
  events for the roofline entry.  Returns (result line, spec)."""
  import torch
  from soda_hip.codegen import spec as specmod
  from soda_hip.runtime import host
  program, spec = open_program(app, iterate, jit)
  program.set_max_depth(max_depth)
  dims = list(dims)
  inputs = make_input(spec, dims)
  cells = int(np.prod(dims))
  din = [host.DeviceArray(a.nbytes) for a in inputs]
  dout = [host.DeviceArray(cells * dt.itemsize) for dt in program.out_dtypes]
  for d, a in zip(din, inputs):
    d.upload(a)
  for d in dout:
    d.zero()
  del inputs
  ip, op = [d.ptr for d in din], [d.ptr for d in dout]
  sync = torch.cuda.synchronize if torch.cuda.is_available() else \
      (lambda: host.capi.check(host.capi.lib().soda_hip_stream_synchronize(None)))
  if split:
    program.set_split(dims, iterate, parse_split(split, iterate))
    no_tune = True
  for _ in range(warmup):
    program.sweep(ip, op, dims, iterate)
  sync()
  if not no_tune:
    # untimed, like the warm-up: the candidate splits of `iterate` and the chunk lengths
    # of the memory-bound launches run as whole sweeps, the fastest on this device is kept
    program.tune(ip, op, dims, iterate)
    program.sweep(ip, op, dims, iterate)
    sync()
  t0 = time.perf_counter()
  for _ in range(steps):
    program.sweep(ip, op, dims, iterate)
  sync()
  wall = time.perf_counter() - t0
  # the same loop under hipEvents, per launch, for the roofline entry: three repeats,
  # every launch at its fastest
  timing = program.sweep_timed(ip, op, dims, iterate, warmup=0, repeats=3)
  schedule = program.schedule(dims, iterate)
  # the shader clock under this load, measured in-run: a probe wavefront sleeps beside
  # ~30 ms of the same sweeps and counts shader cycles against the constant 100 MHz clock
  # (not in the profiling passes, which repeat a given split: their kernel counts stay
  # sweeps x launches per sweep)
  clock = None
  if not split:
    probe_sweeps = max(1, int(0.03 / max(wall / steps, 1e-6)))
    clock = program.shader_clock_during(
        lambda: [program.sweep(ip, op, dims, iterate) for _ in range(probe_sweeps)],
        probe_sweeps * wall / steps)
    sync()
  valid = specmod.valid_cells(spec, dims, iterate)
  nominal = cells * iterate
  ms_per_step = wall / steps * 1e3
  value = valid / (wall / steps) / 1e9
  abytes = specmod.algorithmic_bytes_per_update(spec)
  result = dict(
      metric='gcell_updates_per_s', value=value, unit='Gcell-updates/s',
      n_gpus=1, steps=steps, warmup=warmup, ms_per_step=ms_per_step,
      higher_is_better=True, scaling='strong', vs_baseline=None,
      dtype='f32' if program.in_dtypes[0].kind == 'f' else 'u%d' % (
          8 * program.in_dtypes[0].itemsize),
      data='synthetic',
      config=dict(workload='%s.soda %s %s, iterate %d' % (
          app, program.in_dtypes[0].name, 'x'.join(map(str, dims)),
          iterate), app=app, dims=dims, iterate=iterate,
                  valid_cell_updates=valid, nominal_cell_updates=nominal,
                  nominal_gcell_updates_per_s=nominal / (wall / steps) / 1e9,
                  launches_per_step=timing['launches'],
                  depth_schedule=schedule_text(schedule),
                  depth_split='given (--split)' if split else
                  'measured (soda_hip_plan_tune)' if not no_tune and
                  iterate > 1 else 'calibrated model',
                  # chunk length and workgroups per CU of the memory-bound launches: the
                  # kernels' calibration records, or a pair the tuning step measured to
                  # beat them on this device
                  stream_chunk_choice='calibrated' if no_tune else
                  'measured (%d kernel x box)' % program.tuned_streams()
                  if program.tuned_streams() else 'calibrated (confirmed by measurement)',
                  effective_GBps=valid * abytes / (wall / steps) / 1e9,
                  device=host.device_info(0)['arch']),
      roofline=roofline_block(spec, program, schedule,
                              per_iteration_updates(spec, dims, iterate),
                              timing, dims, iterate, step_us=ms_per_step * 1e3,
                              clock=clock))
  for d in din + dout:
    d.free()
  program.close()
  return result, spec


OTHER_ROOFLINE_KEYS = ('kernel', 'bound', 'frac', 'unit', 'frac_algorithmic', 'traffic',
                       'hbm_measured_frac', 'hbm_floor_frac', 'valu_frac', 'kernel_avg_us',
                       'kernel_launches', 'shader_clock_ghz', 'useful_instruction_share')


def other_config_entry(tag, line):
  """The condensed form of a bench line under config.other_configs."""
  rf = line['roofline']
  return dict(config=tag, workload=line['config']['workload'], ms=line['ms_per_step'],
              gcell_updates_per_s=line['value'], steps=line['steps'],
              warmup=line['warmup'], launches=line['config']['launches_per_step'],
              depth_schedule=line['config']['depth_schedule'],
              effective_GBps=line['config']['effective_GBps'],
              roofline={k: rf.get(k) for k in OTHER_ROOFLINE_KEYS})


def add_other_configs(config, others):
  """config.other_configs plus scalar copies of what the verdict reads (cfgN_ms, _bound,
  _frac, _traffic) in front of it: records that drop list values (the driver's
  BENCH_rNN.json) keep the scalars."""
  for entry in others:
    tag = entry['config']
    rf = entry.get('roofline') or {}
    config[tag + '_ms'] = entry.get('ms')
    config[tag + '_bound'] = rf.get('bound')
    config[tag + '_frac'] = rf.get('frac')
    config[tag + '_traffic'] = rf.get('traffic')
  config['other_configs'] = others
  return config


def is_headline(args):
  return args.app == 'jacobi2d' and list(args.size) == [16384, 16384] and \
      args.iterate == 1000


def run_single(args):
  result, spec = measure(args.app, args.size, args.iterate, args.steps, args.warmup,
                         max_depth=args.max_depth, split=args.split,
                         no_tune=args.no_tune, jit=args.jit)
  if is_headline(args) and not args.no_other_configs and not args.split:
    others = []
    for tag, app, dims, iterate in OTHER_CONFIGS:
      try:
        line, _ = measure(app, dims, iterate, OTHER_STEPS, OTHER_WARMUP, jit=args.jit)
        others.append(other_config_entry(tag, line))
      except Exception as e:   # noqa: BLE001 - a secondary config must not cost the headline
        others.append(dict(config=tag, error='%s: %s' % (type(e).__name__, str(e)[:300])))
    add_other_configs(result['config'], others)
  if args.cpu_seconds > 0:
    result['cpu_baseline'] = cpu_baseline(spec, list(args.size), args.cpu_seconds)
  return result


def main():
  args = parse_args()
  # RCCL between processes needs dmabuf IPC on this driver stack (exported by the
  # image already; kept for environments built by hand)
  os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
  world = int(os.environ.get('WORLD_SIZE', '1'))
  if args.gpus > 1 and 'RANK' not in os.environ:
    # started by hand without a launcher: start one rank per GPU as CHILD
    # processes (nothing here has touched the GPU yet) and pass their status on
    import subprocess
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
           '--nproc-per-node', str(args.gpus), '--master-addr', '127.0.0.1',
           '--master-port', os.environ.get('MASTER_PORT', '29533'),
           os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd))
  if args.force_dist and 'RANK' not in os.environ:      # a one-rank group by hand
    os.environ.update(RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29534')
  if args.gpus > 1 or world > 1 or args.force_dist:
    from soda_hip.runtime import dist
    result = dist.bench_main(args, open_program, make_input, per_iteration_updates,
                             roofline_block, schedule_text, cpu_baseline)
  else:
    result = run_single(args)
  if result is not None:
    print(json.dumps(result))


if __name__ == '__main__':
  main()
