#!/usr/bin/env python3
"""Headline benchmark: jacobi2d float 16384 x 16384, iterate 1000 (BASELINE.json
configs[3], the configuration the metric is quoted on; it fits one MI355X).

  python bench.py --gpus N --steps K --warmup W

A "step" is one whole 1000-iteration sweep of the grid, input already resident
in HBM.  N > 1 is launched by torch.distributed.run, one rank per GPU: the grid
is cut into N slabs along the outer dimension (strong scaling: the grid is fixed)
and neighbours exchange ghost rows over RCCL every `--exchange` iterations.

Prints ONE JSON line (rank 0).  `value` = valid cell-updates / wall time, valid =
the cells the reference semantics define (the box shrinks by the stencil radius
every iteration); `roofline` prices the dominant kernel at
8 algorithmic bytes per cell-update against the 8 TB/s HBM peak -- with temporal
blocking the fraction may exceed 1, which is the point of time-tiling;
`cpu_baseline` is the CPU oracle (a port of the loop nest the reference emits
into `<app>_test`) timed on this host's cores on a bounded sample.
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
                  help='iterations between halo exchanges (0 = auto)')
  ap.add_argument('--cpu-seconds', type=float, default=12.0,
                  help='CPU baseline sample budget (0 = skip)')
  ap.add_argument('--force-dist', action='store_true',
                  help='take the torch.distributed slab path even with 1 rank')
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
    if dt.kind == 'f':
      a = rng.random(shape, dtype=np.float32).astype(dt, copy=False)
    else:
      a = rng.integers(0, np.iinfo(dt).max + 1, size=shape).astype(dt)
    if rows is not None:
      a = np.ascontiguousarray(a[rows[0]:rows[1]])
    out.append(a)
  return out


def cpu_baseline(spec, dims, budget_s):
  """The CPU oracle on this host: a bounded number of whole-grid sweeps."""
  from oracle import soda_oracle
  orc = soda_oracle.Oracle(spec, flags=('-O3', '-march=native'))
  inputs = make_input(spec, dims)
  # The box may show far more logical CPUs than its cgroup grants (256 vs a
  # quota of 16 on the round-1 boxes; 256 threads then run 30x slower than 32).
  # Try team sizes around the quota, keep the fastest.
  quota = orc.cpu_quota()
  logical = len(os.sched_getaffinity(0))
  best = None
  for threads in sorted({max(1, quota // 2), quota, min(logical, quota * 2),
                         min(logical, quota * 4)}):
    orc.set_threads(threads)
    for _ in range(2):      # shared hosts are noisy: best of two short trials
      t, u = orc.time_iterations(inputs, 3, warmup=1)
      if best is None or u / t > best[1]:
        best = (threads, u / t, t / 3)
  threads, _, per = best
  orc.set_threads(threads)
  n = int(max(8, min(400, budget_s / max(per, 1e-6))))
  t, u = orc.time_iterations(inputs, n, warmup=2)
  return dict(value=u / t / 1e9, unit='Gcell-updates/s',
              cores=threads, kind='port',
              sample='%d full-grid sweeps of %s (iterations 3..%d of the run), '
                     'OpenMP team of %d (cgroup quota %d of %d logical CPUs; best '
                     'of the team sizes tried), g++ -O3 -march=native '
                     '-ffp-contract=off; %.2f s' % (
                         n, 'x'.join(map(str, dims)), n + 2, threads, quota,
                         logical, t))


def measured_traffic(kernel):
  """HBM bytes per launch of `kernel` from the newest committed rocprofv3 PMC
  pass of this same command (profiles/rNN_traffic.json; tools/collect_profiles.py
  explains the gfx950 correction).  PMC counters cannot be collected from inside
  the timed process, so the figure comes from that separate run."""
  import glob
  files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_traffic.json')))
  for path in reversed(files):
    with open(path) as f:
      data = json.load(f)
    if kernel in data:
      return data[kernel]['hbm_bytes_per_launch'], os.path.basename(path)
  return None, None


def launch_updates(spec, dims, iterate, depths):
  """Valid cell-updates done by each launch of a depth schedule."""
  from soda_hip.codegen import spec as specmod
  margins = specmod.iteration_margins(spec, iterate)
  per_iter = []
  for lo, hi in margins:
    cells = 1
    for d in range(spec['dim']):
      cells *= max(0, dims[d] - lo[d] - hi[d])
    per_iter.append(cells)
  out, done = [], 0
  for k in depths:
    out.append(sum(per_iter[done:done + k]))
    done += k
  return out


def depth_schedule(program, iterate, max_depth):
  depths = sorted({k['depth'] for k in program.kernels if k['kind'] == 'fused'
                   and (max_depth <= 0 or k['depth'] <= max_depth)}, reverse=True)
  if not depths or max_depth < 0:   # per-stage kernels: one iteration per pass
    return [1] * iterate
  seq, left = [], iterate
  while left > 0:
    k = next(d for d in depths if d <= left)
    seq.append(k)
    left -= k
  return seq


def run_single(args):
  import torch
  from soda_hip.codegen import spec as specmod
  from soda_hip.runtime import host
  program, spec = open_program(args.app, args.iterate, args.jit)
  program.set_max_depth(args.max_depth)
  dims = list(args.size)
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
  for _ in range(args.warmup):
    program.sweep(ip, op, dims, args.iterate)
  sync()
  t0 = time.perf_counter()
  for _ in range(args.steps):
    program.sweep(ip, op, dims, args.iterate)
  sync()
  wall = time.perf_counter() - t0
  # the same loop once more under hipEvents, per launch, for the roofline entry
  timing = program.sweep_timed(ip, op, dims, args.iterate, warmup=0, repeats=1)
  valid = specmod.valid_cells(spec, dims, args.iterate)
  nominal = cells * args.iterate
  ms_per_step = wall / args.steps * 1e3
  value = valid / (wall / args.steps) / 1e9
  abytes = specmod.algorithmic_bytes_per_update(spec)
  seq = depth_schedule(program, args.iterate, args.max_depth)
  per_launch = launch_updates(spec, dims, args.iterate, seq)
  dom_depth = max(seq, key=lambda k: sum(u for d, u in zip(seq, per_launch) if d == k))
  dom_updates = [u for d, u in zip(seq, per_launch) if d == dom_depth]
  dom_avg_s = timing['dominant_us'] / max(1, timing['dominant_launches']) * 1e-6
  achieved = (sum(dom_updates) / len(dom_updates)) * abytes / dom_avg_s / 1e9
  result = dict(
      metric='gcell_updates_per_s', value=value, unit='Gcell-updates/s',
      n_gpus=1, steps=args.steps, warmup=args.warmup, ms_per_step=ms_per_step,
      higher_is_better=True, scaling='strong', vs_baseline=None,
      dtype='f32' if program.in_dtypes[0].kind == 'f' else 'u%d' % (
          8 * program.in_dtypes[0].itemsize),
      data='synthetic',
      config=dict(workload='%s.soda %s %s, iterate %d' % (
          args.app, program.in_dtypes[0].name, 'x'.join(map(str, dims)),
          args.iterate), app=args.app, dims=dims, iterate=args.iterate,
                  valid_cell_updates=valid, nominal_cell_updates=nominal,
                  nominal_gcell_updates_per_s=nominal / (wall / args.steps) / 1e9,
                  launches_per_step=timing['launches'], depth_schedule=
                  '%dx%d' % (seq.count(dom_depth), dom_depth) + ''.join(
                      '+%d' % k for k in seq if k != dom_depth),
                  effective_GBps=valid * abytes / (wall / args.steps) / 1e9,
                  device=host.device_info(0)['arch']),
      roofline=dict(bound='hbm', achieved=achieved, peak=HBM_PEAK_GBPS,
                    unit='GB/s', frac=achieved / HBM_PEAK_GBPS,
                    traffic=measured_traffic(timing['dominant_name'])[0],
                    traffic_source=measured_traffic(timing['dominant_name'])[1],
                    kernel=timing['dominant_name'],
                    kernel_avg_us=dom_avg_s * 1e6,
                    kernel_launches=timing['dominant_launches'],
                    algorithmic_bytes_per_update=abytes,
                    updates_per_launch=sum(dom_updates) / len(dom_updates)))
  # physical HBM rate of the dominant kernel: PMC bytes per launch / its duration
  traffic = result['roofline']['traffic']
  if traffic:
    result['roofline']['hbm_measured_GBps'] = traffic / dom_avg_s / 1e9
    result['roofline']['hbm_measured_frac'] = traffic / dom_avg_s / 1e9 / HBM_PEAK_GBPS
  for d in din + dout:
    d.free()
  program.close()
  if args.cpu_seconds > 0:
    result['cpu_baseline'] = cpu_baseline(spec, dims, args.cpu_seconds)
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
  if args.gpus > 1 or world > 1 or args.force_dist:
    from soda_hip.runtime import dist
    result = dist.bench_main(args, open_program, make_input, cpu_baseline,
                             launch_updates, depth_schedule, HBM_PEAK_GBPS)
  else:
    result = run_single(args)
  if result is not None:
    print(json.dumps(result))


if __name__ == '__main__':
  main()
