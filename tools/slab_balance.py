#!/usr/bin/env python3
"""What the busiest rank of a multi-GPU run costs under the static cut and under slabs
re-cut every super-step, measured on ONE GPU: every rank's sequence of super-step sweeps
(its sub-arrays, its iteration counts, no exchange - launch times do not depend on the
data) timed by itself, one rank after the other.  A step of the real run costs what the
slowest rank costs (plus the exchanges, which one GPU cannot measure).
usage: slab_balance.py [app] [world] [E] [iterate] [dims...]   (defaults: cfg4 on 8 ranks)
"""
import argparse
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'soda-compiler_amd')]

ap = argparse.ArgumentParser(description=__doc__.split('\n\n')[0])
ap.add_argument('app', nargs='?', default='jacobi2d')
ap.add_argument('world', nargs='?', type=int, default=8)
ap.add_argument('exchange', nargs='?', type=int, default=0, help='0 = dist.auto_exchange')
ap.add_argument('iterate', nargs='?', type=int, default=1000)
ap.add_argument('dims', nargs='*', type=int, default=[16384, 16384])
ap.add_argument('--repeats', type=int, default=3)
args = ap.parse_args()

import torch  # noqa: E402  (one HIP runtime per process: torch's first)
from soda_hip import frontend  # noqa: E402
from soda_hip.codegen import spec as specmod  # noqa: E402
from soda_hip.runtime import dist as sdist, host  # noqa: E402

st = frontend.load(os.path.join(ROOT, 'tests', 'samples', args.app + '.soda'),
                   iterate=args.iterate)
spec = specmod.spec_from_stencil(st)
prog = host.open_program(blob=os.path.join(ROOT, 'soda-compiler_amd', 'blobs',
                                           args.app + '.hsaco'), spec=spec)
engine = sdist.HipEngine(prog, torch)
dims, world, iterate = list(args.dims), args.world, args.iterate
r_lo, r_hi = spec['radius']['lo'][-1], spec['radius']['hi'][-1]
deepest = max(k['depth'] for k in prog.kernels if k['kind'] == 'fused')
E = args.exchange or sdist.auto_exchange(dims[-1] // world, max(r_lo, r_hi, 1), deepest, iterate)
table = specmod.iteration_margins(spec, iterate)
dev = torch.device('cuda', 0)


def margins_of(k):
  return ((0,) * len(dims), (0,) * len(dims)) if k == 0 else table[k - 1]


def time_rank(plan):
  shape = tuple(reversed(plan.local_dims))
  a = torch.rand(shape, dtype=torch.float32, device=dev)
  b = torch.zeros_like(a)
  shapes = sdist.super_step_shapes(plan, iterate)
  recut = isinstance(plan, sdist.RecutPlan)

  def run():
    done = 0
    for first, rows, step in shapes:
      lo, hi = (list(v) for v in margins_of(done))
      if recut:
        lo[-1] = hi[-1] = 0
      else:
        lo, hi = plan.valid_margins(done, margins_of)
      engine.sweep(a, b, plan.local_dims, step, lo, hi, rows=(first, first + rows))
      done += step
  run()
  best = None
  for _ in range(args.repeats):
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    run()
    t1.record()
    torch.cuda.synchronize()
    ms = t0.elapsed_time(t1)
    best = ms if best is None else min(best, ms)
  return best, sum(rows * step for _, rows, step in shapes)


print('# %s %s x%d on %d ranks, exchange every %d iterations; one GPU, rank by rank, '
      'compute only (fastest of %d)' % (args.app, 'x'.join(map(str, dims)), iterate, world, E,
                                        args.repeats))
one = sdist.make_plan(True, dims, 0, 1, r_lo, r_hi, iterate, iterate)
whole, _ = time_rank(one)
print('one rank, whole grid: %.3f ms -> an even share is %.3f ms' % (whole, whole / world))
for static in (True, False):
  times = []
  for rank in range(world):
    plan = sdist.make_plan(static, dims, rank, world, r_lo, r_hi, E, iterate)
    ms, rows = time_rank(plan)
    times.append(ms)
  label = 'static cut' if static else 're-cut    '
  print('%s: busiest rank %.3f ms (%.1f %% of N x even share = scaling ceiling before '
        'exchanges)  ranks: %s' % (label, max(times), 100.0 * whole / world / max(times),
                                   ' '.join('%.3f' % t for t in times)), flush=True)
prog.close()
