"""Worker of tests/test_dist.py: one rank of a gloo process group running the
slab decomposition of soda_hip.runtime.dist with a CPU engine built on the
oracle (test infrastructure; the product engine is the HIP program)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'soda-compiler_amd')):
  if p not in sys.path:
    sys.path.insert(0, p)

from soda_hip import frontend                      # noqa: E402
from soda_hip.codegen import spec as specmod       # noqa: E402
from soda_hip.runtime import dist as sdist         # noqa: E402
from oracle import soda_oracle                     # noqa: E402


class OracleEngine:
  """sweep() with the same contract as libsoda_hip's soda_hip_sweep."""

  def __init__(self, spec, launch_depth=0):
    """launch_depth = D > 0: behave like a sweep split into launches of D
    iterations whose destinations alternate as libsoda_hip's do (soda_hip.cpp:
    build_schedule): unless final_only, every second launch counted from the end
    writes ITS level's (larger) box into dst too."""
    self.spec = spec
    self.oracle = soda_oracle.Oracle(spec)
    self.launch_depth = launch_depth

  def sweep(self, src, dst, local_dims, iterations, valid_lo, valid_hi, rows=None,
            final_only=False):
    spec = self.spec
    name_in = spec['inputs'][0]['name']
    name_out = spec['outputs'][0]
    if rows is not None:    # the sub-array of those rows, as HipEngine does it
      local_dims = list(local_dims[:-1]) + [rows[1] - rows[0]]
      src, dst = src[rows[0]:rows[1]], dst[rows[0]:rows[1]]
    cur = src.numpy()
    boxes = soda_oracle.iteration_boxes(spec, iterations)
    locals_ = {s['name']: np.zeros_like(cur, dtype=self.oracle.dtype(s['name']))
               for s in spec['stages'] if s['name'] != name_out}
    for k in range(iterations):
      out = np.zeros_like(cur)
      arrays = dict(locals_)
      arrays[name_in] = cur
      arrays[name_out] = out
      shifted = {n: ([a - b for a, b in zip(lo, valid_lo)],
                     [a + b for a, b in zip(hi, valid_hi)])
                 for n, (lo, hi) in boxes[k].items()}
      self.oracle._call(arrays, tuple(local_dims), shifted)
      cur = out
      d = self.launch_depth
      if d and (k + 1) % d == 0 and k + 1 < iterations and not final_only:
        m = -(-iterations // d)             # launches of this sweep
        i = (k + 1) // d - 1                # the launch that just ended
        if (m - 1 - i) % 2 == 0:            # its destination is `out`
          lo, hi = boxes[k][name_out]
          box = tuple(slice(vl - l, n - vh - h) for l, h, vl, vh, n in reversed(list(
              zip(lo, hi, valid_lo, valid_hi, local_dims))))
          dst[box] = torch.from_numpy(cur)[box]
    # like the kernels: only the valid box is written
    lo, hi = boxes[iterations - 1][name_out]
    box = tuple(slice(vl - l, n - vh - h) for l, h, vl, vh, n in reversed(list(zip(
        lo, hi, valid_lo, valid_hi, local_dims))))
    dst[box] = torch.from_numpy(cur)[box]


class BandsFirst(sdist.SerialSchedule):
  """The overlapping schedule's ORDER without streams: bands, exchange of the
  next super-step, interior - what StreamSchedule runs concurrently on a GPU."""
  overlapped = True


def main():
  app, size, iterate, exchange, out_dir = sys.argv[1:6]
  mode = sys.argv[6] if len(sys.argv) > 6 else ''
  # 'recut' / 'recut+overlap[:D]': slabs cut afresh every super-step (RecutPlan)
  recut = mode.startswith('recut')
  if recut:
    mode = mode[len('recut'):].lstrip('+')
  overlapped = mode.startswith('overlap')
  # 'overlap:D': the CPU engine emulates launches of D iterations
  launch_depth = int(mode.split(':')[1]) if overlapped and ':' in mode else 0
  dims = [int(v) for v in size.split('x')]
  iterate, exchange = int(iterate), int(exchange)
  rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
  dist.init_process_group(backend='gloo')
  st = frontend.load(os.path.join(ROOT, 'tests', 'samples', app + '.soda'),
                     iterate=iterate)
  spec = specmod.spec_from_stencil(st)
  r_lo, r_hi = spec['radius']['lo'][-1], spec['radius']['hi'][-1]
  if recut:
    plan = sdist.RecutPlan(dims, rank, world, r_lo, r_hi, exchange, iterate)
  else:
    plan = sdist.SlabPlan(dims, rank, world, r_lo, r_hi, exchange)
  dt = np.dtype(specmod.NUMPY_NAME[spec['inputs'][0]['c_type']])
  rng = np.random.default_rng(99)
  if dt.kind == 'f':
    full = rng.random(tuple(reversed(dims)), dtype=np.float32).astype(dt)
  else:
    full = rng.integers(0, 65536, size=tuple(reversed(dims))).astype(dt)
  shape = tuple(reversed(plan.local_dims))
  # rows nobody filled are NaN / all-ones: a sweep that read one would show
  poison = np.nan if dt.kind == 'f' else np.iinfo(dt).max
  a = torch.full(shape, poison, dtype=torch.from_numpy(full[:1]).dtype)
  a[plan.ghost_lo:plan.ghost_lo + plan.own] = torch.from_numpy(
      full[plan.start:plan.stop])
  b, c = torch.full_like(a, poison), torch.full_like(a, poison)
  table = specmod.iteration_margins(spec, iterate)

  def margins_of(k):
    return ((0,) * len(dims), (0,) * len(dims)) if k == 0 else table[k - 1]

  order = BandsFirst() if overlapped else None
  engine = OracleEngine(spec, launch_depth)
  if recut:
    def run(ready):
      return sdist.run_recut(engine, plan, [a, b, c], margins_of, dist,
                             ghosts_ready=ready, schedule=order)
    first, last = plan.local(plan.final_rows)
    start, stop = plan.final_rows
  else:
    def run(ready):
      return sdist.run_slab(engine, plan, [a, b, c], iterate, margins_of, dist,
                            ghosts_ready=ready, schedule=order)
    first, last = plan.ghost_lo, plan.ghost_lo + plan.own
    start, stop = plan.start, plan.stop
  result, exchanges = run(False)
  own = result[first:last].numpy().copy()
  # A was not written and now carries the neighbours' level-0 rows: a second
  # sweep may skip its first exchange and must give the same rows
  again, fewer = run(True)
  assert fewer == exchanges - 1, (fewer, exchanges)
  assert np.array_equal(again[first:last].numpy(), own, equal_nan=dt.kind == 'f')
  np.save(os.path.join(out_dir, 'rank%d.npy' % rank), own)
  with open(os.path.join(out_dir, 'rank%d.txt' % rank), 'w') as f:
    f.write('%d %d %d %d\n' % (start, stop, plan.exchange, exchanges))
  dist.barrier()
  dist.destroy_process_group()


if __name__ == '__main__':
  main()
