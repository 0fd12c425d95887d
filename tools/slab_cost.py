#!/usr/bin/env python3
"""What one rank's super-step costs at the slab sizes of 1/2/4/8 ranks, on ONE
GPU: own rows + 2 E ghost rows, E iterations, for several exchange periods E and
caps on the fused depth.  Feeds runtime/dist.py: auto_exchange and DESIGN.md 7.
With `split` the super-step is cut as the overlapping schedule cuts it (two
boundary bands first, then the interior: dist.band_plan), which prices what
hiding the exchange costs in compute.
usage: slab_cost.py [app] [W] [H] ['E,max_depth[,split]' ...]
"""
import sys as _sys
if len(_sys.argv) > 1 and _sys.argv[1] in ('-h', '--help'):   # usage = the text above
  print(__doc__)
  _sys.exit(0)
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'soda-compiler_amd')]
import numpy as np  # noqa: E402
import torch  # noqa: E402,F401  (one HIP runtime per process: torch's first)
from soda_hip import frontend  # noqa: E402
from soda_hip.codegen import spec as specmod  # noqa: E402
from soda_hip.runtime import dist as sdist, host  # noqa: E402

app = sys.argv[1] if len(sys.argv) > 1 else 'jacobi2d'
W = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
H = int(sys.argv[3]) if len(sys.argv) > 3 else 16384
variants = [tuple(v.split(',')) for v in sys.argv[4:]] or [
    (str(e), str(d)) for d in (0, 16, 12) for e in (48, 96, 144, 192)]
st = frontend.load(os.path.join(ROOT, 'tests', 'samples', app + '.soda'), iterate=1000)
spec = specmod.spec_from_stencil(st)
blob = os.path.join(ROOT, 'soda-compiler_amd', 'blobs', app + '.hsaco')
prog = host.open_program(blob=blob, spec=spec)
engine = sdist.HipEngine(prog, torch)
r = max(spec['radius']['lo'][-1], spec['radius']['hi'][-1], 1)
dev = torch.device('cuda', 0)
base = None
for world in [int(v) for v in os.environ.get('SLAB_WORLDS', '1,2,4,8').split(',')]:
  for variant in variants:
    e, depth = int(variant[0]), int(variant[1])
    split = len(variant) > 2 and variant[2] == 'split' and world > 2
    # a rank in the middle of the chain: neighbours on both sides
    plan = sdist.SlabPlan([W, H], min(1, world - 1), world, r, r, e)
    if world == 2:
      rows = plan.own + 2 * e * r           # as round 1: ghosts on both sides
    else:
      rows = plan.local_extent
    a = torch.rand((rows, W), dtype=torch.float32, device=dev)
    b = torch.zeros_like(a)
    prog.set_max_depth(depth)
    zero = [0, 0]

    def super_step():
      if split:
        bands, interior = sdist.band_plan(plan, e)
        for r0, r1, _, _ in bands + [interior]:
          engine.sweep(a, b, [W, rows], e, zero, zero, rows=(r0, r1))
      else:
        engine.sweep(a, b, [W, rows], e, zero, zero)
    # ~40 ms of the same work first: short bursts run at lower clocks
    reps = max(3, int(40000.0 / (e * 40.0 * rows / 16384)))
    for _ in range(reps):
      super_step()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(reps):
      super_step()
    t1.record()
    torch.cuda.synchronize()
    us = t0.elapsed_time(t1) * 1e3 / reps
    per_row = us / e / rows * 1e3
    if world == 1 and base is None:
      base = per_row
    print('ranks %d own %5d rows +%4d ghosts  E %3d  depth<=%2d %-5s: %8.1f us/super-step '
          '%6.2f us/iteration  %.3f ns/row/iteration (%.0f %% of the 1-rank rate)  [%s]' % (
              world, plan.own, rows - plan.own, e, depth, 'split' if split else '',
              us, us / e, per_row, 100.0 * (base or per_row) / per_row,
              '+'.join(str(k['depth']) for k, _ in prog.schedule([W, rows], e))),
          flush=True)
    del a, b
prog.close()
