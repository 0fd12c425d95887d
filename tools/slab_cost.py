#!/usr/bin/env python3
"""What one rank's super-step costs at the slab sizes of 1/2/4/8 ranks, on ONE
GPU: own rows + 2 E ghost rows, E iterations, for several exchange periods E and
caps on the fused depth.  Feeds runtime/dist.py: auto_exchange and DESIGN.md 6.
usage: slab_cost.py [app] [W] [H] ['E,max_depth' ...]
"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'soda-compiler_amd')]
import numpy as np  # noqa: E402
import torch  # noqa: E402,F401  (one HIP runtime per process: torch's first)
from soda_hip import frontend  # noqa: E402
from soda_hip.codegen import spec as specmod  # noqa: E402
from soda_hip.runtime import host  # noqa: E402

app = sys.argv[1] if len(sys.argv) > 1 else 'jacobi2d'
W = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
H = int(sys.argv[3]) if len(sys.argv) > 3 else 16384
variants = [tuple(int(x) for x in v.split(',')) for v in sys.argv[4:]] or [
    (e, d) for d in (16, 12, 8) for e in (48, 96, 144, 192)]
st = frontend.load(os.path.join(ROOT, 'tests', 'samples', app + '.soda'), iterate=1000)
spec = specmod.spec_from_stencil(st)
blob = os.path.join(ROOT, 'soda-compiler_amd', 'blobs', app + '.hsaco')
prog = host.open_program(blob=blob, spec=spec)
r = max(spec['radius']['lo'][-1], spec['radius']['hi'][-1], 1)
rng = np.random.default_rng(1)
for world in [int(v) for v in os.environ.get('SLAB_WORLDS', '1,2,4,8').split(',')]:
  own = H // world
  for e, depth in variants:
    ghosts = 0 if world == 1 else 2 * e * r
    rows = own + ghosts
    a = rng.random((rows, W), dtype=np.float32)
    din = host.DeviceArray(a.nbytes)
    din.upload(a)
    dout = host.DeviceArray(a.nbytes)
    dout.zero()
    prog.set_max_depth(depth)
    # ~40 ms of the same work first: short bursts run at lower clocks
    reps = max(3, int(40000.0 / (e * 45.0 * rows / 16384)))
    t = prog.sweep_timed([din.ptr], [dout.ptr], [W, rows], e, warmup=reps, repeats=reps)
    print('ranks %d own %5d rows +%4d ghosts  E %3d  depth<=%2d : %8.1f us/super-step '
          '%6.2f us/iteration' % (world, own, ghosts, e, depth, t['kernel_us'],
                                  t['kernel_us'] / e), flush=True)
    din.free()
    dout.free()
prog.close()
