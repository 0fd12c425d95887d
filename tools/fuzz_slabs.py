#!/usr/bin/env python3
"""Fuzzing run on the GPU box: the emulated slab decomposition of the GPU tests
(tests/test_gpu_parity.py: test_slab_decomposition_with_the_hip_engine) on random
grids, rank counts, exchange periods and iteration counts.
usage: fuzz_slabs.py first_seed count"""
import sys as _sys
if len(_sys.argv) > 1 and _sys.argv[1] in ('-h', '--help'):   # usage = the text above
  print(__doc__)
  _sys.exit(0)
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'soda-compiler_amd'), os.path.join(ROOT, 'tests')]
import numpy as np
import test_gpu_parity as T
from soda_hip.codegen import spec as specmod

first, count = int(sys.argv[1]), int(sys.argv[2])
failures = 0
done = 0
t0 = time.time()
for seed in range(first, first + count):
  rng = np.random.default_rng(990000 + seed)
  app = str(rng.choice(['jacobi2d', 'jacobi2d', 'skew2d', 'seidel2d', 'jacobi3d', 'heat3d']))
  spec = T.program(app).spec
  dim = spec['dim']
  world = int(rng.integers(2, 9))
  iterate = int(rng.integers(1, 60 if dim == 2 else 16))
  exchange = int(rng.integers(1, iterate + 4))
  r = max(spec['radius']['lo'][-1], spec['radius']['hi'][-1])
  if dim == 2:
    dims = (int(rng.integers(100, 1300)), int(rng.integers(world * 8, 900)))
  else:
    dims = (int(rng.integers(20, 140)), int(rng.integers(20, 90)),
            int(rng.integers(world * 4, 130)))
  # keep something valid after `iterate` iterations, and slabs thicker than the reach
  lo, hi = specmod.iteration_margins(spec, iterate)[-1]
  if any(dims[d] - lo[d] - hi[d] < 2 for d in range(dim)) or dims[-1] // world < r:
    continue
  # round 6: half the cases with slabs re-cut every super-step (up to 8 emulated ranks,
  # whole sweeps or bands first)
  recut = rng.random() < 0.5
  bands = bool(rng.integers(0, 2))
  try:
    if recut:
      T.test_recut_decomposition_with_the_hip_engine(app, dims, world, exchange, iterate, bands)
    else:
      T.test_slab_decomposition_with_the_hip_engine(app, dims, world, exchange, iterate)
  except Exception as e:
    failures += 1
    print('FAIL', app, dims, world, exchange, iterate, 'recut' if recut else 'static', bands,
          repr(e)[:200], flush=True)
  done += 1
  if done % 25 == 0:
    print('%d cases, %d failures, %.0f s' % (done, failures, time.time() - t0), flush=True)
print('done: %d cases, %d failures' % (done, failures))
