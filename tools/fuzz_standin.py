#!/usr/bin/env python3
"""Fuzzing run on the GPU box: the C slab driver (soda_hip_run_slab) over the test-only RCCL
stand-in with random programs, grids, rank counts (2-4), exchange periods, iteration counts
and BOTH orders (serial, bands first) - own rows put together against the oracle, message
and byte counts against the schedule (tests/test_gpu_parity.py: run_slab_over_the_standin).
usage: fuzz_standin.py first_seed count"""
import sys as _sys
if len(_sys.argv) > 1 and _sys.argv[1] in ('-h', '--help'):   # usage = the text above
  print(__doc__)
  _sys.exit(0)
import os
import sys
import tempfile
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'soda-compiler_amd'), os.path.join(ROOT, 'tests')]
import numpy as np
import test_gpu_parity as T

first, count = int(sys.argv[1]), int(sys.argv[2])
failures, t0 = 0, time.time()
with tempfile.TemporaryDirectory() as top:
  standin = T.build_rccl_standin(top)
  for seed in range(first, first + count):
    rng = np.random.default_rng(770000 + seed)
    app = str(rng.choice(['jacobi2d', 'jacobi2d', 'skew2d', 'jacobi3d']))
    spec = T.gpu_util.load_spec(app)
    world = int(rng.integers(2, 5))
    r = max(spec['radius']['lo'][-1], spec['radius']['hi'][-1], 1)
    if spec['dim'] == 2:
      iterate = int(rng.integers(1, 120))
      dims = (int(rng.integers(600, 1500)), int(rng.integers(world * 40, 2200)))
    else:
      iterate = int(rng.integers(1, 24))
      dims = (int(rng.integers(128, 180)), int(rng.integers(64, 100)),
              int(rng.integers(world * 12, 160)))
    # the valid box must not be empty
    if any(n - 2 * iterate * r <= 0 for n in dims):
      iterate = max(1, min(dims) // (2 * r) - 1)
    exchange = int(rng.integers(1, iterate + 3))
    order = int(rng.integers(0, 2))
    cut = 'recut' if rng.random() < 0.6 else 'static'     # round 6: slabs re-cut every super-step
    case = os.path.join(top, 'c%d' % seed)
    os.makedirs(case)
    try:
      T.run_slab_over_the_standin(case, standin, app, dims, world, iterate, exchange, order,
                                  cut=cut)
    except BaseException as e:   # noqa: BLE001 - counted and reported
      failures += 1
      print('FAIL seed %d: %s %s world %d iterate %d exchange %d order %d cut %s: %s' % (
          seed, app, dims, world, iterate, exchange, order, cut, str(e)[:300]), flush=True)
    if (seed - first + 1) % 20 == 0:
      print('%d cases, %d failures, %.0f s' % (seed - first + 1, failures, time.time() - t0),
            flush=True)
print('done: %d cases, %d failures' % (count, failures))
sys.exit(1 if failures else 0)
