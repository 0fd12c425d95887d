#!/usr/bin/env python3
"""Fuzzing run on the GPU box: the prebuilt sample programs on random array shapes
(tiny, ragged, around the kernels' smallest-array limits) and iteration counts,
against the CPU oracle.  usage: fuzz_shapes.py first_seed count"""
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
import gpu_util
from soda_hip.codegen import spec as specmod

first, count = int(sys.argv[1]), int(sys.argv[2])
apps = ['jacobi2d', 'blur', 'seidel2d', 'sobel2d', 'denoise2d', 'skew2d', 'jacobi3d',
        'heat3d', 'denoise3d']
progs, oracles = {}, {}
failures = 0
t0 = time.time()
for seed in range(first, first + count):
  rng = np.random.default_rng(880000 + seed)
  app = apps[int(rng.integers(0, len(apps)))]
  if app not in progs:
    progs[app] = gpu_util.open_prebuilt(app)
    oracles[app] = gpu_util.make_oracle(progs[app].spec)
  prog, orc = progs[app], oracles[app]
  spec = prog.spec
  dim = spec['dim']
  chain = len(spec['inputs']) == len(spec['outputs'])
  iterate = int(rng.integers(1, 41)) if chain else 1
  if dim == 2:
    mode = rng.integers(0, 3)
    w = int(rng.integers(1, 90)) if mode == 0 else int(rng.integers(60, 1500))
    h = int(rng.integers(1, 60)) if mode == 1 else int(rng.integers(20, 600))
    shape = (h, w)
  elif rng.random() < 0.5:
    # wide planes, few of them: the 3-D block form's tiles of 112 x 56 with one to four tile
    # columns, every box start modulo 16 (round 6: the first and last tile of a row store
    # the columns the alignment drops - soda_hip_kernel.edge_slack)
    shape = (int(rng.integers(2, 48)), int(rng.integers(64, 150)), int(rng.integers(128, 480)))
  else:
    hi = 40 if rng.random() < 0.3 else 180
    shape = tuple(int(rng.integers(1, hi)) for _ in range(3))
  inputs = gpu_util.random_inputs(spec, shape, seed=seed, small_ints=(app == 'sobel2d'))
  max_depth = int(rng.choice([0, 0, 0, -1, 1, 2, 4, 8, 12, 16, 20]))
  try:
    prog.set_max_depth(max_depth)
    got = prog.run_numpy(inputs, iterate=iterate)
    want = orc.run(inputs, iterate=iterate)
    sl = orc.valid_slices(tuple(reversed(shape)), iterate)
    for name, g in zip(spec['outputs'], got):
      if not np.array_equal(g[sl], want[name][sl], equal_nan=True):
        failures += 1
        print('MISMATCH', app, shape, iterate, max_depth, name, flush=True)
  except Exception as e:
    failures += 1
    print('EXCEPTION', app, shape, iterate, max_depth, repr(e)[:300], flush=True)
  if (seed - first) % 25 == 24:
    print('%d cases, %d failures, %.0f s' % (seed - first + 1, failures, time.time() - t0),
          flush=True)
print('done, %d failures' % failures)
