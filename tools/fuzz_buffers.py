#!/usr/bin/env python3
"""Fuzzing run on the GPU box: the host-buffer entry (soda_hip_run_buffers, what the
generated `<app>(buffer_t...)` calls) on random shapes: outputs pre-filled with a
marker, only each output's valid box may change and must equal the oracle.
usage: fuzz_buffers.py first_seed count"""
import sys as _sys
if len(_sys.argv) > 1 and _sys.argv[1] in ('-h', '--help'):   # usage = the text above
  print(__doc__)
  _sys.exit(0)
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'soda-compiler_amd'), os.path.join(ROOT, 'tests')]
import numpy as np
import gpu_util
from soda_hip.codegen import spec as specmod

first, count = int(sys.argv[1]), int(sys.argv[2])
apps = ['jacobi2d', 'blur', 'sobel2d', 'denoise2d', 'skew2d', 'jacobi3d', 'heat3d',
        'denoise3d', 'hyper4d', 'outchain', 'casts']
progs, oracles = {}, {}
failures = 0
for seed in range(first, first + count):
  rng = np.random.default_rng(660000 + seed)
  app = apps[int(rng.integers(0, len(apps)))]
  if app not in progs:
    progs[app] = gpu_util.open_prebuilt(app)
    oracles[app] = gpu_util.make_oracle(progs[app].spec)
  prog, orc = progs[app], oracles[app]
  spec = prog.spec
  dim = spec['dim']
  chain = len(spec['inputs']) == len(spec['outputs'])
  iterate = int(rng.integers(1, 9)) if chain else 1
  hi = {2: 400, 3: 70, 4: 24}[dim]
  shape = tuple(int(rng.integers(1, hi)) for _ in range(dim))
  inputs = gpu_util.random_inputs(spec, shape, seed=seed, small_ints=(app == 'sobel2d'))
  outs = [np.full(shape, 77, dtype=dt) for dt in prog.out_dtypes]
  try:
    prog.run_buffers(inputs, outs, iterate)
    want = orc.run(inputs, iterate=iterate)
    boxes = specmod.iteration_boxes(spec, iterate)[-1]
    for name, o in zip(spec['outputs'], outs):
      lo, hi_ = boxes[name]
      own = tuple(slice(-lo[d], max(-lo[d], shape[::-1][d] - hi_[d]))
                  for d in reversed(range(dim)))
      inside = np.zeros(shape, bool)
      inside[own] = True
      if any(shape[::-1][d] + lo[d] - hi_[d] <= 0 for d in range(dim)):
        inside[:] = False
      ok = np.array_equal(o[inside], want[name][inside], equal_nan=True) and \
          (o[~inside] == 77).all()
      if not ok:
        failures += 1
        print('MISMATCH', app, shape, iterate, name, flush=True)
  except Exception as e:
    failures += 1
    print('EXCEPTION', app, shape, iterate, repr(e)[:300], flush=True)
print('buffers: %d cases, %d failures' % (count, failures))
