#!/usr/bin/env python3
"""Fuzzing run on the GPU box: fresh seeds of the random-program generators of the
tests (tests/random_programs.py), HIP path (hiprtc) against the CPU oracle for every
depth split.  Not a test: a hunt; a failing program goes into the committed families.
usage: fuzz_gpu.py family first_seed count [generator options k=v,...]
(family: plain ops struct cube deep)"""
import sys as _sys
if len(_sys.argv) > 1 and _sys.argv[1] in ('-h', '--help'):   # usage = the text above
  print(__doc__)
  _sys.exit(0)
import os
import sys
import tempfile
import time
import traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'soda-compiler_amd'), os.path.join(ROOT, 'tests')]
import numpy as np
import random_programs as rp
from soda_hip import frontend
from soda_hip.codegen import kernel, spec as specmod
from soda_hip.runtime import host
from oracle import soda_oracle

family, first, count = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
gen = {'plain': rp.random_program, 'deep': rp.random_program, 'ops': rp.operator_program,
       'struct': rp.structure_program, 'cube': rp.cube_program}[family]
scratch = tempfile.mkdtemp(prefix='fuzz_oracle_')
failures = 0
t_start = time.time()
for seed in range(first, first + count):
  rng = np.random.default_rng(770000 + seed)
  text, dim, dtype, iterate = gen(rng, seed)
  if family == 'deep':
    if dim != 2 or 'in1' in text:
      continue
    deep = int(rng.integers(8, 30))
    text = text.replace('iterate: %d\n' % iterate, 'iterate: %d\n' % deep)
    iterate = deep
  try:
    spec = specmod.spec_from_stencil(frontend.loads(text))
    boxes = specmod.iteration_boxes(spec, iterate)[-1]
    shape = [61, 333] if dim == 2 else [24, 27, 150]
    if family == 'deep':
      shape = [400, 700]
    if family == 'cube':
      shape = [70, 90, 200]
    for name in spec['outputs']:
      lo, hi = boxes[name]
      for d in range(dim):
        shape[dim - 1 - d] = max(shape[dim - 1 - d], hi[d] - lo[d] + 20)
    gen_options = dict(kv.split('=') for kv in sys.argv[4].split(',')) \
        if len(sys.argv) > 4 else {}
    gen_options = {k: (int(v) if v.lstrip('-').isdigit() else v)
                   for k, v in gen_options.items()}
    if gen_options.get('blk_pairs') and ('int32' in text or 'local' in text):
      gen_options = {k: v for k, v in gen_options.items() if k != 'blk_pairs'}
    src, table = kernel.generate(spec, **gen_options)
    prog = host.open_program(source=src, spec=spec)
    orc = soda_oracle.Oracle(spec, build_dir=scratch)
    inputs = []
    for t in spec['inputs']:
      dt = np.dtype(specmod.NUMPY_NAME[t['c_type']])
      if dt.kind == 'f':
        inputs.append((rng.random(tuple(shape), dtype=np.float32) + np.float32(0.5)).astype(dt))
      else:
        inputs.append(rng.integers(0, 200, size=tuple(shape)).astype(dt))
    wants = orc.run(inputs, iterate=iterate)
    fused = [k['depth'] for k in table if k['kind'] == 'fused']
    for max_depth in sorted({0, 1, -1} if fused else {-1}):
      prog.set_max_depth(max_depth)
      gots = prog.run_numpy(inputs, iterate=iterate)
      for name, got in zip(spec['outputs'], gots):
        blo, bhi = boxes[name]
        own = tuple(slice(-blo[d], max(-blo[d], got.shape[::-1][d] - bhi[d]))
                    for d in reversed(range(dim)))
        if not np.array_equal(got[own], wants[name][own], equal_nan=True):
          failures += 1
          print('MISMATCH %s seed %d output %s max_depth %d fused %s\n%s' % (
              family, seed, name, max_depth, fused, text), flush=True)
    prog.close()
    prog.blob.unload()
  except Exception:
    failures += 1
    print('EXCEPTION %s seed %d\n%s\n%s' % (family, seed, text, traceback.format_exc()),
          flush=True)
  if (seed - first) % 20 == 19:
    print('%s: %d programs, %d failures, %.0f s' % (family, seed - first + 1, failures,
                                                   time.time() - t_start), flush=True)
print('%s: done, %d failures' % (family, failures))
