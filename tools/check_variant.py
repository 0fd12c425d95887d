#!/usr/bin/env python3
"""Parity of a fused-kernel generator variant against the oracle on the GPU box.
usage: check_variant.py app 'key=value,...' [max depth]   (generate() options; with a
max depth the sweeps use kernels of at most that depth: 1 = the depth-1 kernels alone, on
more shapes)"""
import sys as _sys
if len(_sys.argv) > 1 and _sys.argv[1] in ('-h', '--help'):   # usage = the text above
  print(__doc__)
  _sys.exit(0)
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'soda-compiler_amd'), os.path.join(ROOT, 'tests')]
import numpy as np
import torch  # noqa: F401
from soda_hip import frontend
from soda_hip.codegen import kernel, spec as specmod
from soda_hip.runtime import host
from oracle import soda_oracle
app = sys.argv[1]
opts = {k: (int(v) if v.lstrip('-').isdigit() else v) for k, v in (kv.split('=') for kv in sys.argv[2].split(','))} if len(sys.argv) > 2 and sys.argv[2] else {}
ok = True
max_depth = int(sys.argv[3]) if len(sys.argv) > 3 else 0
cases = ((12, (300, 1100)), (25, (257, 1021)), (13, (100, 2049)), (37, (611, 700)), (8, (64, 64)),
         (16, (200, 3000)), (40, (150, 4100)), (28, (90, 1024)))
if max_depth == 1:
  cases = ((1, (300, 1100)), (2, (257, 1021)), (3, (100, 2049)), (1, (611, 700)), (2, (64, 64)),
           (1, (70, 255)), (2, (33, 513)), (1, (9, 11)), (1, (40, 256)), (1, (1500, 768)),
           (3, (50, 4097)))
sample = os.path.join(ROOT, 'tests', 'samples', app + '.soda')
if not os.path.exists(sample):
  sample = os.path.join(ROOT, 'tests', 'samples', 'extra', app + '.soda')
for iterate, shape in cases:
  st = frontend.load(sample, iterate=iterate)
  spec = specmod.spec_from_stencil(st)
  text, table = kernel.generate(spec, **opts)
  prog = host.open_program(source=text, spec=spec)
  if max_depth:
    prog.set_max_depth(max_depth)
  rng = np.random.default_rng(3)
  dt = prog.in_dtypes[0]
  a = rng.random(shape, dtype=np.float32).astype(dt) if dt.kind == 'f' else rng.integers(0, 65536, size=shape).astype(dt)
  got = prog.run_numpy([a], iterate=iterate)[0]
  orc = soda_oracle.Oracle(spec)
  want = orc.run([a], iterate=iterate)[spec['outputs'][0]]
  sl = orc.valid_slices(tuple(reversed(shape)), iterate)
  bad = int((~((got[sl] == want[sl]) | (np.isnan(got[sl]) & np.isnan(want[sl])))).sum())
  print(app, opts, 'iterate', iterate, shape, [k['name'] + ('/wp' if k.get('groups') else '') for k in table if k['kind'] == 'fused'], 'bad', bad, 'of', want[sl].size)
  ok &= bad == 0
  prog.close(); prog.blob.unload()
sys.exit(0 if ok else 1)
