#!/usr/bin/env python3
"""Parity of a 3-D fused-kernel generator variant against the oracle on the GPU box:
the block form alone (deep3d=blk) on ragged shapes, several iteration counts, every
depth split the scheduler takes.  usage: check_variant3d.py app 'key=value,...'"""
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
opts = {k: (int(v) if v.lstrip('-').isdigit() else v)
        for k, v in (kv.split('=') for kv in sys.argv[2].split(',') if kv)} if len(sys.argv) > 2 else {}
st = frontend.load(os.path.join(ROOT, 'tests', 'samples', app + '.soda'), iterate=13)
spec = specmod.spec_from_stencil(st)
text, table = kernel.generate(spec, **opts)
prog = host.open_program(source=text, spec=spec)
orc = soda_oracle.Oracle(spec)
ok = True
for iterate, shape in ((4, (30, 64, 128)), (9, (45, 131, 140)), (13, (150, 70, 257)), (1, (20, 64, 130)),
                       (2, (37, 100, 200)), (6, (64, 64, 128)), (8, (100, 200, 300)),
                       (5, (21, 72, 128)), (12, (40, 80, 140)), (7, (50, 150, 260))):
  a = np.random.default_rng(3).random(shape, dtype=np.float32)
  want = orc.run([a], iterate=iterate)[spec['outputs'][0]]
  sl = orc.valid_slices(tuple(reversed(shape)), iterate)
  for max_depth in (0, 2, 1):
    prog.set_max_depth(max_depth)
    got = prog.run_numpy([a], iterate=iterate)[0]
    bad = int((~((got[sl] == want[sl]) | (np.isnan(got[sl]) & np.isnan(want[sl])))).sum())
    print(app, opts, 'iterate', iterate, shape, 'max_depth', max_depth, 'bad', bad, 'of', want[sl].size)
    ok &= bad == 0
sys.exit(0 if ok else 1)
