#!/usr/bin/env python3
"""Parity of one .soda file against the oracle on the GPU box, per depth limit.
usage: check_program.py file.soda HxW[xD] [iterate] [key=value,...]"""
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
path = sys.argv[1]
shape = tuple(int(v) for v in sys.argv[2].split('x'))
st = frontend.load(path, **({'iterate': int(sys.argv[3])} if len(sys.argv) > 3 and sys.argv[3] else {}))
opts = {k: (int(v) if v.lstrip('-').isdigit() else v) for k, v in (kv.split('=') for kv in sys.argv[4].split(','))} if len(sys.argv) > 4 else {}
spec = specmod.spec_from_stencil(st)
text, table = kernel.generate(spec, **opts)
print([(k['name'], k.get('groups', 0), k.get('pairs', 0)) for k in table if k['kind'] == 'fused'])
print('\n'.join(l for l in text.splitlines() if l.startswith('// depth')))
prog = host.open_program(source=text, spec=spec)
rng = np.random.default_rng(7)
inputs = [(rng.random(shape, dtype=np.float32) + np.float32(0.5)).astype(dt) if dt.kind == 'f' else rng.integers(0, 200, size=shape).astype(dt) for dt in prog.in_dtypes]
orc = soda_oracle.Oracle(spec)
want = orc.run(inputs, iterate=spec['iterate'])[spec['outputs'][0]]
sl = orc.valid_slices(tuple(reversed(shape)), spec['iterate'])
for md in (-1, 1, 2, 4, 0):
  prog.set_max_depth(md)
  got, t = prog.run_numpy(inputs, iterate=spec['iterate'], timed=True)
  bad = np.argwhere(got[0][sl] != want[sl])
  print('max_depth', md, 'used depth', t['max_depth'], 'bad', len(bad), 'of', want[sl].size, 'first', bad[:3].tolist(), 'last', bad[-2:].tolist())
