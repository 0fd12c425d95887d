import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'soda-compiler_amd'), os.path.join(ROOT, 'tests')]
import numpy as np
import torch
from soda_hip import frontend
from soda_hip.codegen import kernel, spec as specmod
from soda_hip.runtime import host
from oracle import soda_oracle
import importlib.util
s_ = importlib.util.spec_from_file_location('rp', os.path.join(ROOT, 'tests/test_gpu_random_programs.py')); m = importlib.util.module_from_spec(s_); s_.loader.exec_module(m)
seed = int(sys.argv[1])
rng = np.random.default_rng(1000 + seed)
text, dim, dtype, iterate = m.random_program(rng, seed)
print(text)
spec = specmod.spec_from_stencil(frontend.loads(text))
shape = (41, 333) if dim == 2 else (19, 23, 150)
inputs = []
for t in spec['inputs']:
  dt = np.dtype(specmod.NUMPY_NAME[t['c_type']])
  inputs.append(rng.random(shape, dtype=np.float32) + np.float32(0.5) if dt.kind == 'f' else rng.integers(0, 200, size=shape).astype(dt))
orc = soda_oracle.Oracle(spec)
for it in range(1, iterate + 1):
  want = orc.run(inputs, iterate=it)['out']
  sl = orc.valid_slices(tuple(reversed(shape)), it)
  for opts in (dict(), dict(skip_fill=0), dict(prefetch=0), dict(skip_fill=0, prefetch=0)):
    src, table = kernel.generate(spec, **opts)
    prog = host.open_program(source=src, spec=spec)
    for md in (0, 1):
      prog.set_max_depth(md)
      got = prog.run_numpy(inputs, iterate=it)[0]
      bad = np.argwhere(got[sl] != want[sl])
      print('iterate', it, opts, 'max_depth', md, 'bad', len(bad), 'of', got[sl].size, bad[:4].tolist(), 'valid slice', sl)
    prog.close(); prog.blob.unload()
print('--- value probe, iterate 2')
src, table = kernel.generate(spec)
prog = host.open_program(source=src, spec=spec)
w1 = orc.run(inputs, iterate=1)['out']
w2 = orc.run(inputs, iterate=2)['out']
prog.set_max_depth(0)
g2 = prog.run_numpy(inputs, iterate=2)[0]
for pos in [(20, 100), (20, 101), (30, 200), (8, 8)]:
  print(pos, 'got', g2[pos], 'want2', w2[pos], 'want1', w1[pos], 'in', inputs[0][pos])
# is got a shifted version of want?
for dy in range(-4, 5):
  for dx in range(-4, 5):
    a = g2[12:36, 20:300]; b = w2[12 + dy:36 + dy, 20 + dx:300 + dx]
    if np.array_equal(a, b): print('got == want shifted by', dy, dx)
