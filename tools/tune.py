#!/usr/bin/env python3
"""Times single launches of fused-kernel variants on the GPU box.
usage: tune.py app N iterate 'depth,cols,chunk_rows,prefetch' ...
"""
import sys as _sys
if len(_sys.argv) > 1 and _sys.argv[1] in ('-h', '--help'):   # usage = the text above
  print(__doc__)
  _sys.exit(0)
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'soda-compiler_amd'), os.path.join(ROOT, 'tests')]
import numpy as np
from soda_hip import frontend
from soda_hip.codegen import kernel, spec as specmod
from soda_hip.runtime import host

app, iterate = sys.argv[1], int(sys.argv[3])
size = [int(v) for v in sys.argv[2].split('x')]
st = frontend.load(os.path.join(ROOT, 'tests', 'samples', app + '.soda'), iterate=iterate)
spec = specmod.spec_from_stencil(st)
dims = (size * spec['dim'])[:spec['dim']] if len(size) == 1 else size
w, h = dims[0], int(np.prod(dims[1:]))
rng = np.random.default_rng(1)
dt = np.dtype(specmod.NUMPY_NAME[spec['inputs'][0]['c_type']])
dt = np.dtype(specmod.NUMPY_NAME[spec['inputs'][0]['c_type']])
a = rng.random((h, w), dtype=np.float32).astype(dt) if dt.kind == 'f' else rng.integers(0, 65536, size=(h, w)).astype(dt)
din = host.DeviceArray(a.nbytes); din.upload(a)
dout = host.DeviceArray(a.nbytes); dout.zero()
for variant in sys.argv[4:]:
  parts = variant.split(',')
  depth, cols, chunk, pf = [int(v) for v in parts[:4]]
  extra = dict(kv.split('=') for kv in parts[4:])
  extra = {k: (int(v) if v.lstrip('-').isdigit() else v) for k, v in extra.items()}
  t0 = time.time()
  if 'wave_groups' not in extra:
    extra.setdefault('vgpr_budget', 400)
  noflags = extra.pop('noflags', 0)     # compile without the per-program -mllvm flags
  text, table = kernel.generate(spec, depths=[depth], cols=cols, chunk_rows=chunk, prefetch=pf, **extra)
  if noflags:
    text = '\n'.join(l for l in text.split('\n') if not l.startswith(kernel.FLAGS_MARK))
  try:
    if os.environ.get('TUNE_HIPCC'):     # offline compile, as the shipped blobs
      path = '/tmp/tune_%d.hsaco' % os.getpid()
      kernel.compile_to_code_object(text, path)
      prog = host.open_program(blob=path, spec=spec)
    else:
      prog = host.open_program(source=text, spec=spec)
  except Exception as e:
    print(variant, 'FAILED', str(e)[:300]); continue
  tc = time.time() - t0
  t = prog.sweep_timed([din.ptr], [dout.ptr], dims, iterate, warmup=int(os.environ.get("TUNE_WARMUP", "6")), repeats=int(os.environ.get("TUNE_REPEATS", "6")))
  valid = specmod.valid_cells(spec, dims, iterate)
  print('%-28s compile %.1fs  %8.1f us/sweep  %d launches  dominant %s %.1f us  -> %.0f Gcell/s valid (%.0f nominal)' % (
      variant, tc, t['kernel_us'], t['launches'], t['dominant_name'], t['dominant_us'] / t['dominant_launches'],
      valid / t['kernel_us'] / 1e3, float(w) * h * iterate / t['kernel_us'] / 1e3), flush=True)
  prog.close(); prog.blob.unload()
