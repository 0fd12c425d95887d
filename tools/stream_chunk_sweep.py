#!/usr/bin/env python3
"""Chunk length of the shallow (HBM-bound) kernels: one launch of the shipped depth-D
kernel under SODA_HIP_CHUNK_ROWS = each value (0 = the launcher's own choice) and
SODA_HIP_WGS_PER_CU caps, fastest of the library's per-launch events.
usage: stream_chunk_sweep.py app N depth 'chunks' 'caps'   e.g. jacobi2d 16384 1 0,8,16,32 0,2"""
import sys as _sys
if len(_sys.argv) > 1 and _sys.argv[1] in ('-h', '--help'):   # usage = the text above
  print(__doc__)
  _sys.exit(0)
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'soda-compiler_amd'), os.path.join(ROOT, 'tests')]
import numpy as np
import gpu_util
from soda_hip.runtime import host

app, n, depth = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
chunks = [int(v) for v in sys.argv[4].split(',')]
caps = [int(v) for v in (sys.argv[5] if len(sys.argv) > 5 else '-1').split(',')]
spec = gpu_util.load_spec(app, iterate=depth)
dims = [n] * spec['dim']
shape = tuple(reversed(dims))
from soda_hip.codegen import spec as specmod
dt = np.dtype(specmod.NUMPY_NAME[spec['inputs'][0]['c_type']])
rng = np.random.default_rng(1)
a = rng.random(shape, dtype=np.float32).astype(dt) if dt.kind == 'f' else \
    rng.integers(0, 65536, size=shape).astype(dt)
din = host.DeviceArray(a.nbytes); din.upload(a)
dout = host.DeviceArray(a.nbytes); dout.zero()
os.environ['SODA_HIP_TUNING'] = '1'
print('%s %s depth %d: us per launch (rows: cap on workgroups per CU, -1 = the kernel\'s own; '
      'columns: chunk, 0 = the launcher\'s own)' % (app, 'x'.join(map(str, dims)), depth))
print('cap   ' + ''.join('%9d' % c for c in chunks))
for cap in caps:
  row = []
  for chunk in chunks:
    os.environ.pop('SODA_HIP_CHUNK_ROWS', None)
    os.environ.pop('SODA_HIP_WGS_PER_CU', None)
    if chunk:
      os.environ['SODA_HIP_CHUNK_ROWS'] = str(chunk)
    if cap >= 0:
      os.environ['SODA_HIP_WGS_PER_CU'] = str(cap)
    prog = host.open_program(blob=os.path.join(gpu_util.BLOBS, app + '.hsaco'), spec=spec)
    prog.set_max_depth(depth)
    t = prog.sweep_timed([din.ptr], [dout.ptr], dims, depth, warmup=3, repeats=6)
    row.append(t['fastest_us'] if 'fastest_us' in t else t['kernel_us'])
    prog.close()
  print('%-6d' % cap + ''.join('%9.1f' % v for v in row), flush=True)
