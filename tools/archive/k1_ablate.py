#!/usr/bin/env python3
"""Timing-only ablations of the generated depth-1 2-D kernel (results may be WRONG at
chunk and array edges): which of its guards costs time.  The variants drop guards, so
the kernel runs on a sub-array of N x (N - 64) rows that starts 32 rows inside the
N x N allocations: rows it touches beyond its own array are still allocated memory.
usage: k1_ablate.py app N"""
import os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'soda-compiler_amd')]
import numpy as np
from soda_hip import frontend
from soda_hip.codegen import kernel, spec as specmod
from soda_hip.runtime import host

app, n = sys.argv[1], int(sys.argv[2])
st = frontend.load(os.path.join(ROOT, 'tests', 'samples', app + '.soda'), iterate=1)
spec = specmod.spec_from_stencil(st)
text, table = kernel.generate(spec, max_depth=1)
dt = np.dtype(specmod.NUMPY_NAME[spec['inputs'][0]['c_type']])
rng = np.random.default_rng(1)
a = rng.random((n, n), dtype=np.float32).astype(dt) if dt.kind == 'f' else \
    rng.integers(0, 65536, size=(n, n)).astype(dt)
din = host.DeviceArray(a.nbytes); din.upload(a)
dout = host.DeviceArray(a.nbytes); dout.zero()


def no_row_clamp(t):
  return re.sub(r' if \(row > H - 1\) row = H - 1;', '', t)


def no_y_check(t):     # every step stores (rows outside the chunk too: WRONG, timing only)
  return re.sub(r'if \(y >= y0 && y < y1\) \{', 'if (y >= -8 && y < H + 8) {', t)


def no_ragged(t):      # only the all-or-nothing vector store
  return re.sub(r'\} else \{\n(?:\s+if \(x \+ \d+ >= st_lo && x \+ \d+ < st_hi\) q\[\d+\] = out_row\[\d+\];\n)+\s+\}',
                '}', t)


def no_fill_check(t):
  return re.sub(r'if \(n \+ \d+ >= \d+\) \{', '{', t)


variants = [('baseline', lambda t: t), ('no row clamp', no_row_clamp),
            ('no y check', no_y_check), ('no ragged path', no_ragged),
            ('no fill check', no_fill_check),
            ('none of them', lambda t: no_fill_check(no_ragged(no_y_check(no_row_clamp(t)))))]
for name, fn in variants:
  src = fn(text)
  assert name == 'baseline' or src != text, name
  path = '/tmp/k1ab_%d.hsaco' % os.getpid()
  kernel.compile_to_code_object(src, path)
  prog = host.open_program(blob=path, spec=spec)
  skip = 32 * n * dt.itemsize
  t = prog.sweep_timed([din.ptr + skip], [dout.ptr + skip], [n, n - 64], 1, warmup=6,
                       repeats=6)
  print('%-20s %8.1f us  [%s]' % (name, t['dominant_us'] / t['dominant_launches'],
                                   t['dominant_name']), flush=True)
  prog.close(); prog.blob.unload()
