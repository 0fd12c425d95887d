#!/usr/bin/env python3
"""Where a wavefront of the 3-D block-form kernel spends its cycles: a stamped build
(kernel_stream3d_blk.emit, stamps=<debug buffer>) sums s_memtime deltas per part of
a step; this prints the average per step over all wavefronts that ran.
usage: blk_stamps.py app N 'key=value,...'   (options of kernel.generate; flags= too)"""
import sys as _sys
if len(_sys.argv) > 1 and _sys.argv[1] in ('-h', '--help'):   # usage = the text above
  print(__doc__)
  _sys.exit(0)
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'soda-compiler_amd')]
import numpy as np
from soda_hip import frontend
from soda_hip.codegen import kernel, spec as specmod
from soda_hip.runtime import host

app, n = sys.argv[1], int(sys.argv[2])
depth = 4
st = frontend.load(os.path.join(ROOT, 'tests', 'samples', app + '.soda'), iterate=depth)
spec = specmod.spec_from_stencil(st)
a = np.random.default_rng(1).random((n, n, n), dtype=np.float32)
din = host.DeviceArray(a.nbytes); din.upload(a)
dout = host.DeviceArray(a.nbytes); dout.zero()
WORDS = 4096 * 8 * 16
dbg = host.DeviceArray(WORDS * 8); dbg.zero()
for variant in sys.argv[3:] or ['deep3d=blk']:
  opts = {k: (int(v) if v.lstrip('-').isdigit() else v)
          for k, v in (kv.split('=', 1) for kv in variant.split(',') if kv)}
  flags = opts.pop('flags', '').split(':') if opts.get('flags') else []
  for stamped in (0, 1):
    o = dict(opts)
    if stamped:
      o['blk_stamps'] = dbg.ptr
    text, table = kernel.generate(spec, **o)
    path = '/tmp/stamps_%d.hsaco' % os.getpid()
    kernel.compile_to_code_object(text, path, extra_flags=flags)
    prog = host.open_program(blob=path, spec=spec)
    dbg.zero()
    t = prog.sweep_timed([din.ptr], [dout.ptr], [n, n, n], depth, warmup=2, repeats=3)
    print('%-50s %s  %8.1f us [%s]' % (variant, 'stamped' if stamped else 'plain  ',
                                       t['kernel_us'], t['dominant_name']), flush=True)
    if stamped:
      raw = dbg.download((WORDS // 16, 16), np.dtype(np.uint64))
      ran = raw[raw[:, 15] > 0]
      steps = ran[:, 15].astype(np.float64)
      # every launch overwrites the buffer: these are the last launch's sums
      parts = ran[:, :depth + 2].astype(np.float64)
      per_step = parts / steps[:, None]
      print('   wavefronts %d, steps %.0f, cycles per step %.0f (median wavefront %.0f)'
            % (len(ran), steps.mean(), per_step.sum(axis=1).mean(),
               np.median(per_step.sum(axis=1))))
      names = ['input plane'] + ['level %d' % k for k in range(1, depth + 1)] + ['barrier']
      for k, nm in enumerate(names):
        print('   %-12s %7.0f cycles  %5.1f %%' % (nm, per_step[:, k].mean(),
              100 * per_step[:, k].mean() / per_step.sum(axis=1).mean()))
    prog.close(); prog.blob.unload()
