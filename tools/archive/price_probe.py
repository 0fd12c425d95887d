#!/usr/bin/env python3
"""Modelled against measured time of every fused depth of a program on one grid (the
prices the scheduler compares).  usage: price_probe.py app N   (GPU box)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'soda-compiler_amd'), os.path.join(ROOT, 'tests')]
import numpy as np
import gpu_util
from soda_hip.runtime import host
app, n = sys.argv[1], int(sys.argv[2])
spec = gpu_util.load_spec(app, iterate=64)
prog = host.open_program(blob=os.path.join(gpu_util.BLOBS, app + '.hsaco'), spec=spec)
dims = [n] * spec['dim']
nbytes = int(np.prod(dims)) * prog.in_dtypes[0].itemsize
a = host.DeviceArray(nbytes); a.zero()
b = host.DeviceArray(nbytes); b.zero()
for k in prog.kernels:
  if k['kind'] != 'fused':
    continue
  d = k['depth']
  try:
    prog.set_split(dims, d, [d])
  except Exception as e:
    print(k['name'], 'cannot be forced:', str(e)[:80]); continue
  sched = prog.schedule(dims, d)
  t = prog.sweep_timed([a.ptr], [b.ptr], dims, d, warmup=3, repeats=5)
  print('%-24s model %8.1f us   measured %8.1f us   (%s; chunk %s cap %s gbps %s)' % (
      sched[0][0]['name'], sched[0][1], t['fastest_us'], '+'.join(e['name'][-3:] for e, _ in sched),
      k.get('stream_chunk'), k.get('stream_wgs_per_cu'), k.get('stream_gbps')))
  prog.set_split(dims, d, [])
