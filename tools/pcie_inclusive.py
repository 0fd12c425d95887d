#!/usr/bin/env python3
"""One-shot host-buffer call (H2D + warm-up + timed sweep + D2H) on the GPU box."""
import sys as _sys
if len(_sys.argv) > 1 and _sys.argv[1] in ('-h', '--help'):   # usage = the text above
  print(__doc__)
  _sys.exit(0)
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0]=[ROOT, ROOT+'/soda-compiler_amd', ROOT+'/tests']
import numpy as np, torch
import gpu_util
prog = gpu_util.open_prebuilt('jacobi2d')
n=16384
a=np.random.default_rng(1).random((n,n),dtype=np.float32)
out=np.zeros_like(a)
for rep in range(3):
  t0=time.perf_counter(); prog.run_buffers([a],[out],1000); t1=time.perf_counter()
  print('run_buffers 16384^2 x1000: %.1f ms' % ((t1-t0)*1e3))
