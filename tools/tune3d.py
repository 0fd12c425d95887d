#!/usr/bin/env python3
"""Times and checks variants of the 3-D fused kernels on the GPU box.
usage: tune3d.py app N iterate 'key=value,...' ...   (options of kernel.generate;
wp_* go to the depth-4 wave-pipelined kernel).  Each variant is first compared
with the oracle on a small ragged grid, then timed on N^3."""
import sys as _sys
if len(_sys.argv) > 1 and _sys.argv[1] in ('-h', '--help'):   # usage = the text above
  print(__doc__)
  _sys.exit(0)
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'soda-compiler_amd')]
import numpy as np
import torch  # noqa: F401
from soda_hip import frontend
from soda_hip.codegen import kernel, spec as specmod
from soda_hip.runtime import host
from oracle import soda_oracle

app, n, iterate = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
st = frontend.load(os.path.join(ROOT, 'tests', 'samples', app + '.soda'), iterate=iterate)
spec = specmod.spec_from_stencil(st)
rng = np.random.default_rng(1)
dins = []
for _ in spec['inputs']:
  a = rng.random((n, n, n), dtype=np.float32)
  d = host.DeviceArray(a.nbytes); d.upload(a); dins.append(d)
dout = host.DeviceArray(a.nbytes); dout.zero()
small = [rng.random((45, 70, 131), dtype=np.float32) for _ in spec['inputs']]
orc = soda_oracle.Oracle(spec)
for variant in sys.argv[4:] or ['']:
  opts = {k: ([int(x) for x in v.split('/')] if '/' in v else
              int(v) if v.lstrip('-').isdigit() else v)
          for k, v in (kv.split('=', 1) for kv in variant.split(',') if kv)}
  # flags=-mllvm:-amdgpu-sched-strategy=max-ilp  -> extra hipcc flags (TUNE_HIPCC=1)
  flags = opts.pop('flags', '').split(':') if opts.get('flags') else []
  text, table = kernel.generate(spec, **opts)
  try:
    if os.environ.get('TUNE_HIPCC'):     # offline compile, as the shipped blobs
      path = '/tmp/tune3d_%d.hsaco' % os.getpid()
      kernel.compile_to_code_object(text, path, extra_flags=flags)
      prog = host.open_program(blob=path, spec=spec)
    else:
      prog = host.open_program(source=text, spec=spec)
  except Exception as e:
    print(variant, 'FAILED', str(e)[:300]); continue
  bad = []
  for it in ((4, 9, 13) if len(spec['inputs']) == len(spec['outputs']) else (1,)):
    got = prog.run_numpy(small, iterate=it)[0]
    want = orc.run(small, iterate=it)[spec['outputs'][0]]
    sl = orc.valid_slices(tuple(reversed(small[0].shape)), it)
    bad.append(int((got[sl] != want[sl]).sum()))
  t = prog.sweep_timed([d.ptr for d in dins], [dout.ptr], [n, n, n], iterate, warmup=3,
                       repeats=5)
  print('%-40s bad %s  %9.1f us/sweep  %d launches  dominant %s %.1f us' % (
      variant or '(default)', bad, t['kernel_us'], t['launches'], t['dominant_name'],
      t['dominant_us'] / max(1, t['dominant_launches'])), flush=True)
  prog.close(); prog.blob.unload()
