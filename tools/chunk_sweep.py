#!/usr/bin/env python3
"""Per-launch time of fused-kernel variants on several box shapes and chunk
lengths (one launch = `depth` iterations).  For the launcher's cost model.
usage: chunk_sweep.py app 'WxH,WxH,...' 'depth;key=val,...;chunk,chunk,...' ...
  chunk 0 = the launcher's own choice.  Offline hipcc builds, as the blobs.
  keys `flags=-mllvm:-opt=value` (extra hipcc flags, ':' between words) and `noflags=1`
  (without the per-program flags of the kernel text) are the compiler's, the rest the
  generator's."""
import sys as _sys
if len(_sys.argv) > 1 and _sys.argv[1] in ('-h', '--help'):   # usage = the text above
  print(__doc__)
  _sys.exit(0)
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'soda-compiler_amd')]
import numpy as np  # noqa: E402
from soda_hip import frontend  # noqa: E402
from soda_hip.codegen import kernel, spec as specmod  # noqa: E402
from soda_hip.runtime import host  # noqa: E402

os.environ['SODA_HIP_TUNING'] = '1'      # the run-time reads chunk overrides only then
app = sys.argv[1]
shapes = [tuple(int(v) for v in s.split('x')) for s in sys.argv[2].split(',')]
rng = np.random.default_rng(1)
for variant in sys.argv[3:]:
  parts = variant.split(';')
  depth = int(parts[0])
  opts = {}
  if len(parts) > 1 and parts[1]:
    opts = dict(kv.split('=') for kv in parts[1].split(','))
    opts = {k: (int(v) if v.lstrip('-').isdigit() else v) for k, v in opts.items()}
  chunks = [int(c) for c in parts[2].split(',')] if len(parts) > 2 and parts[2] else [0]
  st = frontend.load(os.path.join(ROOT, 'tests', 'samples', app + '.soda'),
                     iterate=max(depth, 16))
  spec = specmod.spec_from_stencil(st)
  t0 = time.time()
  # flags=-mllvm:-amdgpu-sched-strategy=iterative-ilp -> extra hipcc flags; noflags=1 drops the
  # per-program flags the kernel text records (an -mllvm option may be given only once)
  flags = str(opts.pop('flags', '')).split(':') if opts.get('flags') else []
  noflags = opts.pop('noflags', 0)
  text, table = kernel.generate(spec, depths=[depth], **opts)
  if noflags:
    text = '\n'.join(l for l in text.split('\n') if not l.startswith(kernel.FLAGS_MARK))
  path = '/tmp/sweep_%d.hsaco' % os.getpid()
  try:
    kernel.compile_to_code_object(text, path, extra_flags=flags)
  except Exception as e:
    print(variant, 'COMPILE FAILED', str(e)[:200], flush=True)
    continue
  entry = [k for k in table if k['kind'] == 'fused' and k['depth'] == depth]
  print('# %s: compile %.0fs, %s' % (variant, time.time() - t0, {
      k: entry[0].get(k) for k in ('fill_rows', 'w_out', 'est_vgprs', 'groups',
                                   'ring', 'pairs', 'block')} if entry else 'NOT FUSED'),
        flush=True)
  for dims in shapes:
    a = rng.random((dims[1], dims[0]), dtype=np.float32)
    din = host.DeviceArray(a.nbytes)
    din.upload(a)
    dout = host.DeviceArray(a.nbytes)
    dout.zero()
    for chunk in chunks:
      if chunk:
        os.environ['SODA_HIP_CHUNK_ROWS'] = str(chunk)
      else:
        os.environ.pop('SODA_HIP_CHUNK_ROWS', None)
      prog = host.open_program(blob=path, spec=spec)
      reps = max(4, int(30000.0 / (45.0 * depth * dims[0] * dims[1] / 16384 ** 2)))
      t = prog.sweep_timed([din.ptr], [dout.ptr], list(dims), depth, warmup=reps,
                           repeats=reps)
      valid = specmod.valid_cells(spec, list(dims), depth)
      print('%-44s %6dx%-6d chunk %4d : %8.1f us/launch  %6.2f us/it  %6.0f G/s  [%s]'
            % (variant[:44], dims[0], dims[1], chunk, t['kernel_us'],
               t['kernel_us'] / depth, valid / t['kernel_us'] / 1e3,
               t['dominant_name']), flush=True)
      prog.close()
      prog.blob.unload()
    din.free()
    dout.free()
