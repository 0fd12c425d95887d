#!/usr/bin/env python3
"""Probe (round 4): the depth-1 kernel as ONE wavefront per strip fed through an LDS ring
(kernel_stream2d_wp with a single group: rows arrive by LDS-direct loads, N - 2 in flight
without registers, counted waits) against the shipped depth-1 kernel, under caps on the
workgroups per CU - does decoupling the loads from registers let the kernel follow the
copy's best shape (tools/copyceil.hip: one workgroup per CU on 1024-row chunks)?
usage: k1_ring_probe.py app N ring [ring ...]     (SODA_HIP_WGS_PER_CU from the environment)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'soda-compiler_amd')]
import numpy as np
from soda_hip import frontend
from soda_hip.codegen import kernel, kernel_stream2d, kernel_stream2d_wp, spec as specmod
from soda_hip.runtime import host

app, n = sys.argv[1], int(sys.argv[2])
st = frontend.load(os.path.join(ROOT, 'tests', 'samples', app + '.soda'), iterate=1)
spec = specmod.spec_from_stencil(st)
dt = np.dtype(specmod.NUMPY_NAME[spec['inputs'][0]['c_type']])
rng = np.random.default_rng(1)
a = rng.random((n, n), dtype=np.float32).astype(dt) if dt.kind == 'f' else \
    rng.integers(0, 65536, size=(n, n)).astype(dt)
din = host.DeviceArray(a.nbytes); din.upload(a)
dout = host.DeviceArray(a.nbytes); dout.zero()
original = kernel_stream2d.emit
for ring in [0] + [int(v) for v in sys.argv[3:]]:
  def patched(spec_, depth, **kw):
    if depth == 1 and ring:
      return kernel_stream2d_wp.emit(spec_, 1, cols=kw.get('cols'), chunk_rows=kw.get('chunk_rows', 256),
                                     prefetch=3, align='full', groups=1, ring=ring, vgpr_budget=250,
                                     max_period=ring)
    return original(spec_, depth, **kw)
  kernel_stream2d.emit = patched
  try:
    text, table = kernel.generate(spec)
  finally:
    kernel_stream2d.emit = original
  path = '/tmp/k1ring_%d.hsaco' % os.getpid()
  kernel.compile_to_code_object(text, path)
  prog = host.open_program(blob=path, spec=spec)
  t = prog.sweep_timed([din.ptr], [dout.ptr], [n, n], 1, warmup=10, repeats=10)
  k1 = [k for k in table if k['kind'] == 'fused' and k['depth'] == 1][0]
  print('%-8s ring %2d  cap %s  %8.1f us  (%s, tile %s, est %s VGPRs)' % (
      app, ring, os.environ.get('SODA_HIP_WGS_PER_CU', '-'), t['kernel_us'],
      'ring form' if k1.get('groups') else 'shipped', k1['tile'][:2], k1.get('est_vgprs')), flush=True)
  prog.close(); prog.blob.unload()
