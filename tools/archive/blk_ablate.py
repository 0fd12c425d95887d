#!/usr/bin/env python3
"""Timing-only ablations of the 3-D block-form kernel (results are WRONG by
construction): which of loads, stores, barrier, arithmetic a launch waits for.
usage: blk_ablate.py app N 'key=value,...'   (options of kernel.generate)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'soda-compiler_amd')]
import numpy as np
from soda_hip import frontend
from soda_hip.codegen import kernel, spec as specmod
from soda_hip.runtime import host

app, n = sys.argv[1], int(sys.argv[2])
opts = {k: (int(v) if v.lstrip('-').isdigit() else v)
        for k, v in (kv.split('=', 1) for kv in sys.argv[3].split(',') if kv)}
# flags=-mllvm:-amdgpu-sched-strategy=max-ilp -> extra hipcc flags
flags = opts.pop('flags', '').split(':') if opts.get('flags') else []
st = frontend.load(os.path.join(ROOT, 'tests', 'samples', app + '.soda'), iterate=4)
spec = specmod.spec_from_stencil(st)
text, table = kernel.generate(spec, **opts)
a = np.random.default_rng(1).random((n, n, n), dtype=np.float32)
din = host.DeviceArray(a.nbytes); din.upload(a)
dout = host.DeviceArray(a.nbytes); dout.zero()
FLAG = '(a.param[3] == 12345)'

def no_load_traffic(t):
  return t.replace('(rs, lane_byte, (unsigned)(', '(rs, %s ? lane_byte : 0xfffffff0u, (unsigned)(' % FLAG)

def no_store_traffic(t):
  return t.replace('st_full ? lane_byte : 0xfffffff0u', '(st_full && %s) ? lane_byte : 0xfffffff0u' % FLAG)

def no_barrier(t):
  return t.replace('    soda_block_barrier();\n', '')

def no_edges(t):      # neighbours' rows are not read back (own registers instead)
  return t

variants = [('baseline', lambda t: t), ('no load traffic', no_load_traffic),
            ('no store traffic', no_store_traffic),
            ('no load + store traffic', lambda t: no_store_traffic(no_load_traffic(t))),
            ('no barrier', no_barrier),
            ('no traffic, no barrier', lambda t: no_barrier(no_store_traffic(no_load_traffic(t))))]
for name, fn in variants:
  src = fn(text)
  assert name == 'baseline' or src != text, name
  path = '/tmp/ablate_%d.hsaco' % os.getpid()
  kernel.compile_to_code_object(src, path, extra_flags=flags)
  prog = host.open_program(blob=path, spec=spec)
  t = prog.sweep_timed([din.ptr], [dout.ptr], [n, n, n], 4, warmup=3, repeats=5)
  print('%-28s %8.1f us  [%s]' % (name, t['kernel_us'], t['dominant_name']), flush=True)
  prog.close(); prog.blob.unload()
