#!/usr/bin/env python3
"""Timing-only probe (results are WRONG by construction): what the 3-D block kernel's z-chunk
FILL planes cost in memory traffic.  A chunk of n planes loads n + 8 input planes (depth 4);
the 8 outside its own range are its neighbours' planes, read again at another time.  Here those
loads are redirected to the chunk's own boundary plane (same instruction count, no new bytes):
the difference to the shipped kernel is the most that time-aligning neighbouring chunks (walking
alternate chunks in opposite z directions) could gain.
usage: fill_reads_probe.py [N iterate]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'soda-compiler_amd')]
if len(sys.argv) > 1 and sys.argv[1] == '--child':
  import numpy as np
  from soda_hip import frontend
  from soda_hip.codegen import kernel, spec as specmod
  from soda_hip.runtime import host
  n, iterate, hack = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4] == '1'
  st = frontend.load(os.path.join(ROOT, 'tests', 'samples', 'jacobi3d.soda'), iterate=iterate)
  spec = specmod.spec_from_stencil(st)
  text, _ = kernel.generate(spec, deep3d='blk')
  if hack:
    a0 = text.index('DEV void jacobi3d_fused_k4b_band')
    a1 = text.index('GLOBAL', a0)
    body = text[a0:a1].replace(
        'if (zz > D - 1) zz = D - 1;',
        'if (zz > D - 1) zz = D - 1; if (zz < z0) zz = z0; if (zz > z1 - 1) zz = z1 - 1;')
    assert body != text[a0:a1]
    text = text[:a0] + body + text[a1:]
  path = '/tmp/fillprobe_%d.hsaco' % os.getpid()
  kernel.compile_to_code_object(text, path)
  prog = host.open_program(blob=path, spec=spec)
  a = np.random.default_rng(1).random((n, n, n), dtype=np.float32)
  din = host.DeviceArray(a.nbytes); din.upload(a)
  dout = host.DeviceArray(a.nbytes); dout.zero()
  t = prog.sweep_timed([din.ptr], [dout.ptr], [n, n, n], iterate, warmup=10, repeats=3)
  print('TOTAL %.1f' % t['kernel_us'])
  sys.exit(0)
n, iterate = (sys.argv[1:3] + ['512', '200'])[:2] if len(sys.argv) > 2 else ('512', '200')
for label, hack in (('shipped', '0'), ('fill planes redirected', '1'), ('shipped', '0'),
                    ('fill planes redirected', '1')):
  env = dict(os.environ, SODA_HIP_TUNING='1', SODA_HIP_LAUNCH_TRACE='1')
  p = subprocess.run([sys.executable, __file__, '--child', n, iterate, hack], env=env,
                     capture_output=True, text=True)
  launches = [(m.group(1), float(m.group(2))) for m in re.finditer(
      r'launch\s+\d+ \S+\s+([\d.]+) us \(model\s+[\d.]+\)  box (\d+) x', p.stderr)]
  launches = [(float(a), b) for a, b in [(x[0], x[1]) for x in launches]]
  total = re.search(r'TOTAL ([\d.]+)', p.stdout)
  if not total:
    print(label, 'FAILED', p.stderr[-300:]); continue
  by_box = {int(box): us for us, box in launches}
  print('%-24s sweep %.1f us (10 warm-up sweeps); box 504 %.1f  448 %.1f  400 %.1f  344 %.1f  256 %.1f us' % (
      label, float(total.group(1)), *[by_box.get(b, 0) for b in (504, 448, 400, 344, 256)]), flush=True)
