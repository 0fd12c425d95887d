#!/usr/bin/env python3
"""Per-launch times of one sweep under several kernel sets (e.g. both depth-4 forms,
the block form only, the wave-pipelined form only): how often the scheduler's cost
model picks the faster kernel for a box, and what the best per-launch choice would
give.  Uses the library's own per-launch events (SODA_HIP_LAUNCH_TRACE).
usage: launch_compare.py app N iterate 'key=value,...' 'key=value,...' ..."""
import sys as _sys
if len(_sys.argv) > 1 and _sys.argv[1] in ('-h', '--help'):   # usage = the text above
  print(__doc__)
  _sys.exit(0)
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == '--child':
  sys.path[:0] = [ROOT, os.path.join(ROOT, 'soda-compiler_amd')]
  import numpy as np
  from soda_hip import frontend
  from soda_hip.codegen import kernel, spec as specmod
  from soda_hip.runtime import host
  app, n, iterate, variant = sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
  st = frontend.load(os.path.join(ROOT, 'tests', 'samples', app + '.soda'), iterate=iterate)
  spec = specmod.spec_from_stencil(st)
  opts = {k: ([int(x) for x in v.split('/')] if '/' in v else
              int(v) if v.lstrip('-').isdigit() else v)
          for k, v in (kv.split('=', 1) for kv in variant.split(',') if kv)}
  # flags=-mllvm:-amdgpu-sched-strategy=iterative-ilp  -> extra hipcc flags
  flags = opts.pop('flags', '').split(':') if opts.get('flags') else []
  text, _ = kernel.generate(spec, **opts)
  path = '/tmp/lc_%d.hsaco' % os.getpid()
  kernel.compile_to_code_object(text, path, extra_flags=flags)
  prog = host.open_program(blob=path, spec=spec)
  shape = (n,) * spec['dim']
  if os.environ.get('LC_WARM'):
    pass
  a = np.random.default_rng(1).random(shape, dtype=np.float32)
  din = host.DeviceArray(a.nbytes); din.upload(a)
  dout = host.DeviceArray(a.nbytes); dout.zero()
  t = prog.sweep_timed([din.ptr], [dout.ptr], list(shape), iterate,
                       warmup=int(os.environ.get('LC_WARMUP', '2')), repeats=3)
  print('TOTAL %.1f' % t['kernel_us'])
  sys.exit(0)

app, n, iterate = sys.argv[1:4]
runs = []
for variant in sys.argv[4:]:
  env = dict(os.environ, SODA_HIP_TUNING='1', SODA_HIP_LAUNCH_TRACE='1')
  # `env:NAME=VALUE` entries of a variant are environment variables of ITS run
  # (e.g. env:SODA_HIP_PREFER=k4b), the rest are generator options
  parts = [kv for kv in variant.split(',') if kv]
  env.update(kv[4:].split('=', 1) for kv in parts if kv.startswith('env:'))
  variant = ','.join(kv for kv in parts if not kv.startswith('env:'))
  p = subprocess.run([sys.executable, __file__, '--child', app, n, iterate, variant],
                     env=env, capture_output=True, text=True)
  launches = [(m.group(2), float(m.group(3)), float(m.group(4)), m.group(5),
               '%sx%s' % (m.group(6), m.group(7)))
              for m in re.finditer(r'launch\s+(\d+) (\S+)\s+([\d.]+) us \(model\s+([\d.]+)\)  box (\S+ x \S+ x \S+)'
                                   r'\s+grid (\d+) x \d+ x \d+\s+chunk (\d+)', p.stderr)]
  total = re.search(r'TOTAL ([\d.]+)', p.stdout)
  if not launches:
    print(variant, 'FAILED', p.stderr[-400:])
    continue
  label = ','.join(parts)
  runs.append((label, launches, float(total.group(1))))
  print('%-40s %d launches  sum of fastest %.1f us  sweep %.1f us' % (
      label, len(launches), sum(l[1] for l in launches), float(total.group(1))))
if runs and len({len(r[1]) for r in runs}) == 1:
  print('%3s  %-18s' % ('#', 'box') + ''.join('  %-32s' % r[0][:32] for r in runs))
  best_sum = 0.0
  for i in range(len(runs[0][1])):
    cells = [r[1][i] for r in runs]
    best_sum += min(c[1] for c in cells[1:]) if len(cells) > 1 else cells[0][1]
    print('%3d  %-18s' % (i, cells[0][3]) + ''.join(
        '  %-5s %7.1f us (model %6.1f) %-9s' % (c[0].split('_fused_')[-1], c[1], c[2], c[4])
        for c in cells))
  print('best single-kernel choice per launch (columns 2..): %.1f us' % best_sum)
