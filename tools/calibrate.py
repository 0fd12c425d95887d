#!/usr/bin/env python3
"""Measures, on the GPU box, the step times the run-time scheduler prices launches
with (include/soda_hip.h: soda_hip_kernel.step_ns_full / step_ns_one / stream_gbps)
for every fused streaming kernel of the sample programs, and writes them to
soda_hip/codegen/calibration.json (keyed by program + kernel shape:
kernel.calibration_key).  kernel.generate() puts the figures into the blob metadata;
kernels without an entry are priced by the model (step_valu / step_bytes).

Per kernel three single launches, each `depth` iterations of a blob that holds only
that deep kernel:
  * cache-resident array, chip full          -> step_ns_full
  * cache-resident array, <= 1 workgroup/CU   -> step_ns_one  (forced long chunks)
  * array far beyond the Infinity Cache       -> stream_gbps
  * an array in between (about twice the cache) -> fade_lo_mib / fade_hi_mib: the
    footprints between which the HBM term of the price fades in (soda_hip.cpp:
    step_seconds), placed so that the model reproduces this launch
  * shallow kernels (the ones HBM bounds): the streaming launch again under a few chunk
    lengths and caps on workgroups per CU -> stream_chunk (+ stream_wgs_per_cu) when one
    of them beats the launcher's own chunk by 3 % or more; stream_gbps is then the rate
    of THAT launch
step = launch time / (rounds x (chunk + fill rows)), from the library's own launch
trace (SODA_HIP_LAUNCH_TRACE).

usage: calibrate.py [app ...]      (default: every sample with deep kernels)"""
import sys as _sys
if len(_sys.argv) > 1 and _sys.argv[1] in ('-h', '--help'):   # usage = the text above
  print(__doc__)
  _sys.exit(0)
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIZES = {   # (cache-resident, streaming, in between) extents per dimension
    2: (3072, 16384, 8192),
    3: (256, 512, 384),
}
FADE_DEFAULT = (128, 512)      # MiB; soda_hip.cpp: kCacheResidentMiB, kStreamingMiB
# chunk lengths tried for the streaming launch of kernels up to this depth
STREAM_CHUNKS = (8, 12, 16, 24, 32, 48, 64, 96)
STREAM_SWEEP_MAX_DEPTH = {2: 4, 3: 2}


def fade_footprints(spec, mid_extent, cached_ns, mid_ns, stream_ns):
  """(fade_lo_mib, fade_hi_mib) from the step time on the in-between array: weight w of
  the HBM term there = how far its step has moved from the cache-resident one towards
  the streaming one; the fade is the line through (fade_lo, 0) and (footprint, w)."""
  from soda_hip.codegen import spec as specmod
  types = specmod.tensor_c_types(spec)
  cell = sum(specmod.ELEM_SIZE[t['c_type']] for t in spec['inputs']) + \
      sum(specmod.ELEM_SIZE[types[n]] for n in spec['outputs'])
  footprint = cell * float(mid_extent) ** spec['dim'] / 2.0 ** 20
  lo, hi = FADE_DEFAULT
  if stream_ns <= cached_ns * 1.05:      # never bound by HBM: nothing to place
    return lo, hi
  w = (mid_ns - cached_ns) / (stream_ns - cached_ns)
  if w <= 0.1:        # still cache-speed at this footprint: the fade starts here
    return int(footprint), int(max(2 * footprint, hi))
  if w >= 0.95:       # already streaming
    return int(min(lo, footprint / 2)), int(footprint)
  return lo, int(min(8192, max(footprint, lo + (footprint - lo) / w)))
APPS = ['jacobi2d', 'seidel2d', 'blur', 'sobel2d', 'jacobi3d', 'heat3d']
TRACE = re.compile(r'launch\s+\d+ (\S+)\s+([\d.]+) us \(model\s+[\d.]+\)  box (\d+) x (\d+) x (\d+)  '
                   r'grid (\d+) x (\d+) x (\d+)  chunk (\d+)  fill (\d+)  resident (\d+)'
                   r'  lds \d+  rounds (\d+)')


def child(app, depth, form, n, chunk, name=None):
  sys.path[:0] = [ROOT, os.path.join(ROOT, 'soda-compiler_amd')]
  import numpy as np
  from soda_hip import frontend
  from soda_hip.codegen import kernel, spec as specmod
  from soda_hip.runtime import host
  import __graft_entry__ as entry
  iterate = entry.BLOB_ITERATE.get(app)
  st = frontend.load(entry.sample_path(app), iterate=iterate)
  spec = specmod.spec_from_stencil(st)
  opts = dict(depths=[depth])
  if form:
    opts['deep3d'] = form
  text, table = kernel.generate(spec, **opts)
  mine = [k for k in table if k['kind'] == 'fused' and k['depth'] == depth and
          (not name or k['name'] == name)]
  if not mine:
    print('NOKERNEL')
    return
  path = '/tmp/calib_%d.hsaco' % os.getpid()
  kernel.compile_to_code_object(text, path)
  prog = host.open_program(blob=path, spec=spec)
  dims = [n] * spec['dim']
  shape = tuple(reversed(dims))
  dt = np.dtype(specmod.NUMPY_NAME[spec['inputs'][0]['c_type']])
  rng = np.random.default_rng(1)
  a = rng.random(shape, dtype=np.float32).astype(dt) if dt.kind == 'f' else \
      rng.integers(0, 65536, size=shape).astype(dt)
  din = host.DeviceArray(a.nbytes)
  din.upload(a)
  dout = host.DeviceArray(a.nbytes)
  dout.zero()
  lowered = specmod.inline_pointwise(spec)
  print('ENTRY ' + json.dumps(dict(key=kernel.calibration_key(mine[-1], lowered),
                                   entry=mine[-1])))
  # clocks settle only after tens of milliseconds of work: warm up that long first
  # (un-traced), then time five repeats
  os.environ.pop('SODA_HIP_LAUNCH_TRACE', None)
  probe = prog.sweep_timed([din.ptr], [dout.ptr], dims, depth, warmup=1, repeats=1)
  warm = int(min(2000, max(3, 80000.0 / max(1.0, probe['kernel_us']))))
  os.environ['SODA_HIP_LAUNCH_TRACE'] = '1'
  prog.sweep_timed([din.ptr], [dout.ptr], dims, depth, warmup=warm, repeats=5)
  prog.close()


def parse_launch(fields):
  name, us, bx, by, bz, gx, gy, gz, chunk_used, fill, resident, rounds = fields
  blocks = int(gx) * int(gy) * int(gz)
  # the rounds the LAUNCHER prices (under a cap on workgroups per CU more than blocks /
  # resident): step_ns and stream_gbps must reproduce this launch through ITS formula
  rounds = int(rounds)
  steps = rounds * (int(chunk_used) + int(fill))
  return dict(name=name, us=float(us), blocks=blocks, resident=int(resident), steps=steps,
              chunk=int(chunk_used),
              # (the scheduler adds 2 us per launch on top of the steps)
              step_ns=max(1.0, float(us) - 2.0) * 1e3 / steps)


def child_sweep(app, depth, form, n, name, settings):
  """One compile, then the streaming launch under every (chunk, cap) of `settings`
  ('chunk:cap,...'; cap '' = the kernel's own, -1 = none): a marker line and the
  library's launch trace per setting on stderr."""
  sys.path[:0] = [ROOT, os.path.join(ROOT, 'soda-compiler_amd')]
  import numpy as np
  from soda_hip import frontend
  from soda_hip.codegen import kernel, spec as specmod
  from soda_hip.runtime import host
  import __graft_entry__ as entry
  st = frontend.load(entry.sample_path(app), iterate=entry.BLOB_ITERATE.get(app))
  spec = specmod.spec_from_stencil(st)
  opts = dict(depths=[depth])
  if form:
    opts['deep3d'] = form
  text, table = kernel.generate(spec, **opts)
  path = '/tmp/calib_%d.hsaco' % os.getpid()
  kernel.compile_to_code_object(text, path)
  dims = [n] * spec['dim']
  shape = tuple(reversed(dims))
  dt = np.dtype(specmod.NUMPY_NAME[spec['inputs'][0]['c_type']])
  rng = np.random.default_rng(1)
  a = rng.random(shape, dtype=np.float32).astype(dt) if dt.kind == 'f' else \
      rng.integers(0, 65536, size=shape).astype(dt)
  din = host.DeviceArray(a.nbytes)
  din.upload(a)
  dout = host.DeviceArray(a.nbytes)
  dout.zero()
  os.environ['SODA_HIP_PREFER'] = name[name.index('_fused_'):]
  for setting in settings.split(','):
    chunk, cap = setting.split(':')
    os.environ['SODA_HIP_CHUNK_ROWS'] = chunk
    os.environ.pop('SODA_HIP_WGS_PER_CU', None)
    if cap:
      os.environ['SODA_HIP_WGS_PER_CU'] = cap
    prog = host.open_program(blob=path, spec=spec)
    os.environ.pop('SODA_HIP_LAUNCH_TRACE', None)
    probe = prog.sweep_timed([din.ptr], [dout.ptr], dims, depth, warmup=1, repeats=1)
    warm = int(min(500, max(3, 40000.0 / max(1.0, probe['kernel_us']))))
    sys.stderr.write('SETTING %s\n' % setting)
    sys.stderr.flush()
    os.environ['SODA_HIP_LAUNCH_TRACE'] = '1'
    prog.sweep_timed([din.ptr], [dout.ptr], dims, depth, warmup=warm, repeats=5)
    prog.close()


def measure_sweep(app, depth, form, n, name, settings):
  """{(chunk, cap): launch record} of the streaming launch under each setting."""
  env = dict(os.environ, SODA_HIP_TUNING='1')
  p = subprocess.run([sys.executable, __file__, '--sweep', app, str(depth), form or '-',
                      str(n), name, ','.join('%d:%s' % s for s in settings)], env=env,
                     capture_output=True, text=True)
  out = {}
  current = None
  for line in p.stderr.splitlines():
    if line.startswith('SETTING '):
      chunk, cap = line.split()[1].split(':')
      current = (int(chunk), cap)
    else:
      m = TRACE.search(line)
      if m and current is not None and m.group(1) == name:
        out[current] = parse_launch(m.groups())
        current = None
  return out


def measure(app, depth, form, n, chunk=0, name=None):
  env = dict(os.environ, SODA_HIP_TUNING='1', SODA_HIP_LAUNCH_TRACE='1')
  # (-1: the launcher's own chunk, whatever an earlier calibration says)
  env['SODA_HIP_CHUNK_ROWS'] = str(chunk if chunk else -1)
  if name:      # several kernels of this depth in the blob: this one, whatever its price
    env['SODA_HIP_PREFER'] = name[name.index('_fused_'):]
  p = subprocess.run([sys.executable, __file__, '--child', app, str(depth), form or '-',
                      str(n), str(chunk), name or '-'], env=env, capture_output=True,
                     text=True)
  m = re.search(r'ENTRY (.*)', p.stdout)
  launches = TRACE.findall(p.stderr)
  if not m or not launches:
    return None
  info = json.loads(m.group(1))
  return dict(info, **parse_launch(launches[0]))


def main():
  apps = sys.argv[1:] or APPS
  sys.path[:0] = [ROOT, os.path.join(ROOT, 'soda-compiler_amd')]
  from soda_hip import frontend
  from soda_hip.codegen import kernel, spec as specmod
  import __graft_entry__ as entry
  path = kernel.CALIBRATION_FILE
  try:
    table = json.load(open(path))
  except (OSError, ValueError):
    table = dict(kernels={})
  table['note'] = ('measured by tools/calibrate.py on one MI355X; keyed by program '
                   'hash / kernel name / shape digest (kernel.calibration_key)')
  for app in apps:
    st = frontend.load(entry.sample_path(app), iterate=entry.BLOB_ITERATE.get(app))
    spec = specmod.spec_from_stencil(st)
    _, kernels = kernel.generate(spec)
    dim = spec['dim']
    small, big, between = SIZES[dim]
    for k in kernels:
      if k['kind'] != 'fused' or not k.get('fill_rows'):
        continue
      form = None
      if dim == 3 and k['depth'] > 2:
        form = 'blk' if k.get('stack') else 'wp'
      full = measure(app, k['depth'], form, small, name=k['name'])
      if not full or full['name'] != k['name']:
        print('%-28s not measured' % k['name'])
        continue
      # <= one workgroup per CU: chunks long enough that the grid is below 256
      inner = max(1, full['blocks'] * full['chunk'] // max(1, small))
      long_chunk = min(small, -(-small * inner // 240 // 4) * 4 + 4)
      one = measure(app, k['depth'], form, small, chunk=max(8, long_chunk), name=k['name'])
      stream = measure(app, k['depth'], form, big, name=k['name'])
      mid = measure(app, k['depth'], form, between, name=k['name'])
      if not one or not stream or not mid:
        print('%-28s partly measured' % k['name'])
        continue
      chosen = None
      if k['depth'] <= STREAM_SWEEP_MAX_DEPTH.get(dim, 0):
        own_cap = int(k.get('stream_wgs_per_cu', 0))
        caps = ['', '2'] if own_cap != 2 else ['', '-1']
        settings = [(c, cap) for cap in caps for c in STREAM_CHUNKS]
        sweep = measure_sweep(app, k['depth'], form, big, k['name'], settings)
        if sweep:
          (c, cap), rec = min(sweep.items(), key=lambda kv: kv[1]['us'])
          print('%-28s streaming launch: own chunk %d %.1f us; best of %d settings: chunk %d '
                'cap %s %.1f us' % (k['name'], stream['chunk'], stream['us'], len(sweep), c,
                                    cap or 'own', rec['us']), flush=True)
          if rec['us'] < 0.97 * stream['us']:
            # (the cap in effect, spelled out: "the kernel's own" may itself come from
            # an earlier calibration, which this record replaces)
            chosen = dict(stream_chunk=c,
                          stream_wgs_per_cu=own_cap if not cap else max(0, int(cap)))
            stream = dict(stream, **rec)
      active = min(stream['blocks'], stream['resident'])
      gbps = active * k['step_bytes'] / stream['step_ns']      # bytes / ns = GB/s
      # the step on the in-between array, at the occupancy it ran at, against the cached
      # and the streaming step at that occupancy
      share = min(1.0, mid['blocks'] / max(1.0, mid['resident']))
      cached_ns = one['step_ns'] + (full['step_ns'] - one['step_ns']) * share
      fade_lo, fade_hi = fade_footprints(
          specmod.inline_pointwise(spec), between, cached_ns, mid['step_ns'],
          max(cached_ns, min(mid['blocks'], mid['resident']) * k['step_bytes'] / gbps))
      table['kernels'][full['key']] = dict(
          name=k['name'], app=app, step_ns_full=int(round(full['step_ns'])),
          step_ns_one=int(round(one['step_ns'])), stream_gbps=int(round(gbps)),
          fade_lo_mib=fade_lo, fade_hi_mib=fade_hi, **(chosen or {}),
          measured=dict(full=[small, full['blocks'], full['steps'], full['us']],
                        one=[small, one['blocks'], one['steps'], one['us']],
                        stream=[big, stream['blocks'], stream['steps'], stream['us']],
                        between=[between, mid['blocks'], mid['steps'], mid['us']]))
      if gbps > 8000 and not (chosen or {}).get('stream_chunk'):
        # only a short measured chunk (loads past its last row skipped, no stores during
        # fill) may price above the HBM peak: the figure is a price constant valid at that
        # chunk length (include/soda_hip.h); anything else is a measurement gone wrong
        print('WARNING: %s: stream_gbps %.0f above the HBM peak without a measured chunk'
              % (k['name'], gbps), flush=True)
      print('%-28s full %6.0f ns/step (%4d wgs)  one %6.0f ns/step (%4d wgs)  stream %5.0f '
            'GB/s of step_bytes (%4d wgs, %.0f ns/step)  at %d: %.0f ns/step -> fade %d..%d MiB'
            % (k['name'], full['step_ns'], full['blocks'], one['step_ns'], one['blocks'],
               gbps, stream['blocks'], stream['step_ns'], between, mid['step_ns'], fade_lo,
               fade_hi), flush=True)
  # entries of the calibrated apps whose kernel shape no longer exists are stale
  live = set()
  for app in apps:
    st = frontend.load(entry.sample_path(app), iterate=entry.BLOB_ITERATE.get(app))
    spec = specmod.spec_from_stencil(st)
    for k in kernel.generate(spec)[1]:
      if k['kind'] == 'fused':
        live.add(kernel.calibration_key(k, specmod.inline_pointwise(spec)))
  for key in [key for key, v in table['kernels'].items()
              if v.get('app') in apps and key not in live]:
    del table['kernels'][key]
  with open(path, 'w') as f:
    json.dump(table, f, indent=1, sort_keys=True)
  print('wrote', path)


if __name__ == '__main__':
  if len(sys.argv) > 1 and sys.argv[1] == '--child':
    child(sys.argv[2], int(sys.argv[3]), None if sys.argv[4] == '-' else sys.argv[4],
          int(sys.argv[5]), int(sys.argv[6]),
          None if len(sys.argv) < 8 or sys.argv[7] == '-' else sys.argv[7])
  elif len(sys.argv) > 1 and sys.argv[1] == '--sweep':
    child_sweep(sys.argv[2], int(sys.argv[3]), None if sys.argv[4] == '-' else sys.argv[4],
                int(sys.argv[5]), sys.argv[6], sys.argv[7])
  else:
    main()
