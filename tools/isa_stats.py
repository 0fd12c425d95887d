#!/usr/bin/env python3
"""Static figures of generated kernels, no GPU needed: registers, spills, the
instruction mix of the steady-state loop and how much of its VALU stream is
back-to-back DEPENDENT (instruction i+1 reads what instruction i wrote: such a pair
cannot issue in consecutive slots of one wavefront; at two wavefronts per SIMD that
is what bounds a kernel whose cells were serialised on one accumulator).
usage: isa_stats.py app iterate kernel-name 'key=value,...' ['flags=-mllvm -x']..."""
import sys as _sys
if len(_sys.argv) > 1 and _sys.argv[1] in ('-h', '--help'):   # usage = the text above
  print(__doc__)
  _sys.exit(0)
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'soda-compiler_amd')]
from soda_hip import frontend
from soda_hip.codegen import kernel, spec as specmod

OBJDUMP = '/opt/rocm/lib/llvm/bin/llvm-objdump'
READELF = '/opt/rocm/lib/llvm/bin/llvm-readelf'


def parse_opts(text):
  flags = []
  opts = {}
  for kv in text.split(','):
    if not kv:
      continue
    k, v = kv.split('=', 1)
    if k == 'flags':
      flags = v.split()
    else:
      opts[k] = ([int(x) for x in v.split('/')] if '/' in v else
                 int(v) if v.lstrip('-').isdigit() else v)
  return opts, flags


def sample(app):
  p = os.path.join(ROOT, 'tests', 'samples', app + '.soda')
  return p if os.path.exists(p) else os.path.join(ROOT, 'tests', 'samples', 'extra', app + '.soda')


def kernel_body(path, name):
  text = subprocess.check_output([OBJDUMP, '-d', path]).decode()
  m = re.search(r'^[0-9a-f]+ <%s>:\n(.*?)(?=^[0-9a-f]+ <|\Z)' % re.escape(name), text,
                re.S | re.M)
  return [l.split('//')[0].strip() for l in m.group(1).splitlines() if l.strip()]


def regs(path, name):
  notes = subprocess.check_output([READELF, '--notes', path]).decode()
  block = ''
  for part in re.split(r'\n\s+- \.agpr_count:', notes):
    if re.search(r'\.name:\s+%s\n' % re.escape(name), part):
      block = part
  out = {}
  for key in ('vgpr_count', 'sgpr_count', 'vgpr_spill_count', 'group_segment_fixed_size',
              'private_segment_fixed_size'):
    mm = re.findall(r'\.%s:\s+(\d+)' % key, block)
    out[key] = int(mm[-1]) if mm else None
  return out


def dst_src(line):
  parts = line.replace(',', ' ').split()
  ops = [p for p in parts[1:] if re.match(r'^v(\d+|\[\d+:\d+\])$', p)]

  def expand(p):
    m = re.match(r'v\[(\d+):(\d+)\]', p)
    return set(range(int(m.group(1)), int(m.group(2)) + 1)) if m else {int(p[1:])}
  if not ops:
    return set(), set()
  src = set()
  for p in ops[1:]:
    src |= expand(p)
  return expand(ops[0]), src


def loop_stats(body):
  """The biggest backward-branch region = the steady-state loop."""
  valu = [l for l in body if l.startswith('v_')]
  dep = 0
  prev_dst = set()
  for l in body:
    if l.startswith('v_'):
      d, s = dst_src(l)
      if prev_dst & s:
        dep += 1
      prev_dst = d
    elif l.startswith('s_nop'):
      pass
    else:
      prev_dst = set()
  hist = {}
  for l in body:
    op = l.split()[0]
    hist[op] = hist.get(op, 0) + 1
  return len(valu), dep, hist


def main():
  app, iterate, name = sys.argv[1], int(sys.argv[2]), sys.argv[3]
  st = frontend.load(sample(app), iterate=iterate)
  spec = specmod.spec_from_stencil(st)
  for variant in sys.argv[4:] or ['']:
    opts, flags = parse_opts(variant)
    try:
      text, table = kernel.generate(spec, **opts)
    except Exception as e:
      print(variant, 'GENERATE FAILED', e)
      continue
    with tempfile.NamedTemporaryFile(suffix='.hsaco', delete=False) as f:
      path = f.name
    try:
      kernel.compile_to_code_object(text, path, extra_flags=flags)
      r = regs(path, name)
      body = kernel_body(path, name)
      nvalu, dep, hist = loop_stats(body)
      top = sorted(hist.items(), key=lambda kv: -kv[1])[:8]
      print('%-50s vgpr %s spill %s lds %s scratch %s | VALU %d dependent-pairs %d (%.0f%%) '
            's_nop %d | %s' % (variant or '(default)', r['vgpr_count'], r['vgpr_spill_count'],
                               r['group_segment_fixed_size'], r['private_segment_fixed_size'],
                               nvalu, dep, 100.0 * dep / max(1, nvalu), hist.get('s_nop', 0),
                               ' '.join('%s:%d' % kv for kv in top)), flush=True)
      if os.environ.get('KEEP_ASM'):
        with open(os.environ['KEEP_ASM'], 'w') as f:
          f.write('\n'.join(body))
    finally:
      os.unlink(path)


if __name__ == '__main__':
  main()
