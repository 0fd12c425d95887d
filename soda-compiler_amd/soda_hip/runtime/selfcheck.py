"""The self-check half of `<app>_test`: recompute on the CPU, compare, count.

The reference's generated `<app>_test` ends by re-running the stencil as plain
loops on the host and comparing the device result against it (reference
host.py:1073-1158).  That check is part of the drop-in surface (the function
returns the mismatch count and prints `INFO: PASS!` / `INFO: FAIL!`), so the HIP
back end keeps it.  It is NOT a compute path: nothing here produces a value that
is handed back to a caller, the device result is only read.

The loops are emitted as C++ and built with the host compiler on first use, the
way the reference's host program is built by the user (README.md:93-96).
"""
import ctypes
import hashlib
import os
import subprocess
import tempfile

from ..codegen import spec as specmod

_COORD = 'pqrs'

# Same system headers as the reference's host translation unit
# (host.py:12-36): they decide which overload an unqualified `sqrt(float)` is.
_HEADERS = ('cassert cfloat cmath cstdbool cstddef cstdint cstdio cstdlib '
            'cstring algorithm array string unordered_map').split()


def generate_cpp(spec):
  dim = spec['dim']
  types = specmod.tensor_c_types(spec)
  ins = [t['name'] for t in spec['inputs']]
  outs = spec['outputs']
  o = ['#include <%s>' % h for h in _HEADERS]
  o.append('using std::array; using std::count; using std::fill; '
           'using std::string; using std::unordered_map;')
  o.append('extern "C" long long soda_selfcheck(void* const* inputs, '
           'void* const* device_outputs, const long long* dims, int iterate, '
           'const long long* bounds, double threshold, int max_report) {')
  o.append('  long long error_count = 0;')
  o.append('  size_t cells = 1;')
  o.append('  for (int d = 0; d < %d; ++d) cells *= (size_t)dims[d];' % dim)
  o.append('  const long long stride0 = 1; (void)stride0;')
  for d in range(1, dim):
    o.append('  const long long stride%d = %s;' % (
        d, ' * '.join('dims[%d]' % x for x in range(d))))
  # storage: two alternating sets for the tensors that feed the next iteration
  for s in spec['stages']:
    n = s['name']
    copies = 2 if n in outs else 1
    for k in range(copies):
      o.append('  %s* buf_%s_%d = new %s[cells]();' % (types[n], n, k, types[n]))
  o.append('  for (int it = 0; it < iterate; ++it) {')
  o.append('    const bool last = it == iterate - 1;')
  for j, n in enumerate(ins):
    if len(ins) == len(outs):
      # output j of the previous iteration feeds input j (core.py:342-360)
      o.append('    const %s* %s_img = it == 0 ? (const %s*)inputs[%d] : '
               '(((it - 1) & 1) ? buf_%s_1 : buf_%s_0);'
               % (types[n], n, types[n], j, outs[j], outs[j]))
    else:
      o.append('    const %s* %s_img = (const %s*)inputs[%d];'
               % (types[n], n, types[n], j))
  for s in spec['stages']:
    n = s['name']
    if n in outs:
      o.append('    %s* %s_img = (it & 1) ? buf_%s_1 : buf_%s_0;'
               % (types[n], n, n, n))
    else:
      o.append('    %s* %s_img = buf_%s_0;' % (types[n], n, n))

  def load(name, rel):
    return '%s_img[%s]' % (name, ' + '.join(
        '(%c%+d)*stride%d' % (_COORD[d], rel[d], d) for d in range(dim)))

  for si, s in enumerate(spec['stages']):
    n = s['name']
    o.append('    {  // produce %s' % n)
    o.append('      const long long* lo = bounds + (it * %d + %d) * %d;'
             % (len(spec['stages']), si, 2 * dim))
    o.append('      const long long* hi = lo + %d;' % dim)
    if n in outs:
      o.append('      const %s* fpga = (const %s*)device_outputs[%d];'
               % (types[n], types[n], outs.index(n)))
      o.append('#pragma omp parallel for reduction(+:error_count)')
    else:
      o.append('#pragma omp parallel for')
    for d in reversed(range(dim)):
      o.append('      for (long long {v} = lo[{d}]; {v} < hi[{d}]; ++{v})'.format(
          v=_COORD[d], d=d))
    o.append('      {')
    for let in s['lets']:
      o.append('        const %s %s = %s;' % (
          let['c_type'], let['name'], specmod.substitute_loads(let['expr'], load)))
    cell = ' + '.join('%c*stride%d' % (_COORD[d], d) for d in range(dim))
    o.append('        const %s result = %s;' % (
        types[n], specmod.substitute_loads(s['expr'], load)))
    o.append('        %s_img[%s] = result;' % (n, cell))
    if n in outs:
      coords = ', '.join('(int)%c' % _COORD[d] for d in range(dim))
      fmt = ', '.join(['%d'] * dim)
      o.append('        if (last) {')
      o.append('          const %s val_fpga = fpga[%s];' % (types[n], cell))
      o.append('          const %s val_cpu = result;' % types[n])
      if specmod.is_float_type(s['haoda_type']):
        # reference comparator: squared relative error (host.py:1124-1137)
        o.append('          if (double(val_fpga-val_cpu)*double(val_fpga-val_cpu)/'
                 '(double(val_cpu)*double(val_cpu)) > threshold * threshold) {')
        o.append('            if (error_count < max_report) fprintf(stderr, '
                 '"%%lf != %%lf @(%s)\\n", double(val_fpga), double(val_cpu), %s);'
                 % (fmt, coords))
      else:
        o.append('          if (val_fpga != val_cpu) {')
        o.append('            if (error_count < max_report) fprintf(stderr, '
                 '"%%ld != %%ld @(%s)\\n", (long)val_fpga, (long)val_cpu, %s);'
                 % (fmt, coords))
      o.append('            ++error_count;')
      o.append('          }')
      o.append('        }')
    o.append('      }')
    o.append('    }')
  o.append('  }')
  for s in spec['stages']:
    n = s['name']
    for k in range(2 if n in outs else 1):
      o.append('  delete[] buf_%s_%d;' % (n, k))
  o.append('  return error_count;')
  o.append('}')
  return '\n'.join(o) + '\n'


def _build(spec):
  src = generate_cpp(spec)
  tag = hashlib.sha1(src.encode()).hexdigest()[:16]
  cache = os.environ.get('SODA_HIP_CACHE') or os.path.join(
      tempfile.gettempdir(), 'soda_hip_cache_%d' % os.getuid())
  os.makedirs(cache, exist_ok=True)
  so = os.path.join(cache, 'selfcheck_%s_%s.so' % (spec['app_name'], tag))
  if not os.path.exists(so):
    cpp = so[:-3] + '.cpp'
    with open(cpp, 'w') as f:
      f.write(src)
    tmp = '%s.%d.tmp' % (so, os.getpid())
    subprocess.check_call(
        [os.environ.get('CXX', 'g++'), '-std=c++11', '-O2', '-fopenmp',
         '-ffp-contract=off', '-fPIC', '-shared', cpp, '-o', tmp])
    os.replace(tmp, so)
  return ctypes.CDLL(so)


def count_mismatches(spec, inputs, device_outputs, iterate, threshold=1e-5,
                     max_report=32):
  """Recomputes `iterate` iterations from `inputs` on the CPU and returns how
  many cells of `device_outputs` differ in the region the reference compares
  (its loop bounds, host.py:1082-1091)."""
  lib = _build(spec)
  dim = spec['dim']
  dims = tuple(reversed(inputs[0].shape))
  bounds = []
  for boxes in specmod.iteration_boxes(spec, iterate):
    for s in spec['stages']:
      lo, hi = boxes[s['name']]
      bounds += [-v for v in lo] + [dims[d] - hi[d] for d in range(dim)]
  fn = lib.soda_selfcheck
  fn.restype = ctypes.c_longlong
  fn.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p),
                 ctypes.POINTER(ctypes.c_longlong), ctypes.c_int,
                 ctypes.POINTER(ctypes.c_longlong), ctypes.c_double, ctypes.c_int]
  pin = (ctypes.c_void_p * len(inputs))(*[a.ctypes.data for a in inputs])
  pout = (ctypes.c_void_p * len(device_outputs))(
      *[a.ctypes.data for a in device_outputs])
  return fn(pin, pout, (ctypes.c_longlong * dim)(*dims), iterate,
            (ctypes.c_longlong * len(bounds))(*bounds), threshold, max_report)
