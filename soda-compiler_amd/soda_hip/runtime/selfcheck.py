"""The self-check half of `<app>_test`: recompute on the CPU, compare, count.

The reference's generated `<app>_test` ends by re-running the stencil as plain
loops on the host and comparing the device result against it (reference
host.py:1073-1158).  That check is part of the drop-in surface (the function
returns the mismatch count and prints `INFO: PASS!` / `INFO: FAIL!`), so the HIP
back end keeps it.  It is NOT a compute path: nothing here produces a value that
is handed back to a caller, the device result is only read.

The loops are the ones `sodac --hip-host-cpp` emits for C callers
(`codegen/host_cpp.print_selfcheck`); here they are built into a small shared
object with the host compiler on first use, the way the reference's host program
is built by the user (README.md:93-96).
"""
import ctypes
import hashlib
import io
import os
import subprocess
import tempfile

from ..codegen import host_cpp


def generate_cpp(spec):
  buf = io.StringIO()
  host_cpp.print_prologue(buf)
  host_cpp.print_selfcheck(spec, buf)
  return buf.getvalue()


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
  dims = list(reversed(inputs[0].shape))
  fn = getattr(lib, spec['app_name'] + '_selfcheck')
  fn.restype = ctypes.c_longlong
  fn.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p),
                 ctypes.POINTER(ctypes.c_int), ctypes.c_int, ctypes.c_double,
                 ctypes.c_int]
  pin = (ctypes.c_void_p * len(inputs))(*[a.ctypes.data for a in inputs])
  pout = (ctypes.c_void_p * len(device_outputs))(
      *[a.ctypes.data for a in device_outputs])
  cdims = (ctypes.c_int * 4)(*(dims + [0] * (4 - len(dims))))
  return fn(pin, pout, cdims, iterate, threshold, max_report)
