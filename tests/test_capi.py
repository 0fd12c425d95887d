"""CPU-only checks of the drop-in boundary: libsoda_hip.so loads, exports every
symbol include/soda_hip.h declares, and its GPU-free entry points behave."""
import ctypes
import os
import re
import subprocess

import pytest

from soda_hip.runtime import capi, host

from conftest import ROOT

HEADER = os.path.join(ROOT, 'include', 'soda_hip.h')


@pytest.fixture(scope='module')
def lib():
  subprocess.check_call(['make', '-s', '-C',
                         os.path.join(ROOT, 'soda-compiler_amd', 'csrc')])
  return capi.lib()


def declared_functions():
  text = open(HEADER).read()
  text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
  return sorted(set(re.findall(r'\b(soda_hip_[a-z0-9_]+)\s*\(', text)))


def test_every_declared_symbol_is_exported_and_bound(lib):
  names = declared_functions()
  assert len(names) >= 25
  for name in names:
    assert hasattr(lib, name), name
  assert sorted(capi.SIGNATURES) == names


def test_abi_version_and_error_names(lib):
  assert lib.soda_hip_abi_version() == capi.ABI_VERSION
  text = open(HEADER).read()
  assert int(re.search(r'#define SODA_HIP_ABI_VERSION (\d+)', text).group(1)) == \
      capi.ABI_VERSION
  # Halide numbering of reference host.py:118-133
  assert lib.soda_hip_error_name(-3) == b'bad_elem_size'
  assert lib.soda_hip_error_name(-16) == b'device_malloc_failed'
  assert lib.soda_hip_error_name(-23) == b'device_run_failed'
  assert lib.soda_hip_error_name(0) == b'ok'


def test_struct_layouts_match_the_header(lib, tmp_path):
  # soda_hip_args is what the generated kernels receive by value
  assert ctypes.sizeof(capi.BufferT) == 72          # legacy Halide buffer_t
  assert ctypes.sizeof(capi.KernelDesc) == 96 + 4 * (3 + 3 + 4 + 1 + 3 + 2 + 3 + 1 + 3 + 1 + 1)
  assert ctypes.sizeof(capi.Window) == 4 * (2 + 4 + 4)
  assert ctypes.sizeof(capi.ProgramDesc) == 4 * (4 + 16 + 8 + 1) + 64 * 40
  # ... and against the C compiler's view of include/soda_hip.h: size of every struct
  # the binding mirrors and the offset of every field of the kernel descriptor
  import subprocess
  fields = [name for name, _ in capi.KernelDesc._fields_]
  src = tmp_path / 'layout.c'
  src.write_text(
      '#include <stdio.h>\n#include <stddef.h>\n#include "soda_hip.h"\n'
      'int main(void) {\n'
      '  printf("%zu %zu %zu %zu %zu\\n", sizeof(soda_hip_kernel), sizeof(soda_hip_window),\n'
      '         sizeof(soda_hip_program), sizeof(soda_hip_slab), sizeof(soda_hip_timing));\n' +
      ''.join('  printf("%%zu\\n", offsetof(soda_hip_kernel, %s));\n' % f for f in fields) +
      ''.join('  printf("%%zu\\n", offsetof(soda_hip_slab, %s));\n' % f
              for f, _ in capi.Slab._fields_) +
      '  return 0;\n}\n')
  exe = tmp_path / 'layout'
  subprocess.check_call(['gcc', '-I', os.path.join(ROOT, 'include'), str(src), '-o', str(exe)])
  out = subprocess.check_output([str(exe)], text=True).split()
  assert [int(v) for v in out[:5]] == [ctypes.sizeof(t) for t in (
      capi.KernelDesc, capi.Window, capi.ProgramDesc, capi.Slab, capi.Timing)]
  assert [int(v) for v in out[5:5 + len(fields)]] == [getattr(capi.KernelDesc, f).offset
                                                      for f in fields]
  # the slab descriptor grew in ABI 8 (cut, abort_on_error): every field where C has it
  assert [int(v) for v in out[5 + len(fields):]] == [getattr(capi.Slab, f).offset
                                                     for f, _ in capi.Slab._fields_]
  assert [f for f, _ in capi.Slab._fields_][-4:] == ['order', 'cut', 'abort_on_error',
                                                     'reserved']


def test_null_arguments_are_errors_not_crashes(lib):
  assert lib.soda_hip_device_count(None) == -12
  assert b'NULL' in lib.soda_hip_last_error()
  assert lib.soda_hip_module_load_file(None, None) == -12
  assert lib.soda_hip_plan_margins(None, 1, None, None) == -12
  assert lib.soda_hip_plan_destroy(None) == 0
  assert lib.soda_hip_module_unload(None) == 0


def test_c_and_python_slab_drivers_agree_on_the_exchange_period(lib):
  """soda_hip_slab_exchange (what the generated `<app>_multi_gpu` uses) and
  soda_hip.runtime.dist.SlabPlan clamp a too-deep ghost region to the same value and
  refuse the same cuts."""
  from soda_hip.runtime import dist as sdist
  for rows in (3, 20, 64, 100, 2048, 16384):
    for world in (1, 2, 3, 4, 8):
      for r_lo, r_hi in ((1, 1), (2, 1), (0, 2), (0, 0), (3, 3)):
        for wanted in (1, 5, 24, 100, 192):
          got = ctypes.c_int(-1)
          rc = lib.soda_hip_slab_exchange(rows, world, r_lo, r_hi, wanted,
                                          ctypes.byref(got))
          try:
            want = sdist.SlabPlan([64, rows], 0, world, r_lo, r_hi, wanted).exchange
          except ValueError:
            want = None
          if want is None:
            assert rc == -8 and b'thinner than the stencil reach' in lib.soda_hip_last_error()
          else:
            assert rc == 0 and got.value == want, (rows, world, r_lo, r_hi, wanted)


def test_no_gpu_means_loud_failure_not_fallback(lib):
  if host.device_count() > 0:
    pytest.skip('a GPU is present')
  h = ctypes.c_void_p()
  rc = lib.soda_hip_module_load_file(b'/nonexistent.hsaco', ctypes.byref(h))
  assert rc == -101
  with pytest.raises(capi.SodaHipError):
    host.Blob.from_source('extern "C" __global__ void k() {}')


def test_missing_library_raises(monkeypatch):
  monkeypatch.setenv('SODA_HIP_LIB', '/nonexistent/libsoda_hip.so')
  monkeypatch.setattr(capi, '_LIB', None)
  with pytest.raises(capi.SodaHipError) as e:
    capi.lib()
  assert 'no CPU fallback' in str(e.value).replace('\n', ' ') or \
      'not found' in str(e.value)


def test_product_never_imports_the_oracle():
  """The oracle is test infrastructure: nothing under soda-compiler_amd/ may
  import or execute it."""
  bad = []
  for dirpath, _, files in os.walk(os.path.join(ROOT, 'soda-compiler_amd')):
    for f in files:
      if f.endswith(('.py', '.cpp', '.h')) or f == 'sodac':
        text = open(os.path.join(dirpath, f), errors='replace').read()
        if re.search(r'^\s*(from|import)\s+oracle\b', text, re.M) or \
            'soda_oracle' in text:
          bad.append(os.path.join(dirpath, f))
  assert not bad, bad
