"""ctypes binding of libsoda_hip.so (include/soda_hip.h), nothing more.

There is no fallback: if the library cannot be loaded every entry point raises
`SodaHipError`; the stencil sweep runs on the GPU through this ABI or not at all.
"""
import ctypes
import os

MAX_DIMS = 4
MAX_TENSORS = 16
MAX_IO = 8
MAX_WINDOWS = 64
MAX_KERNELS = 32
ABI_VERSION = 8

KERNEL_STAGE = 0
KERNEL_FUSED = 1
ERR_NO_KERNEL = -102        # SODA_HIP_ERR_NO_KERNEL (include/soda_hip.h)


class SodaHipError(RuntimeError):
  def __init__(self, code, name, detail):
    self.code, self.name, self.detail = code, name, detail
    super().__init__('libsoda_hip: %s (%d): %s' % (name, code, detail))


class Window(ctypes.Structure):
  _fields_ = [('stage', ctypes.c_int32), ('parent', ctypes.c_int32),
              ('lo', ctypes.c_int32 * MAX_DIMS), ('hi', ctypes.c_int32 * MAX_DIMS)]


class ProgramDesc(ctypes.Structure):
  _fields_ = [('dim', ctypes.c_int32), ('n_inputs', ctypes.c_int32),
              ('n_stages', ctypes.c_int32), ('n_outputs', ctypes.c_int32),
              ('elem_size', ctypes.c_int32 * MAX_TENSORS),
              ('output_tensor', ctypes.c_int32 * MAX_IO),
              ('n_windows', ctypes.c_int32),
              ('window', Window * MAX_WINDOWS)]


class KernelDesc(ctypes.Structure):
  _fields_ = [('name', ctypes.c_char * 96), ('kind', ctypes.c_int32),
              ('depth', ctypes.c_int32), ('stage', ctypes.c_int32),
              ('block', ctypes.c_int32 * 3), ('tile', ctypes.c_int32 * MAX_DIMS),
              ('fill_rows', ctypes.c_int32), ('origin_align', ctypes.c_int32),
              ('min_extent', ctypes.c_int32 * 2),
              ('step_valu', ctypes.c_int32), ('step_bytes', ctypes.c_int32),
              ('step_ns_full', ctypes.c_int32), ('step_ns_one', ctypes.c_int32),
              ('stream_gbps', ctypes.c_int32),
              ('xcd_tiles', ctypes.c_int32),
              ('stream_wgs_per_cu', ctypes.c_int32),
              ('fade_lo_mib', ctypes.c_int32), ('fade_hi_mib', ctypes.c_int32),
              ('stream_chunk', ctypes.c_int32), ('edge_slack', ctypes.c_int32)]


class Slab(ctypes.Structure):
  _fields_ = [('rank', ctypes.c_int32), ('world', ctypes.c_int32),
              ('reach_lo', ctypes.c_int32), ('reach_hi', ctypes.c_int32),
              ('exchange', ctypes.c_int32),
              ('dims', ctypes.c_int64 * MAX_DIMS),
              ('own_first', ctypes.c_int64), ('own_last', ctypes.c_int64),
              ('order', ctypes.c_int32), ('cut', ctypes.c_int32),
              ('abort_on_error', ctypes.c_int32), ('reserved', ctypes.c_int32)]


SLAB_SERIAL, SLAB_BANDS_FIRST = 0, 1
SLAB_CUT_STATIC, SLAB_CUT_RECUT = 0, 1


class Timing(ctypes.Structure):
  _fields_ = [('kernel_us', ctypes.c_double), ('launches', ctypes.c_int32),
              ('max_depth', ctypes.c_int32), ('dominant_us', ctypes.c_double),
              ('dominant_launches', ctypes.c_int32),
              ('dominant_name', ctypes.c_char * 96),
              ('fastest_us', ctypes.c_double)]


class BufferT(ctypes.Structure):
  """Legacy Halide buffer_t (reference header.py:36-48)."""
  _fields_ = [('dev', ctypes.c_uint64), ('host', ctypes.c_void_p),
              ('extent', ctypes.c_int32 * 4), ('stride', ctypes.c_int32 * 4),
              ('min', ctypes.c_int32 * 4), ('elem_size', ctypes.c_int32),
              ('host_dirty', ctypes.c_uint8), ('dev_dirty', ctypes.c_uint8),
              ('_padding', ctypes.c_uint8 * (10 - ctypes.sizeof(ctypes.c_void_p)))]


_VP = ctypes.c_void_p
_VPP = ctypes.POINTER(ctypes.c_void_p)
_I64P = ctypes.POINTER(ctypes.c_int64)
_I32P = ctypes.POINTER(ctypes.c_int32)

# name -> (restype, argtypes); every exported symbol of include/soda_hip.h
SIGNATURES = {
    'soda_hip_error_name': (ctypes.c_char_p, [ctypes.c_int]),
    'soda_hip_last_error': (ctypes.c_char_p, []),
    'soda_hip_abi_version': (ctypes.c_int, []),
    'soda_hip_device_count': (ctypes.c_int, [ctypes.POINTER(ctypes.c_int)]),
    'soda_hip_set_device': (ctypes.c_int, [ctypes.c_int]),
    'soda_hip_device_info': (ctypes.c_int, [ctypes.c_int, ctypes.c_char_p,
                                            ctypes.c_size_t,
                                            ctypes.POINTER(ctypes.c_int),
                                            ctypes.POINTER(ctypes.c_uint64)]),
    'soda_hip_malloc': (ctypes.c_int, [_VPP, ctypes.c_size_t]),
    'soda_hip_free': (ctypes.c_int, [_VP]),
    'soda_hip_memset': (ctypes.c_int, [_VP, ctypes.c_int, ctypes.c_size_t, _VP]),
    'soda_hip_memcpy_h2d': (ctypes.c_int, [_VP, _VP, ctypes.c_size_t, _VP]),
    'soda_hip_memcpy_d2h': (ctypes.c_int, [_VP, _VP, ctypes.c_size_t, _VP]),
    'soda_hip_memcpy_d2d': (ctypes.c_int, [_VP, _VP, ctypes.c_size_t, _VP]),
    'soda_hip_stream_synchronize': (ctypes.c_int, [_VP]),
    'soda_hip_module_load_file': (ctypes.c_int, [ctypes.c_char_p, _VPP]),
    'soda_hip_module_load_data': (ctypes.c_int, [_VP, ctypes.c_size_t, _VPP]),
    'soda_hip_module_compile': (ctypes.c_int, [ctypes.c_char_p, ctypes.c_char_p,
                                               ctypes.POINTER(ctypes.c_char_p),
                                               ctypes.c_int, _VPP]),
    'soda_hip_module_image': (ctypes.c_int, [_VP, _VPP,
                                             ctypes.POINTER(ctypes.c_size_t)]),
    'soda_hip_module_meta': (ctypes.c_int, [_VP, ctypes.c_char_p, ctypes.c_size_t,
                                            ctypes.POINTER(ctypes.c_size_t)]),
    'soda_hip_module_unload': (ctypes.c_int, [_VP]),
    'soda_hip_plan_create': (ctypes.c_int, [_VP, ctypes.POINTER(ProgramDesc),
                                            ctypes.POINTER(KernelDesc),
                                            ctypes.c_int, _VPP]),
    'soda_hip_plan_destroy': (ctypes.c_int, [_VP]),
    'soda_hip_plan_margins': (ctypes.c_int, [_VP, ctypes.c_int, _I32P, _I32P]),
    'soda_hip_plan_schedule': (ctypes.c_int, [_VP, _I64P, ctypes.c_int, _I32P, _I32P,
                                              _I32P, ctypes.POINTER(ctypes.c_double),
                                              ctypes.c_int,
                                              ctypes.POINTER(ctypes.c_int)]),
    'soda_hip_plan_set_max_depth': (ctypes.c_int, [_VP, ctypes.c_int]),
    'soda_hip_plan_set_out_final_only': (ctypes.c_int, [_VP, ctypes.c_int]),
    'soda_hip_plan_tune': (ctypes.c_int, [_VP, _VPP, _VPP, _I64P, ctypes.c_int, _I32P,
                                          _I32P, _VP]),
    'soda_hip_plan_set_split': (ctypes.c_int, [_VP, _I64P, ctypes.c_int, _I32P,
                                               ctypes.c_int]),
    'soda_hip_sweep': (ctypes.c_int, [_VP, _VPP, _VPP, _I64P, ctypes.c_int, _I32P,
                                      _I32P, _VP]),
    'soda_hip_sweep_timed': (ctypes.c_int, [_VP, _VPP, _VPP, _I64P, ctypes.c_int,
                                            ctypes.c_int, ctypes.c_int, _VP,
                                            ctypes.POINTER(Timing)]),
    'soda_hip_clock_probe_start': (ctypes.c_int, [_VP, ctypes.c_int]),
    'soda_hip_clock_probe_finish': (ctypes.c_int, [_VP, ctypes.POINTER(ctypes.c_double),
                                                   ctypes.POINTER(ctypes.c_double)]),
    'soda_hip_slab_exchange': (ctypes.c_int, [ctypes.c_int64, ctypes.c_int, ctypes.c_int,
                                              ctypes.c_int, ctypes.c_int,
                                              ctypes.POINTER(ctypes.c_int)]),
    'soda_hip_slab_extent': (ctypes.c_int, [_VP, ctypes.POINTER(Slab), _I64P, _I64P,
                                            _I64P]),
    'soda_hip_slab_layout': (ctypes.c_int, [_VP, ctypes.POINTER(Slab), ctypes.c_int, _I64P,
                                            _I64P, _I64P, _I64P, _I64P]),
    'soda_hip_plan_tuned_streams': (ctypes.c_int, [_VP, ctypes.POINTER(ctypes.c_int)]),
    'soda_hip_run_slab': (ctypes.c_int, [_VP, ctypes.POINTER(Slab), _VP, _VP, _VP, _VP,
                                         ctypes.c_int, _VP, _VPP,
                                         ctypes.POINTER(ctypes.c_int)]),
    'soda_hip_run_buffers': (ctypes.c_int, [_VP,
                                            ctypes.POINTER(ctypes.POINTER(BufferT)),
                                            ctypes.POINTER(ctypes.POINTER(BufferT)),
                                            ctypes.c_int, ctypes.POINTER(Timing)]),
}

_LIB = None


def library_path():
  env = os.environ.get('SODA_HIP_LIB')
  if env:
    return env
  here = os.path.dirname(os.path.abspath(__file__))
  return os.path.normpath(os.path.join(here, '..', '..', 'csrc', 'libsoda_hip.so'))


def lib():
  """Loads libsoda_hip.so once; raises if it is missing or has the wrong ABI."""
  global _LIB
  if _LIB is None:
    path = library_path()
    if not os.path.exists(path) and not os.environ.get('SODA_HIP_LIB'):
      # build on demand (the host compiler is enough: no device code inside)
      import subprocess
      try:
        subprocess.check_call(['make', '-s', '-C', os.path.dirname(path)])
      except (OSError, subprocess.CalledProcessError):
        pass
    if not os.path.exists(path):
      raise SodaHipError(-19, 'no_device_interface',
                         '%s not found: build it with `make -C soda-compiler_amd/'
                         'csrc` (or __graft_entry__.build()); there is no CPU '
                         'fallback' % path)
    handle = ctypes.CDLL(path)
    for name, (restype, argtypes) in SIGNATURES.items():
      fn = getattr(handle, name)   # AttributeError = symbol missing
      fn.restype, fn.argtypes = restype, argtypes
    if handle.soda_hip_abi_version() != ABI_VERSION:
      raise SodaHipError(-103, 'blob_program_mismatch',
                         'libsoda_hip.so has ABI %d, binding expects %d'
                         % (handle.soda_hip_abi_version(), ABI_VERSION))
    _LIB = handle
  return _LIB


def check(code):
  if code != 0:
    handle = lib()
    raise SodaHipError(code, handle.soda_hip_error_name(code).decode(),
                       handle.soda_hip_last_error().decode())
  return code
