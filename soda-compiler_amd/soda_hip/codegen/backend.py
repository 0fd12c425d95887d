"""The sodac plug-in surface of the HIP back end.

A back end of the reference driver is a module with two functions,
`add_arguments(parser)` and `print_code(stencil, args)` (reference
src/soda/codegen/xilinx/opencl.py:11, :33; registered at src/sodac:66 and :127).
This module has the same two, with `--hip-*` flags where the Xilinx one has
`--xocl-*`:

  --hip-kernel FILE   HIP kernel text              (cf. --xocl-kernel)
  --hip-host FILE     Python host shim             (cf. --xocl-host)
  --hip-header FILE   C header                     (cf. --xocl-header)
  --hip-host-cpp FILE C++ host defining `<app>` / `<app>_test` for C callers
                      on top of libsoda_hip.so     (cf. --xocl-host)
  --hip-blob FILE     gfx950 code object, built by running hipcc on the kernel
                      text (cf. --xocl-hw-xo, which runs the Vivado tools)
  --hip [DIR]         all of the above under default names (cf. --xocl)

`-` writes to stdout.  `stencil` may be this project's `frontend.Stencil` or the
reference's `soda.core.Stencil`.
"""
import logging
import os
import sys

from . import header, host_cpp, host_shim, kernel
from . import spec as specmod

_logger = logging.getLogger().getChild(__name__)


def add_arguments(parser):
  parser.add_argument('--hip', type=str, dest='hip_output_dir', metavar='dir',
                      nargs='?', const='',
                      help='directory to generate kernel, host, header and blob')
  parser.add_argument('--hip-kernel', type=str, dest='hip_kernel_file',
                      metavar='file', help='HIP kernel code for gfx950')
  parser.add_argument('--hip-host', type=str, dest='hip_host_file',
                      metavar='file', help='Python host shim')
  parser.add_argument('--hip-host-cpp', type=str, dest='hip_host_cpp_file',
                      metavar='file', help='C++ host program over libsoda_hip.so')
  parser.add_argument('--hip-header', type=str, dest='hip_header_file',
                      metavar='file', help='C header of the entry points')
  parser.add_argument('--hip-blob', type=str, dest='hip_blob_file',
                      metavar='file', help='gfx950 code object (runs hipcc)')
  parser.add_argument('--hip-max-depth', type=int, dest='hip_max_depth',
                      metavar='K', help='deepest fused (time-tiled) kernel to '
                      'emit, in iterations per launch, never deeper than the '
                      'program iterates (default %d; with the default, programs '
                      'the packed wave-pipelined form covers also get depths %s)'
                      % (kernel.DEFAULT_MAX_DEPTH,
                         ', '.join(map(str, kernel.PACKED_DEEP_DEPTHS))))
  parser.add_argument('--hip-cols', type=int, dest='hip_cols', metavar='C',
                      help='columns per lane (default: 16-byte vectors, capped '
                      'by burst width)')
  parser.add_argument('--hip-chunk-rows', type=int, dest='hip_chunk_rows',
                      metavar='ROWS', help='outer-dimension rows per workgroup')


def to_spec(stencil):
  if isinstance(stencil, dict):
    return stencil
  if hasattr(stencil, 'chronological_tensors'):   # the reference's Stencil
    return specmod.spec_from_reference_stencil(stencil)
  return specmod.spec_from_stencil(stencil)


def _open(path):
  if path == '-':
    return sys.stdout, False
  return open(path, 'w'), True


IGNORED_OVERRIDES = (('unroll_factor', '--unroll-factor'), ('tile_size', '--tile-size'),
                     ('dram_in', '--dram-in'), ('dram_out', '--dram-out'))


def warn_ignored_overrides(args):
  """The reference's FPGA knobs are accepted (same command lines keep working) and
  carried in the kernel metadata, but no HIP generator reads them: they change how
  the FPGA computes, never what (the reference's CPU loops, host.py:1076-1117, do
  not mention them; unroll factor = PEs per stage, hls_kernel.py:157-160; tile size
  = line-buffer length, core.py:612-635).  Here the cells per lane follow from
  `burst width` (--hip-cols) and tiles / chunks are chosen per launch by the
  run-time.  Said once, at warning level, when a flag is given explicitly."""
  given = [flag for attr, flag in IGNORED_OVERRIDES
           if getattr(args, attr, None) is not None]
  if given:
    _logger.warning(
        'the HIP back end ignores %s: FPGA micro-architecture knobs (results never '
        'depend on them); use --hip-cols / --hip-max-depth / --hip-chunk-rows to '
        'steer the GPU kernels', ', '.join(given))
  return given


def print_code(stencil, args):
  spec = to_spec(stencil)
  warn_ignored_overrides(args)
  max_depth = getattr(args, 'hip_max_depth', None)
  files = dict(kernel=getattr(args, 'hip_kernel_file', None),
               host=getattr(args, 'hip_host_file', None),
               header=getattr(args, 'hip_header_file', None),
               host_cpp=getattr(args, 'hip_host_cpp_file', None),
               blob=getattr(args, 'hip_blob_file', None))
  out_dir = getattr(args, 'hip_output_dir', None)
  if out_dir is not None:
    if out_dir and not os.path.isdir(out_dir):
      os.makedirs(out_dir)
    app = spec['app_name']
    defaults = dict(kernel='%s_kernel.hip' % app, host='%s.py' % app,
                    header='%s.h' % app, blob='%s.hsaco' % app,
                    host_cpp='%s_host.cpp' % app)
    for key, name in defaults.items():
      if files[key] is None:
        files[key] = os.path.join(out_dir, name)
  if not any(files.values()):
    return
  text = table = None
  if files['kernel'] or files['blob'] or files['host_cpp']:
    text, table = kernel.generate(
        spec, max_depth=max_depth, cols=getattr(args, 'hip_cols', None),
        chunk_rows=getattr(args, 'hip_chunk_rows', None))
  if files['kernel']:
    _logger.info('generate HIP kernel code as %s', files['kernel'])
    f, close = _open(files['kernel'])
    f.write(text)
    if close:
      f.close()
  if files['host']:
    _logger.info('generate host shim as %s', files['host'])
    f, close = _open(files['host'])
    host_shim.print_code(spec, f)
    if close:
      f.close()
  if files['header']:
    _logger.info('generate header as %s', files['header'])
    f, close = _open(files['header'])
    header.print_code(spec, f)
    if close:
      f.close()
  if files['host_cpp']:
    _logger.info('generate C++ host as %s', files['host_cpp'])
    f, close = _open(files['host_cpp'])
    host_cpp.print_code(spec, table, f, lowered=specmod.inline_pointwise(spec))
    if close:
      f.close()
  if files['blob']:
    _logger.info('build code object %s', files['blob'])
    kernel.compile_to_code_object(text, files['blob'])
