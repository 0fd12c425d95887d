"""Host side of the generated program: what `sodac --hip-host` shims call.

Mirrors the three generated functions of the reference's host printer:

  `<app>_wrapped` / `<app>`   reference host.py:186-945  ->  `Program.run_buffers`
  `<app>_test`                reference host.py:984-1167 ->  `app_test`

plus the device-pointer interface (`Program.sweep`) that benchmarks and the
multi-GPU driver use.  All GPU work goes through libsoda_hip.so (capi.py).
"""
import ctypes
import json
import os
import sys

import numpy as np

from . import capi
from ..codegen import spec as specmod


class DeviceArray:
  """A device allocation owned through the C ABI."""

  def __init__(self, nbytes):
    self.nbytes = int(nbytes)
    ptr = ctypes.c_void_p()
    capi.check(capi.lib().soda_hip_malloc(ctypes.byref(ptr), self.nbytes))
    self.ptr = ptr.value

  def upload(self, array):
    array = np.ascontiguousarray(array)
    assert array.nbytes <= self.nbytes
    capi.check(capi.lib().soda_hip_memcpy_h2d(
        self.ptr, array.ctypes.data, array.nbytes, None))
    capi.check(capi.lib().soda_hip_stream_synchronize(None))

  def download(self, shape, dtype):
    out = np.empty(shape, dtype=dtype)
    assert out.nbytes <= self.nbytes
    capi.check(capi.lib().soda_hip_memcpy_d2h(
        out.ctypes.data, self.ptr, out.nbytes, None))
    capi.check(capi.lib().soda_hip_stream_synchronize(None))
    return out

  def zero(self):
    capi.check(capi.lib().soda_hip_memset(self.ptr, 0, self.nbytes, None))

  def free(self):
    if self.ptr:
      capi.check(capi.lib().soda_hip_free(self.ptr))
      self.ptr = None

  def __del__(self):
    try:
      self.free()
    except Exception:   # interpreter shutdown
      pass


def device_count():
  n = ctypes.c_int()
  try:
    capi.check(capi.lib().soda_hip_device_count(ctypes.byref(n)))
  except capi.SodaHipError:
    return 0
  return n.value


def device_info(ordinal=0):
  name = ctypes.create_string_buffer(128)
  cu = ctypes.c_int()
  mem = ctypes.c_uint64()
  capi.check(capi.lib().soda_hip_device_info(
      ordinal, name, len(name), ctypes.byref(cu), ctypes.byref(mem)))
  return dict(arch=name.value.decode(), compute_units=cu.value,
              memory_bytes=mem.value)


class Blob:
  """A loaded kernel blob (the reference's `xclbin` argument): either a gfx950
  code object file, or kernel text compiled on the spot with hiprtc."""

  def __init__(self, handle):
    self.handle = handle
    self._meta = None

  @classmethod
  def from_file(cls, path):
    with open(path, 'rb') as f:
      head = f.read(4)
    if head == b'\x7fELF' or head.startswith(b'__CL'):
      h = ctypes.c_void_p()
      capi.check(capi.lib().soda_hip_module_load_file(
          os.fsencode(path), ctypes.byref(h)))
      return cls(h)
    with open(path) as f:
      return cls.from_source(f.read())

  @classmethod
  def from_source(cls, text, arch=None, options=('-fno-slp-vectorize', '-fwrapv')):
    """`options` are part of the numerical contract (no SLP packing, wrapping
    signed overflow) together with the flags recorded in the kernel text; there
    is no environment override - experiments pass their own `options`."""
    from ..codegen.kernel import flags_from_text
    options = tuple(options) + tuple(flags_from_text(text))
    h = ctypes.c_void_p()
    opts = (ctypes.c_char_p * len(options))(*[o.encode() for o in options])
    capi.check(capi.lib().soda_hip_module_compile(
        text.encode(), arch.encode() if arch else None, opts, len(options),
        ctypes.byref(h)))
    return cls(h)

  @property
  def meta(self):
    if self._meta is None:
      n = ctypes.c_size_t()
      capi.check(capi.lib().soda_hip_module_meta(self.handle, None, 0,
                                                 ctypes.byref(n)))
      buf = ctypes.create_string_buffer(n.value + 1)
      capi.check(capi.lib().soda_hip_module_meta(self.handle, buf, len(buf),
                                                 ctypes.byref(n)))
      if not buf.value:
        raise capi.SodaHipError(-103, 'blob_program_mismatch',
                                'blob carries no soda_hip_meta symbol')
      self._meta = json.loads(buf.value.decode())
    return self._meta

  def image(self):
    ptr = ctypes.c_void_p()
    n = ctypes.c_size_t()
    capi.check(capi.lib().soda_hip_module_image(self.handle, ctypes.byref(ptr),
                                                ctypes.byref(n)))
    return ctypes.string_at(ptr.value, n.value)

  def unload(self):
    if self.handle:
      capi.check(capi.lib().soda_hip_module_unload(self.handle))
      self.handle = None


def program_desc(spec):
  """soda_hip_program from a program spec."""
  names = [t['name'] for t in spec['inputs']] + [s['name'] for s in spec['stages']]
  index = {n: i for i, n in enumerate(names)}
  types = specmod.tensor_c_types(spec)
  desc = capi.ProgramDesc()
  desc.dim = spec['dim']
  desc.n_inputs = len(spec['inputs'])
  desc.n_stages = len(spec['stages'])
  desc.n_outputs = len(spec['outputs'])
  if len(names) > capi.MAX_TENSORS:
    raise ValueError('program has %d tensors, libsoda_hip handles %d'
                     % (len(names), capi.MAX_TENSORS))
  for n, i in index.items():
    desc.elem_size[i] = specmod.ELEM_SIZE[types[n]]
  for j, o in enumerate(spec['outputs']):
    desc.output_tensor[j] = index[o]
  k = 0
  for stage, wins in specmod.stage_windows(spec).items():
    for parent, (lo, hi) in wins.items():
      if k >= capi.MAX_WINDOWS:
        raise ValueError('too many stage windows')
      w = desc.window[k]
      w.stage, w.parent = index[stage], index[parent]
      for d in range(spec['dim']):
        w.lo[d], w.hi[d] = lo[d], hi[d]
      k += 1
  desc.n_windows = k
  return desc


def kernel_descs(table):
  arr = (capi.KernelDesc * len(table))()
  for d, k in zip(arr, table):
    d.name = k['name'].encode()
    d.kind = capi.KERNEL_FUSED if k['kind'] == 'fused' else capi.KERNEL_STAGE
    d.depth, d.stage = k['depth'], k['stage']
    for i in range(3):
      d.block[i] = k['block'][i]
    for i in range(4):
      d.tile[i] = k['tile'][i]
    d.fill_rows = k.get('fill_rows', 0)
    d.origin_align = k.get('origin_align', 0)
    for i, v in enumerate(k.get('min_extent', [0, 0])):
      d.min_extent[i] = v
    d.step_valu = int(k.get('step_valu', 0))
    d.step_bytes = int(k.get('step_bytes', 0))
    d.step_ns_full = int(k.get('step_ns_full', 0))
    d.step_ns_one = int(k.get('step_ns_one', 0))
    d.stream_gbps = int(k.get('stream_gbps', 0))
    d.xcd_tiles = int(k.get('xcd_tiles', 0))
    d.stream_wgs_per_cu = int(k.get('stream_wgs_per_cu', 0))
    d.fade_lo_mib = int(k.get('fade_lo_mib', 0))
    d.fade_hi_mib = int(k.get('fade_hi_mib', 0))
    d.stream_chunk = int(k.get('stream_chunk', 0))
    d.edge_slack = int(k.get('edge_slack', 0))
  return arr


class Program:
  """A stencil program bound to a blob: the run-time object behind the
  generated `<app>` / `<app>_test` functions."""

  def __init__(self, blob, spec=None):
    self.blob = blob
    meta = blob.meta
    if spec is not None:
      from ..codegen.kernel_common import program_hash
      if program_hash(spec) != meta['program_hash']:
        raise capi.SodaHipError(
            -103, 'blob_program_mismatch',
            'blob was generated for program %s (%s), host expects %s (%s)' % (
                meta['app_name'], meta['program_hash'][:12], spec['app_name'],
                program_hash(spec)[:12]))
    self.spec = meta['spec'] if spec is None else spec
    self.kernels = meta['kernels']
    # the plan runs the LOWERED program the kernels were generated from
    # (pointwise-only locals folded away); inputs and outputs are the source's
    self.lowered = meta['spec']
    self._desc = program_desc(self.lowered)
    self._kdesc = kernel_descs(self.kernels)
    h = ctypes.c_void_p()
    capi.check(capi.lib().soda_hip_plan_create(
        blob.handle, ctypes.byref(self._desc), self._kdesc, len(self.kernels),
        ctypes.byref(h)))
    self.handle = h
    types = specmod.tensor_c_types(self.spec)
    self.in_dtypes = [np.dtype(specmod.NUMPY_NAME[t['c_type']])
                      for t in self.spec['inputs']]
    self.out_dtypes = [np.dtype(specmod.NUMPY_NAME[types[o]])
                       for o in self.spec['outputs']]

  def close(self):
    if self.handle:
      capi.check(capi.lib().soda_hip_plan_destroy(self.handle))
      self.handle = None

  def set_max_depth(self, depth):
    """0 = deepest fused kernels available, -1 = per-stage kernels only."""
    capi.check(capi.lib().soda_hip_plan_set_max_depth(self.handle, depth))

  def set_out_final_only(self, on):
    """Sweeps write their output arrays with the last launch only (callers that
    sweep one array piece by piece: soda_hip_plan_set_out_final_only)."""
    capi.check(capi.lib().soda_hip_plan_set_out_final_only(self.handle, int(bool(on))))

  def margins(self, iterations):
    lo = (ctypes.c_int32 * 4)()
    hi = (ctypes.c_int32 * 4)()
    capi.check(capi.lib().soda_hip_plan_margins(self.handle, iterations, lo, hi))
    d = self.spec['dim']
    return tuple(lo[:d]), tuple(hi[:d])

  @staticmethod
  def _ptr_array(ptrs):
    return (ctypes.c_void_p * len(ptrs))(*ptrs)

  def _dims(self, dims):
    d = list(dims)[:self.spec['dim']]
    return (ctypes.c_int64 * 4)(*(d + [1] * (4 - len(d))))

  def sweep(self, in_ptrs, out_ptrs, dims, iterate, valid_lo=None, valid_hi=None,
            stream=None):
    """Asynchronous device sweep on raw device pointers."""
    vlo = vhi = None
    if valid_lo is not None:
      vlo = (ctypes.c_int32 * 4)(*(list(valid_lo) + [0] * (4 - len(valid_lo))))
    if valid_hi is not None:
      vhi = (ctypes.c_int32 * 4)(*(list(valid_hi) + [0] * (4 - len(valid_hi))))
    capi.check(capi.lib().soda_hip_sweep(
        self.handle, self._ptr_array(in_ptrs), self._ptr_array(out_ptrs),
        self._dims(dims), iterate, vlo, vhi, stream))

  def tune(self, in_ptrs, out_ptrs, dims, iterate, valid_lo=None, valid_hi=None,
           stream=None):
    """Runs the candidate splits of `iterate` as whole sweeps and keeps the fastest
    for later sweeps of the same extents (soda_hip_plan_tune)."""
    vlo = vhi = None
    if valid_lo is not None:
      vlo = (ctypes.c_int32 * 4)(*(list(valid_lo) + [0] * (4 - len(valid_lo))))
    if valid_hi is not None:
      vhi = (ctypes.c_int32 * 4)(*(list(valid_hi) + [0] * (4 - len(valid_hi))))
    capi.check(capi.lib().soda_hip_plan_tune(
        self.handle, self._ptr_array(in_ptrs), self._ptr_array(out_ptrs),
        self._dims(dims), iterate, vlo, vhi, stream))

  def tuned_streams(self):
    """How many (kernel, box) pairs run a MEASURED (chunk length, workgroups per CU)
    instead of the kernel's calibrated pair (soda_hip_plan_tuned_streams)."""
    n = ctypes.c_int()
    capi.check(capi.lib().soda_hip_plan_tuned_streams(self.handle, ctypes.byref(n)))
    return n.value

  def set_split(self, dims, iterate, depths):
    """Fixes the split of `iterate` into fused depths for these extents, in launch
    order; an empty list gives the choice back to the scheduler
    (soda_hip_plan_set_split)."""
    depths = list(depths or [])
    arr = (ctypes.c_int32 * max(1, len(depths)))(*depths)
    capi.check(capi.lib().soda_hip_plan_set_split(
        self.handle, self._dims(dims), iterate, arr, len(depths)))

  def schedule(self, dims, iterate, valid_lo=None, valid_hi=None):
    """The launches `sweep` would issue: [(kernel table entry, modelled us)]."""
    vlo = vhi = None
    if valid_lo is not None:
      vlo = (ctypes.c_int32 * 4)(*(list(valid_lo) + [0] * (4 - len(valid_lo))))
    if valid_hi is not None:
      vhi = (ctypes.c_int32 * 4)(*(list(valid_hi) + [0] * (4 - len(valid_hi))))
    n = ctypes.c_int()
    capi.check(capi.lib().soda_hip_plan_schedule(
        self.handle, self._dims(dims), iterate, vlo, vhi, None, None, 0,
        ctypes.byref(n)))
    idx = (ctypes.c_int32 * max(1, n.value))()
    est = (ctypes.c_double * max(1, n.value))()
    capi.check(capi.lib().soda_hip_plan_schedule(
        self.handle, self._dims(dims), iterate, vlo, vhi, idx, est, n.value,
        ctypes.byref(n)))
    return [(self.kernels[idx[i]], est[i]) for i in range(n.value)]

  def sweep_timed(self, in_ptrs, out_ptrs, dims, iterate, warmup=1, repeats=1,
                  stream=None):
    t = capi.Timing()
    capi.check(capi.lib().soda_hip_sweep_timed(
        self.handle, self._ptr_array(in_ptrs), self._ptr_array(out_ptrs),
        self._dims(dims), iterate, warmup, repeats, stream, ctypes.byref(t)))
    return dict(kernel_us=t.kernel_us, launches=t.launches, max_depth=t.max_depth,
                dominant_us=t.dominant_us, dominant_launches=t.dominant_launches,
                dominant_name=t.dominant_name.decode(), fastest_us=t.fastest_us)

  def shader_clock_during(self, work, seconds):
    """GHz the shader clock holds while `work()` (a callable that enqueues about
    `seconds` of sweeps and returns without waiting) runs: one probe wavefront sleeps
    beside it for ~60 % of that time (soda_hip_clock_probe_start / _finish).  None for
    blobs without the probe kernel."""
    spins = max(4, int(seconds * 0.6 / 3.9e-6))
    try:
      capi.check(capi.lib().soda_hip_clock_probe_start(self.handle, spins))
    except capi.SodaHipError as e:
      if e.code == capi.ERR_NO_KERNEL:     # a blob from before the probe kernel
        return None
      raise                                # an allocation or launch failure is an error
    ghz, took = ctypes.c_double(), ctypes.c_double()
    failed = True
    try:
      work()
      failed = False
    finally:
      # always collected: a probe left running would refuse every later one
      rc = capi.lib().soda_hip_clock_probe_finish(self.handle, ctypes.byref(ghz),
                                                  ctypes.byref(took))
      if not failed:
        capi.check(rc)
    return dict(ghz=ghz.value, seconds=took.value)

  # -- numpy conveniences (tests, <app>_test) --------------------------------
  def run_numpy(self, inputs, iterate=None, timed=False):
    """inputs: C-ordered arrays of shape reversed(dims).  Returns the output
    arrays (cells outside the valid box are zero) [and the timing dict]."""
    spec = self.spec
    iterate = spec['iterate'] if iterate is None else iterate
    shape = inputs[0].shape
    dims = tuple(reversed(shape))
    assert len(inputs) == len(spec['inputs'])
    for a, dt in zip(inputs, self.in_dtypes):
      assert a.dtype == dt and a.shape == shape, (a.dtype, dt, a.shape, shape)
    cells = int(np.prod(shape))
    din = [DeviceArray(cells * dt.itemsize) for dt in self.in_dtypes]
    dout = [DeviceArray(cells * dt.itemsize) for dt in self.out_dtypes]
    try:
      for d, a in zip(din, inputs):
        d.upload(a)
      for d in dout:
        d.zero()
      timing = None
      if timed:
        timing = self.sweep_timed([d.ptr for d in din], [d.ptr for d in dout],
                                  dims, iterate)
      else:
        self.sweep([d.ptr for d in din], [d.ptr for d in dout], dims, iterate)
        capi.check(capi.lib().soda_hip_stream_synchronize(None))
      raw = [d.download(shape, dt) for d, dt in zip(dout, self.out_dtypes)]
    finally:
      for d in din + dout:
        d.free()
    # every output is defined on ITS OWN box (reference host.py:1082-1091)
    boxes = specmod.iteration_boxes(spec, iterate)[-1]
    outs = []
    for name, a in zip(spec['outputs'], raw):
      lo, hi = boxes[name]
      # (an empty box stays empty: a negative stop would wrap around in numpy)
      sl = tuple(slice(-lo[d], max(-lo[d], dims[d] - hi[d]))
                 for d in reversed(range(spec['dim'])))
      clean = np.zeros_like(a)
      clean[sl] = a[sl]
      outs.append(clean)
    return (outs, timing) if timed else outs

  def run_buffers(self, inputs, outputs, iterate=None):
    """The generated `<app>(buffer_t*...)` (reference host.py:931-945): numpy
    arrays stand in for buffer_t; outputs are written in place, valid interior
    only; prints the reference's two timing lines."""
    spec = self.spec
    iterate = spec['iterate'] if iterate is None else iterate

    def as_buffer(a):
      b = capi.BufferT()
      b.host = a.ctypes.data
      stride = 1
      for d, n in enumerate(reversed(a.shape)):
        b.extent[d] = n
        b.stride[d] = stride
        stride *= n
      b.elem_size = a.dtype.itemsize
      return b
    ins = [as_buffer(a) for a in inputs]
    outs = [as_buffer(a) for a in outputs]
    pin = (ctypes.POINTER(capi.BufferT) * len(ins))(*[ctypes.pointer(b) for b in ins])
    pout = (ctypes.POINTER(capi.BufferT) * len(outs))(
        *[ctypes.pointer(b) for b in outs])
    t = capi.Timing()
    sys.stdout.flush()
    capi.check(capi.lib().soda_hip_run_buffers(self.handle, pin, pout, iterate,
                                               ctypes.byref(t)))
    return dict(kernel_us=t.kernel_us, launches=t.launches, max_depth=t.max_depth)


def open_program(blob=None, source=None, spec=None):
  """`blob`: path to a code object or to kernel text; `source`: kernel text."""
  if blob is not None:
    b = Blob.from_file(blob)
  elif source is not None:
    b = Blob.from_source(source)
  else:
    raise ValueError('need a blob path or kernel source')
  return Program(b, spec)


def reference_init(spec, dims):
  """Input pattern of the reference's `<app>_test` (host.py:1033-1051)."""
  shape = tuple(reversed(dims[:spec['dim']]))
  grids = np.meshgrid(*[np.arange(n, dtype=np.int64) for n in shape], indexing='ij')
  s = sum(grids)
  floaty = specmod.is_float_type(spec['inputs'][0]['haoda_type'])
  out = []
  for t in spec['inputs']:
    dt = np.dtype(specmod.NUMPY_NAME[t['c_type']])
    if floaty:
      out.append(np.ascontiguousarray(
          (s.astype(dt) / dt.type(sum(dims[:spec['dim']]))).astype(dt)))
    else:
      out.append(np.ascontiguousarray(s.astype(dt)))
  return out


def app_test(spec, blob, dims, source=None, iterate=None):
  """The generated `int <app>_test(const char* xclbin, const int dims[4])`
  (reference host.py:984-1167): zero-filled arrays, the reference's input
  pattern, one call of `<app>`, CPU re-computation, comparison on the valid
  region (exact for integers, relative 1e-5 -- env THRESHOLD overrides -- for
  floats), `INFO: PASS!` / `INFO: FAIL!` on stderr, returns the mismatch count."""
  from . import selfcheck
  iterate = spec['iterate'] if iterate is None else iterate
  dims = [int(v) for v in list(dims)[:spec['dim']]]
  prog = open_program(blob=blob, source=source, spec=spec)
  try:
    inputs = reference_init(spec, dims)
    shape = inputs[0].shape
    outputs = [np.zeros(shape, dtype=dt) for dt in prog.out_dtypes]
    prog.run_buffers(inputs, outputs, iterate)
    threshold = float(os.environ.get('THRESHOLD', '0.00001'))
    errors = int(selfcheck.count_mismatches(spec, inputs, outputs, iterate,
                                            threshold))
  finally:
    prog.close()
    prog.blob.unload()
  sys.stderr.write('INFO: PASS!\n' if errors == 0 else 'INFO: FAIL!\n')
  sys.stderr.flush()
  return errors
