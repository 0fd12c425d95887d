"""The neutral, JSON-able description of a stencil program that the HIP back end
works from ("program spec").

Everything downstream of the analysis -- the kernel printer, the host-shim
printer, the run-time library binding, the CPU oracle used by the tests -- sees a
program only through this dictionary, so the back end can be driven by this
project's own front end (`spec_from_stencil`) or plugged into the reference's
driver and fed its `soda.core.Stencil` (`spec_from_reference_stencil`).

Layout:

  app_name, dim, iterate, burst_width, unroll_factor, tile_size
  inputs   [{name, haoda_type, c_type}]                 program inputs, in order
  outputs  [name]                                       output j feeds input j
  stages   [{name, kind, haoda_type, c_type,            one iteration's stages in
             lets: [{name, c_type, expr}], expr,        execution order
             loads: [[tensor, [d0, d1, ..]], ..]}]
  radius   {lo: [..], hi: [..]}                         growth per iteration

`expr` is the C++ text of the reference's `c_expr` (reference
src/haoda/ir/__init__.py:213-222, :320-346) with each tensor element spelled
`{tensor:d0,d1,..}`, the offsets being RELATIVE to the element being produced
(`ld.idx - st.idx`, reference src/soda/codegen/xilinx/host.py:1093-1102).
"""
import json
import re

LOAD_RE = re.compile(r'\{(\w+):(-?\d+(?:,-?\d+)*)\}')

# HIP/C types of the DSL types the back end accepts
_NATIVE = {'uint8': 'uint8_t', 'uint16': 'uint16_t', 'uint32': 'uint32_t',
           'uint64': 'uint64_t', 'int8': 'int8_t', 'int16': 'int16_t',
           'int32': 'int32_t', 'int64': 'int64_t', 'float': 'float',
           'float32': 'float', 'double': 'double', 'float64': 'double',
           'half': '_Float16', 'float16': '_Float16'}
ELEM_SIZE = {'uint8_t': 1, 'int8_t': 1, 'uint16_t': 2, 'int16_t': 2,
             '_Float16': 2, 'uint32_t': 4, 'int32_t': 4, 'float': 4,
             'uint64_t': 8, 'int64_t': 8, 'double': 8}
NUMPY_NAME = {'uint8_t': 'uint8', 'int8_t': 'int8', 'uint16_t': 'uint16',
              'int16_t': 'int16', '_Float16': 'float16', 'uint32_t': 'uint32',
              'int32_t': 'int32', 'float': 'float32', 'uint64_t': 'uint64',
              'int64_t': 'int64', 'double': 'float64'}


class UnsupportedProgram(Exception):
  """The program is valid SODA but outside what the HIP back end handles."""


def native_type(haoda_type):
  try:
    return _NATIVE[haoda_type]
  except KeyError:
    raise UnsupportedProgram(
        'type %s has no native GPU representation (the HIP back end supports '
        '[u]int8/16/32/64, half, float, double)' % haoda_type)


def is_float_type(haoda_type):
  return haoda_type in ('half', 'double') or haoda_type.startswith('float')


def load_placeholder(name, rel):
  return '{%s:%s}' % (name, ','.join(str(int(v)) for v in rel))


# ---------------------------------------------------------------------------
# builders
# ---------------------------------------------------------------------------
def spec_from_stencil(st):
  """From this project's `soda_hip.frontend.Stencil`."""
  from ..frontend import expr as ex

  def expr_text(node, stage):
    def load_text(ld):
      return load_placeholder(
          ld.name, [a - b for a, b in zip(ld.idx, stage.st_idx)])
    return ex.c_text(node, load_text, native_type)

  stages = []
  for name in st.order:
    stage = st.stages[name]
    stages.append(dict(
        name=name, kind=stage.kind, haoda_type=stage.haoda_type,
        c_type=native_type(stage.haoda_type),
        lets=[dict(name=n, c_type=native_type(t), expr=expr_text(e, stage))
              for n, t, e in stage.lets],
        expr=expr_text(stage.expr, stage),
        loads=[[n, list(rel)] for n, rel in stage.rel_loads()]))
  lo, hi = st.radius()
  return _finish(dict(
      app_name=st.app_name, dim=st.dim, iterate=st.iterate,
      burst_width=st.burst_width, unroll_factor=st.unroll_factor,
      tile_size=list(st.tile_size),
      inputs=[dict(name=n, haoda_type=t, c_type=native_type(t))
              for n, t in zip(st.input_names, st.input_types)],
      outputs=list(st.output_names), stages=stages,
      radius=dict(lo=list(lo), hi=list(hi))))


def spec_from_reference_stencil(st):
  """From the REFERENCE's `soda.core.Stencil` (duck-typed; only works inside
  the reference's own process, where `haoda.ir` is importable).  This is what
  lets `soda_hip.codegen.backend.print_code(stencil, args)` be called from the
  reference's `sodac` with two added lines."""
  from haoda import ir  # the reference's module, present in its process only

  first_iter = []
  for tensor in st.chronological_tensors:
    if tensor.is_input():
      continue
    first_iter.append(tensor)
    if len(first_iter) == len(st.local_stmts) + len(st.output_stmts):
      break
  # names of iteration 0: outputs are renamed `<input>_iter1` when iterate > 1
  def base_name(name):
    if name.endswith('_iter1'):
      return st.output_names[st.input_names.index(name[:-6])]
    return name

  stages = []
  for tensor in first_iter:
    def mutate(obj, args, tensor=tensor):
      if isinstance(obj, ir.Ref):
        return ir.make_var(load_placeholder(
            obj.name, [a - b for a, b in zip(obj.idx, tensor.st_ref.idx)]))
      return obj
    loads, seen = [], set()
    def collect(obj, args, tensor=tensor):
      if isinstance(obj, ir.Ref):
        rel = [a - b for a, b in zip(obj.idx, tensor.st_ref.idx)]
        if (obj.name, tuple(rel)) not in seen:
          seen.add((obj.name, tuple(rel)))
          loads.append([obj.name, rel])
      return obj
    tensor.visit_loads(collect)
    name = base_name(tensor.name)
    stages.append(dict(
        name=name, kind='output' if name in st.output_names else 'local',
        haoda_type=tensor.haoda_type, c_type=native_type(tensor.haoda_type),
        lets=[dict(name=l.name, c_type=native_type(l.haoda_type),
                   expr=_retype(l.expr.visit(mutate).c_expr)) for l in tensor.lets],
        expr=_retype(tensor.expr.visit(mutate).c_expr), loads=loads))
  spec = dict(
      app_name=st.app_name, dim=st.dim, iterate=st.iterate,
      burst_width=st.burst_width, unroll_factor=st.unroll_factor,
      tile_size=list(st.tile_size),
      inputs=[dict(name=n, haoda_type=t, c_type=native_type(t))
              for n, t in zip(st.input_names, st.input_types)],
      outputs=list(st.output_names), stages=stages)
  lo, hi = iteration_margins(spec, 1)[0]
  spec['radius'] = dict(lo=list(lo), hi=list(hi))
  return _finish(spec)


def _retype(text):
  """`static_cast<ap_uint<5> >` style casts cannot reach here (native_type
  raised already); native casts print identically in both worlds."""
  return text


def _finish(spec):
  names = [t['name'] for t in spec['inputs']] + [s['name'] for s in spec['stages']]
  if len(set(names)) != len(names):
    raise UnsupportedProgram('duplicate tensor names: %s' % names)
  if spec['dim'] not in (1, 2, 3, 4):
    raise UnsupportedProgram('%d-D programs are not supported' % spec['dim'])
  return spec


# ---------------------------------------------------------------------------
# queries on a spec
# ---------------------------------------------------------------------------
def dumps(spec):
  return json.dumps(spec, sort_keys=True)


def tensor_c_types(spec):
  types = {t['name']: t['c_type'] for t in spec['inputs']}
  types.update({s['name']: s['c_type'] for s in spec['stages']})
  return types


def substitute_loads(text, fn):
  """Replaces every `{tensor:d0,d1}` in `text` by `fn(tensor, (d0, d1))`."""
  return LOAD_RE.sub(
      lambda m: fn(m.group(1), tuple(int(v) for v in m.group(2).split(','))),
      text)


def stage_windows(spec):
  """{stage: {parent: (lo, hi)}}: hull of each stage's relative load offsets."""
  out = {}
  for stage in spec['stages']:
    win = {}
    for name, rel in stage['loads']:
      if name in win:
        lo, hi = win[name]
        win[name] = ([min(a, b) for a, b in zip(lo, rel)],
                     [max(a, b) for a, b in zip(hi, rel)])
      else:
        win[name] = (list(rel), list(rel))
    out[stage['name']] = win
  return out


def iteration_boxes(spec, iterations):
  """Per iteration, {stage: (lo, hi)} = hull of the composed read offsets back
  to the ORIGINAL program inputs (lo <= 0 <= hi per dimension).  Stage values
  are defined on `[-lo_d, N_d - hi_d)` (reference core.py:794-835 composed
  window, host.py:1082-1091 loop bounds)."""
  dim = spec['dim']
  windows = stage_windows(spec)
  zero = ([0] * dim, [0] * dim)
  feed = {t['name']: zero for t in spec['inputs']}
  in_names = [t['name'] for t in spec['inputs']]
  result = []
  for _ in range(iterations):
    boxes = dict(feed)
    for stage in spec['stages']:
      lo, hi = None, None
      for parent, (wlo, whi) in windows[stage['name']].items():
        plo, phi = boxes[parent]
        clo = [a + b for a, b in zip(plo, wlo)]
        chi = [a + b for a, b in zip(phi, whi)]
        lo = clo if lo is None else [min(a, b) for a, b in zip(lo, clo)]
        hi = chi if hi is None else [max(a, b) for a, b in zip(hi, chi)]
      # the cell itself must be inside the array: boxes contain the origin
      boxes[stage['name']] = ([min(0, v) for v in lo], [max(0, v) for v in hi])
    result.append({s['name']: boxes[s['name']] for s in spec['stages']})
    if len(in_names) == len(spec['outputs']):
      feed = {i: boxes[o] for i, o in zip(in_names, spec['outputs'])}
  return result


def inline_pointwise(spec):
  """Lowers a spec by folding away every local stage that all its readers read
  ONLY at offset 0 (diff_u, r0, r1 ... of the denoise programs): each
  `{L:0,0}` in a reader becomes `static_cast<T_L >(<L's expression>)`, L
  disappears from the stage list.  The arithmetic is unchanged -- the same
  operations in the same order, the cast reproducing the rounding of L's store
  -- but L no longer costs an array pass (per-stage kernels) or a register
  window (fused kernels).  Stages with `let`s are left alone.  Returns a new
  spec; inputs, outputs and composed windows are the same as the original's."""
  import copy
  spec = copy.deepcopy(spec)
  dim = spec['dim']
  zero = [0] * dim
  outs = set(spec['outputs'])
  while True:
    victim = None
    for stage in spec['stages']:
      name = stage['name']
      if name in outs or stage['lets']:
        continue
      readers = [s for s in spec['stages']
                 if any(t == name for t, _ in s['loads'])]
      if not readers:
        continue
      if all(rel == zero for s in readers for t, rel in s['loads'] if t == name):
        victim = stage
        break
    if victim is None:
      return spec
    name = victim['name']
    inlined = 'static_cast<%s >(%s)' % (victim['c_type'], victim['expr'])
    hole = load_placeholder(name, zero)
    for s in spec['stages']:
      if not any(t == name for t, _ in s['loads']):
        continue
      s['expr'] = s['expr'].replace(hole, inlined)
      for let in s['lets']:
        let['expr'] = let['expr'].replace(hole, inlined)
      merged = []
      for t, rel in s['loads']:
        for item in (victim['loads'] if t == name else [[t, rel]]):
          if item not in merged:
            merged.append([item[0], list(item[1])])
      s['loads'] = merged
    spec['stages'] = [s for s in spec['stages'] if s['name'] != name]


def iteration_margins(spec, iterations):
  """[(lo, hi)] per iteration for the program outputs (hull over outputs), as
  non-negative margins: after k+1 iterations outputs live on [lo_d, N_d-hi_d)."""
  out = []
  for boxes in iteration_boxes(spec, iterations):
    lo = hi = None
    for name in spec['outputs']:
      blo, bhi = boxes[name]
      lo = blo if lo is None else [min(a, b) for a, b in zip(lo, blo)]
      hi = bhi if hi is None else [max(a, b) for a, b in zip(hi, bhi)]
    out.append((tuple(-v for v in lo), tuple(hi)))
  return out


def valid_cells(spec, dims, iterations):
  """Number of defined output cells summed over iterations 1..N (the "valid
  cell-updates" of SURVEY.md section 8d)."""
  total = 0
  for lo, hi in iteration_margins(spec, iterations):
    cells = 1
    for d in range(spec['dim']):
      cells *= max(0, dims[d] - lo[d] - hi[d])
    total += cells
  return total


def algorithmic_bytes_per_update(spec):
  """sizeof(T) x (#program inputs + #program outputs); intermediates never
  need to touch HBM."""
  types = tensor_c_types(spec)
  return (sum(ELEM_SIZE[t['c_type']] for t in spec['inputs']) +
          sum(ELEM_SIZE[types[n]] for n in spec['outputs']))
