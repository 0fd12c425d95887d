"""Per-stage kernels: one launch computes ONE stage of ONE iteration over its
box, operands straight from global memory, intermediates through HBM.

This is the general-purpose form (any dimension 1..4, any number of inputs and
outputs, any window): the direct GPU counterpart of one loop nest of the
reference's CPU golden model (reference host.py:1076-1117).  It is used (a) for
programs the fused generators do not cover and (b) as the in-GPU cross-check of
the fused kernels in the tests.

Each work-item produces `V` consecutive cells along dimension 0 (V elements =
16 bytes where the types allow).  For every distinct (tensor, offset in the
outer dimensions) row the stage reads, the work-item issues ONE V-wide vector
load for the cells straight above/below/beside its own and single-element loads
only for the few cells that stick out left and right; the per-cell expression
then picks its operands out of those registers.  A 7-point 3-D stencil costs
5 vector + 2 scalar loads per 4 cells instead of 28 scalar loads.  Neighbouring
work-items' overlapping reads are served by L1/L2; HBM sees each tensor about
once, so the kernel is bound at
  (#distinct tensors read + 1) x sizeof(T) bytes per cell per stage.
"""
from . import spec as specmod
from .kernel_common import builtin_type, device_expr, tensor_index

BLOCK = 256


def kernel_name(spec, stage):
  return '%s_stage_%s' % (spec['app_name'], stage['name'])


def vector_width(spec, stage):
  types = specmod.tensor_c_types(spec)
  sizes = [specmod.ELEM_SIZE[types[t]] for t, _ in stage['loads']]
  sizes.append(specmod.ELEM_SIZE[stage['c_type']])
  return max(1, min(8, 16 // max(sizes)))


def emit(spec):
  """Returns (text, kernel table entries)."""
  dim = spec['dim']
  types = specmod.tensor_c_types(spec)
  index = tensor_index(spec)
  out, table = [], []
  for stage in spec['stages']:
    name = kernel_name(spec, stage)
    V = vector_width(spec, stage)
    ctype = builtin_type(stage['c_type'])
    # rows read: (tensor, outer offsets) -> (min dx, max dx)
    rows = {}
    for tensor, rel in stage['loads']:
      key = (tensor, tuple(rel[1:]))
      # the row always covers dx = 0: the V-wide centre load lands in
      # r[-lo .. -lo+V), so a one-sided window (all dx > 0 or all < 0) must not
      # shrink the register row below it
      lo, hi = rows.get(key, (min(rel[0], 0), max(rel[0], 0)))
      rows[key] = (min(lo, rel[0]), max(hi, rel[0]))
    lines = ['// stage `%s`: %d cells per work-item along dimension 0'
             % (stage['name'], V),
             'GLOBAL WG_SIZE(%d) void %s(soda_hip_args a) {' % (BLOCK, name),
             '  const i64 x = a.box_lo[0] + ((i64)__builtin_amdgcn_workgroup_id_x()'
             ' * %d + __builtin_amdgcn_workitem_id_x()) * %d;' % (BLOCK, V)]
    lines.append('  if (x >= a.box_hi[0]) return;')
    # rows / planes: one per workgroup.  Boxes with more than 65535 rows or planes
    # (the limit of grid.y / grid.z) are launched with the rows and planes folded
    # into one index n = wgid_y + wgid_z * grid.y (param[0] = 1, soda_hip.cpp:
    # make_launch); everything else keeps the direct mapping.
    if dim == 2:
      lines.append('  const i64 y = a.box_lo[1] + __builtin_amdgcn_workgroup_id_y() + '
                   '(i64)__builtin_amdgcn_workgroup_id_z() * __builtin_amdgcn_grid_size_y();')
      lines.append('  if (y >= a.box_hi[1]) return;')
    if dim == 3:
      lines.append('  i64 y = a.box_lo[1] + __builtin_amdgcn_workgroup_id_y();')
      lines.append('  i64 z = a.box_lo[2] + __builtin_amdgcn_workgroup_id_z();')
      lines.append('  if (a.param[0] == 1) {')
      lines.append('    const i64 ey = a.box_hi[1] - a.box_lo[1];')
      lines.append('    const i64 n = __builtin_amdgcn_workgroup_id_y() + '
                   '(i64)__builtin_amdgcn_workgroup_id_z() * __builtin_amdgcn_grid_size_y();')
      lines.append('    y = a.box_lo[1] + n % ey;')
      lines.append('    z = a.box_lo[2] + n / ey;')
      lines.append('    if (z >= a.box_hi[2]) return;')
      lines.append('  }')
    if dim == 4:
      # four dimensions: the launcher always folds dimensions 1..3 into grid.y x grid.z
      lines.append('  const i64 ey = a.box_hi[1] - a.box_lo[1];')
      lines.append('  const i64 ez = a.box_hi[2] - a.box_lo[2];')
      lines.append('  const i64 n = __builtin_amdgcn_workgroup_id_y() + '
                   '(i64)__builtin_amdgcn_workgroup_id_z() * __builtin_amdgcn_grid_size_y();')
      lines.append('  const i64 y = a.box_lo[1] + n % ey;')
      lines.append('  const i64 z = a.box_lo[2] + n / ey % ez;')
      lines.append('  const i64 w = a.box_lo[3] + n / (ey * ez);')
      lines.append('  if (w >= a.box_hi[3]) return;')
    if dim >= 2:
      lines.append('  const i64 s1 = a.dims[0];')
    if dim >= 3:
      lines.append('  const i64 s2 = a.dims[0] * a.dims[1];')
    if dim >= 4:
      lines.append('  const i64 s3 = a.dims[0] * a.dims[1] * a.dims[2];')
    cell = ('x' + (' + y * s1' if dim >= 2 else '') + (' + z * s2' if dim >= 3 else '')
            + (' + w * s3' if dim >= 4 else ''))
    lines.append('  const i64 c = %s;' % cell)
    lines.append('  const bool whole = x + %d <= a.box_hi[0];' % V)
    seen = []
    for tensor, _ in stage['loads']:
      if tensor not in seen:
        seen.append(tensor)
        lines.append('  const %s* __restrict__ t_%s = (const %s*)a.tensor[%d];' % (
            builtin_type(types[tensor]), tensor, builtin_type(types[tensor]),
            index[tensor]))
    lines.append('  %s* __restrict__ t_out = (%s*)a.tensor[%d];' % (
        ctype, ctype, index[stage['name']]))

    # load every row once: r_<id>[j] holds element x + lo + j of that row.
    # Reads stay inside the array: the run time guarantees box +- window does,
    # and elements past the box in a partial vector are clamped to the last cell.
    row_ids = {}
    for n, ((tensor, outer), (lo, hi)) in enumerate(rows.items()):
      rid = 'r%d' % n
      row_ids[(tensor, outer)] = (rid, lo)
      tt = builtin_type(types[tensor])
      base = ['c']
      if dim >= 2 and outer[0]:
        base.append('(%d) * s1' % outer[0])
      if dim >= 3 and outer[1]:
        base.append('(%d) * s2' % outer[1])
      if dim >= 4 and outer[2]:
        base.append('(%d) * s3' % outer[2])
      width = hi - lo + V
      lines.append('  %s %s[%d];  // %s, outer offset %s, dx %d..%d'
                   % (tt, rid, width, tensor, list(outer), lo, hi))
      lines.append('  {')
      lines.append('    const %s* p = t_%s + %s;' % (tt, tensor, ' + '.join(base)))
      if V > 1:
        vec = '%s __attribute__((ext_vector_type(%d), aligned(%d)))' % (
            tt, V, specmod.ELEM_SIZE[types[tensor]])
        lines.append('    if (whole) {')
        lines.append('      typedef %s vec_t;' % vec)
        lines.append('      const vec_t v = *(const vec_t*)p;')
        for j in range(V):
          lines.append('      %s[%d] = v[%d];' % (rid, j - lo, j))
        lines.append('    } else {')
        for j in range(V):
          # readable while inside box + this row's reach to the right
          lines.append('      %s[%d] = p[x + %d < a.box_hi[0] + %d ? %d : 0];'
                       % (rid, j - lo, j, hi, j))
        lines.append('    }')
      else:
        lines.append('    %s[%d] = p[0];' % (rid, -lo))
      for j in range(lo, 0):
        lines.append('    %s[%d] = p[%d];' % (rid, j - lo, j))
      for j in range(1, hi + 1):
        lines.append('    %s[%d] = whole ? p[%d] : p[x + %d < a.box_hi[0] + %d ? '
                     '%d : 0];' % (rid, V - 1 + j - lo, V - 1 + j, V - 1 + j, hi,
                                   V - 1 + j))
      lines.append('  }')

    lines.append('  %s result[%d];' % (ctype, V))
    for j in range(V):
      def load(tensor, rel, j=j):
        rid, lo = row_ids[(tensor, tuple(rel[1:]))]
        return '%s[%d]' % (rid, j + rel[0] - lo)
      if stage['lets']:
        lines.append('  {')
        for let in stage['lets']:
          lines.append('    const %s %s = %s;' % (
              builtin_type(let['c_type']), let['name'],
              specmod.substitute_loads(device_expr(let['expr']), load)))
        lines.append('    result[%d] = %s;' % (
            j, specmod.substitute_loads(device_expr(stage['expr']), load)))
        lines.append('  }')
      else:
        lines.append('  result[%d] = %s;' % (
            j, specmod.substitute_loads(device_expr(stage['expr']), load)))
    if V > 1:
      lines.append('  if (whole) {')
      lines.append('    typedef %s __attribute__((ext_vector_type(%d), aligned(%d)))'
                   ' out_t;' % (ctype, V, specmod.ELEM_SIZE[stage['c_type']]))
      lines.append('    out_t v;')
      for j in range(V):
        lines.append('    v[%d] = result[%d];' % (j, j))
      lines.append('    *(out_t*)(t_out + c) = v;')
      lines.append('  } else {')
      for j in range(V):
        lines.append('    if (x + %d < a.box_hi[0]) t_out[c + %d] = result[%d];'
                     % (j, j, j))
      lines.append('  }')
    else:
      lines.append('  t_out[c] = result[0];')
    lines.append('}')
    out.append('\n'.join(lines))
    table.append(dict(name=name, kind='stage', depth=0, stage=index[stage['name']],
                      block=[BLOCK, 1, 1], tile=[BLOCK * V, 1, 1, 1], vector=V))
  return '\n\n'.join(out) + '\n', table
