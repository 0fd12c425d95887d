"""Per-stage kernels: one launch computes ONE stage of ONE iteration over its
box, one cell per work-item, operands straight from global memory.

This is the general-purpose form (any dimension 1..3, any number of inputs and
outputs, any window): the direct GPU counterpart of one loop nest of the
reference's CPU golden model (reference host.py:1076-1117).  Intermediates go
through HBM, so it is bandwidth-bound at
  (#loads' distinct tensors + 1) x sizeof(T) per cell per stage
and is used (a) for programs the fused generators do not cover and (b) as the
in-GPU cross-check of the fused kernels in the tests.
"""
from . import spec as specmod
from .kernel_common import builtin_type, device_expr, tensor_index

BLOCK = 256


def kernel_name(spec, stage):
  return '%s_stage_%s' % (spec['app_name'], stage['name'])


def emit(spec):
  """Returns (text, kernel table entries)."""
  dim = spec['dim']
  types = specmod.tensor_c_types(spec)
  index = tensor_index(spec)
  out, table = [], []
  for stage in spec['stages']:
    name = kernel_name(spec, stage)
    parents = []
    for tensor, _ in stage['loads']:
      if tensor not in parents:
        parents.append(tensor)
    lines = ['// stage `%s`: one cell per work-item' % stage['name'],
             'GLOBAL WG_SIZE(%d) void %s(soda_hip_args a) {' % (BLOCK, name),
             '  const i64 x = a.box_lo[0] + (i64)__builtin_amdgcn_workgroup_id_x()'
             ' * %d + __builtin_amdgcn_workitem_id_x();' % BLOCK]
    if dim >= 2:
      lines.append('  const i64 y = a.box_lo[1] + __builtin_amdgcn_workgroup_id_y();')
    if dim >= 3:
      lines.append('  const i64 z = a.box_lo[2] + __builtin_amdgcn_workgroup_id_z();')
    lines.append('  if (x >= a.box_hi[0]) return;')
    if dim >= 2:
      lines.append('  const i64 s1 = a.dims[0];')
    if dim >= 3:
      lines.append('  const i64 s2 = a.dims[0] * a.dims[1];')
    cell = 'x' + (' + y * s1' if dim >= 2 else '') + (' + z * s2' if dim >= 3 else '')
    lines.append('  const i64 c = %s;' % cell)
    for tensor in parents:
      lines.append('  const %s* __restrict__ t_%s = (const %s*)a.tensor[%d];' % (
          builtin_type(types[tensor]), tensor, builtin_type(types[tensor]),
          index[tensor]))

    def load(tensor, rel):
      off = ['c']
      if rel[0]:
        off.append('(%d)' % rel[0])
      if dim >= 2 and rel[1]:
        off.append('(%d) * s1' % rel[1])
      if dim >= 3 and rel[2]:
        off.append('(%d) * s2' % rel[2])
      return 't_%s[%s]' % (tensor, ' + '.join(off))

    for let in stage['lets']:
      lines.append('  const %s %s = %s;' % (
          builtin_type(let['c_type']), let['name'],
          specmod.substitute_loads(device_expr(let['expr']), load)))
    ctype = builtin_type(stage['c_type'])
    lines.append('  ((%s*)a.tensor[%d])[c] = %s;' % (
        ctype, index[stage['name']],
        specmod.substitute_loads(device_expr(stage['expr']), load)))
    lines.append('}')
    out.append('\n'.join(lines))
    table.append(dict(name=name, kind='stage', depth=0, stage=index[stage['name']],
                      block=[BLOCK, 1, 1], tile=[BLOCK, 1, 1, 1]))
  return '\n\n'.join(out) + '\n', table
