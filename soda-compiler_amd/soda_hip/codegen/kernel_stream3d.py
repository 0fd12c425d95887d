"""Fused 3-D kernels: `depth` iterations of a single-output 3-D program in one
launch, streaming PLANES along the outermost dimension ("2.5-D blocking") with
every intermediate in registers.

Same construction as kernel_stream2d (stage instances trailing a load head,
windows rotated by unrolling), one dimension up:

  * dimension 2 (z) is streamed: a window slot holds one plane tile;
  * dimension 0 (x): lane l holds C consecutive columns, x-neighbours across
    lanes by DPP wave shifts, as in 2-D;
  * dimension 1 (y): the lane also holds R consecutive ROWS of its columns, so
    a wavefront owns a (64*C) x R tile of every live plane and y-neighbours are
    plain registers of the same lane.  Tiles overlap by the composed x/y window
    of `depth` iterations (halo cells recomputed), nothing is shared between
    wavefronts: no LDS, no barrier.

For a 7-point stencil only the centre plane of a window is read off-axis; the
planes z-1 / z+1 contribute their centre cell, so a level costs three plane tiles
of registers (3*R*C per lane).  With R = 16, C = 2 and depth 2 that is ~200
VGPRs, two waves per SIMD, and HBM sees (4/0.73 + 4)/2 = 4.7 B per update
instead of 8.
"""
from . import kernel_common
from . import spec as specmod
from .kernel_common import builtin_type, cell_assignment, device_expr, tensor_index
from .kernel_stream2d import (LANES, WAVES_PER_BLOCK, NotFusable,
                              build_pipeline)


def kernel_name(spec, depth):
  return '%s_fused_k%d' % (spec['app_name'], depth)


def pipeline(spec, depth, prefetch):
  """build_pipeline() streams along dimension 1; re-key the loads so that it
  sees the plane offset there, then restore the full offsets."""
  flat = dict(spec, dim=2)
  flat['stages'] = [dict(s, loads=[[t, [rel[0], rel[2]]] for t, rel in s['loads']])
                    for s in spec['stages']]
  insts, final = build_pipeline(flat, depth, prefetch)
  # build_pipeline de-duplicated nothing (loads are unique in 3-D already, but
  # two 3-D loads can collapse onto one (x, z) pair): rebuild reads in 3-D
  by_ident = {i.ident: i for i in insts}
  ins = [t['name'] for t in spec['inputs']]
  current = {n: by_ident['in_%s' % n] for n in ins}
  for it in range(depth):
    for stage in spec['stages']:
      inst = by_ident['k%d_%s' % (it, stage['name'])]
      inst.reads = [(current[t], tuple(rel), t) for t, rel in stage['loads']]
      current[stage['name']] = inst
    if len(ins) == len(spec['outputs']):
      for i, o in zip(ins, spec['outputs']):
        current[i] = current[o]
  return insts, final


def emit(spec, depth, cols=2, rows=16, chunk_planes=64, prefetch=0, max_period=12,
         vgpr_budget=250, waves_per_eu=0, nt=0):
  """Returns (text, kernel table entry) for one fused depth.

  nt = 4: launches whose box (all inputs and outputs) is larger than the Infinity
  Cache store around the caches - a second instantiation of the interior path, chosen
  by the kernel's entry (kernel_common.NT_STREAMING_BYTES); 2 = always."""
  if spec['dim'] != 3:
    raise NotFusable('stream3d handles 3-D programs')
  if len(spec['outputs']) != 1:
    raise NotFusable('stream3d handles single-output programs')
  types = specmod.tensor_c_types(spec)
  index = tensor_index(spec)
  in_type = spec['inputs'][0]['c_type']
  out_name = spec['outputs'][0]
  elem = specmod.ELEM_SIZE[in_type]
  if any(specmod.ELEM_SIZE[t] != elem for t in types.values()):
    raise NotFusable('mixed element widths')
  if elem not in (4, 8):
    raise NotFusable('stream3d handles 4- and 8-byte elements')
  if depth > 1 and len(spec['inputs']) != 1:
    raise NotFusable('depth > 1 needs one input feeding one output')
  C, R = cols, rows
  insts, final = pipeline(spec, depth, prefetch)
  margins = specmod.iteration_margins(spec, depth)
  lo, hi = margins[-1] if len(spec['inputs']) == len(spec['outputs']) else margins[0]
  halo_lo = -(-lo[0] // C) * C
  halo_hi = -(-hi[0] // C) * C
  w_out = LANES * C - halo_lo - halo_hi
  y_lo, y_hi = lo[1], hi[1]
  r_out = R - y_lo - y_hi
  if w_out < C or r_out < 1:
    raise NotFusable('depth %d leaves no output cells in a %dx%d tile'
                     % (depth, LANES * C, R))
  for inst in insts:
    for src, rel, _ in inst.reads:
      if abs(rel[0]) > C:
        raise NotFusable('x offset %d exceeds the %d columns a lane holds'
                         % (rel[0], C))
  best = None
  for candidate in range(1, max_period + 1):
    if max(inst.keep for inst in insts) > candidate:
      continue
    divisors = [d for d in range(1, candidate + 1) if candidate % d == 0]
    padded = [min(d for d in divisors if d >= inst.keep) if inst.keep else 0
              for inst in insts]
    cost = (sum(padded), candidate)
    if best is None or cost < best[0]:
      best = (cost, candidate, padded)
  if best is None:
    raise NotFusable('window deeper than the rotation period limit')
  period = best[1]
  for inst, keep in zip(insts, best[2]):
    inst.keep = keep
  per_elem = max(1, elem // 4)
  est_vgprs = sum(inst.keep * R * C * per_elem for inst in insts) + \
      R * C * per_elem + 24
  if est_vgprs > vgpr_budget:
    raise NotFusable('depth %d would need about %d VGPRs (budget %d)'
                     % (depth, est_vgprs, vgpr_budget))
  stage_boxes = specmod.iteration_boxes(spec, depth)
  name = kernel_name(spec, depth)
  L = final.lag
  T_in = builtin_type(in_type)
  T_out = builtin_type(types[out_name])
  o = []
  line = o.append
  line('// fused depth-%d 3-D kernel: tile %d x %d per wavefront (%d x %d out),'
       % (depth, LANES * C, R, w_out, r_out))
  line('// rotation period %d, prefetch %d planes, ~%d VGPRs' % (period, prefetch,
                                                                  est_vgprs))
  for inst in insts:
    line('//   %-18s lag %2d keep %2d%s' % (inst.ident, inst.lag, inst.keep,
                                           '  -> HBM' if inst.final else ''))
  vec_in = 'vec_%s_in' % name
  vec_out = 'vec_%s_out' % name
  line('typedef %s %s __attribute__((ext_vector_type(%d), aligned(%d)));'
       % (T_in, vec_in, C, elem))
  line('typedef %s %s __attribute__((ext_vector_type(%d), aligned(%d)));'
       % (T_out, vec_out, C, elem))
  line('template <bool INTERIOR, bool NT = %s>' % ('true' if nt & 2 else 'false'))
  line('DEV void %s_tile(const soda_hip_args& a, const i64 xs, const i64 x, '
       'const i64 yb, const i64 z0, const i64 z1) {' % name)
  line('  const i64 W = a.dims[0], H = a.dims[1], D = a.dims[2];')
  line('  const i64 plane = W * H;')
  line('  const i64 st_lo = xs > a.box_lo[0] ? xs : a.box_lo[0];')
  line('  const i64 st_hi = xs + %d < a.box_hi[0] ? xs + %d : a.box_hi[0];'
       % (w_out, w_out))
  # rows of the tile, clamped into the array (clamped rows only feed halo
  # cells); wave-uniform, so they live in scalar registers
  line('  i64 row_y[%d];' % R)
  for r in range(R):
    line('  { i64 y = yb + %d; y = y < 0 ? 0 : (y > H - 1 ? H - 1 : y); '
         'row_y[%d] = y * W; }' % (r, r))
  for t in spec['inputs']:
    line('  const %s* __restrict__ g_%s = (const %s*)a.tensor[%d];' % (
        builtin_type(t['c_type']), t['name'], builtin_type(t['c_type']),
        index[t['name']]))
  line('  %s* __restrict__ g_out = (%s*)a.tensor[%d];' % (T_out, T_out,
                                                          index[out_name]))
  for inst in insts:
    if inst.keep:
      line('  %s %s[%d][%d][%d];' % (builtin_type(inst.c_type), inst.ident,
                                     inst.keep, R, C))
  for inst in insts:
    for k in range(inst.keep):
      for r in range(R):
        line('  ' + ' '.join('%s[%d][%d][%d] = 0;' % (inst.ident, k, r, c)
                             for c in range(C)))
  line('  i64 head = z0 - %d;' % lo[2])
  line('  const i64 steps = (z1 - z0) + %d;' % (L + lo[2]))
  line('  for (i64 n = 0; n < steps; n += %d, head += %d) {' % (period, period))

  def slot(inst, u, back):
    return (u - back) % inst.keep

  def operand(reader, src, rel, u, r, c):
    back = reader.lag - src.lag - rel[2]
    assert 0 <= back < src.keep, (reader.ident, src.ident, rel, back, src.keep)
    rr = min(max(r + rel[1], 0), R - 1)      # clamped rows are halo rows
    row = '%s[%d][%d]' % (src.ident, slot(src, u, back), rr)
    j = c + rel[0]
    if 0 <= j < C:
      return '%s[%d]' % (row, j)
    if j < 0:
      return 'from_lane_below(%s[%d])' % (row, C + j)
    return 'from_lane_above(%s[%d])' % (row, j - C)

  for u in range(period):
    line('    {  // unrolled step %d' % u)
    for inst in insts:
      if inst.stage is None:
        s = slot(inst, u, 0)
        line('      {  // load plane head+%d of %s' % (u, inst.tensor))
        line('        i64 zz = head + %d; if (zz > D - 1) zz = D - 1;' % u)
        line('        const %s* p = g_%s + zz * plane;' % (builtin_type(inst.c_type),
                                                          inst.tensor))
        line('        if (INTERIOR) {')
        for r in range(R):
          line('          { const %s v = *(const %s*)(p + row_y[%d] + x);%s }' % (
              vec_in, vec_in, r, ''.join(' %s[%d][%d][%d] = v[%d];' % (
                  inst.ident, s, r, c, c) for c in range(C))))
        line('        } else {')
        for r in range(R):
          for c in range(C):
            line('          %s[%d][%d][%d] = (x + %d >= 0 && x + %d < W) ? '
                 'p[row_y[%d] + x + %d] : (%s)0;' % (inst.ident, s, r, c, c, c, r, c,
                                                   builtin_type(inst.c_type)))
        line('        }')
        line('      }')
        continue
      stage = inst.stage
      ctype = builtin_type(inst.c_type)
      by_name = {}
      for src, rel, load_name in inst.reads:
        by_name[(load_name, rel)] = src
      if inst.final:
        line('      %s out_tile[%d][%d];' % (ctype, R, C))
      # rows whose whole dependency cone lies inside the tile; the others could
      # only produce halo garbage and are left at zero
      blo, bhi = stage_boxes[inst.iteration][stage['name']]
      rows_needed = range(-blo[1], R - bhi[1])
      for r in rows_needed:
        for c in range(C):
          def load(tensor, rel, u=u, r=r, c=c, inst=inst, by_name=by_name):
            return operand(inst, by_name[(tensor, tuple(rel))], tuple(rel), u, r, c)
          target = ('out_tile[%d][%d]' % (r, c)) if inst.final else \
              '%s[%d][%d][%d]' % (inst.ident, slot(inst, u, 0), r, c)
          cell_assignment(stage, target, load, line, '      ')
      if inst.final:
        line('      {  // store plane head+%d-%d' % (u, L))
        line('        const i64 z = head + %d;' % (u - L))
        line('        if (z >= z0 && z < z1) {')
        line('          %s* q = g_out + z * plane;' % T_out)
        for r in range(y_lo, R - y_hi):
          line('          if (yb + %d >= a.box_lo[1] && yb + %d < a.box_hi[1]) {'
               % (r, r))
          line('            if (x >= st_lo && x + %d <= st_hi) {' % C)
          line('              %s v;%s' % (vec_out, ''.join(
              ' v[%d] = out_tile[%d][%d];' % (c, r, c) for c in range(C))))
          if nt & 6:
            line('              if (NT) __builtin_nontemporal_store(v, (%s*)(q + row_y[%d] + x)); '
                 'else *(%s*)(q + row_y[%d] + x) = v;' % (vec_out, r, vec_out, r))
          else:
            line('              *(%s*)(q + row_y[%d] + x) = v;' % (vec_out, r))
          line('            } else {')
          for c in range(C):
            line('              if (x + %d >= st_lo && x + %d < st_hi) '
                 'q[row_y[%d] + x + %d] = out_tile[%d][%d];' % (c, c, r, c, r, c))
          line('            }')
          line('          }')
        line('        }')
        line('      }')
    line('    }')
  line('  }')
  line('}')
  line('')
  occupancy = ''
  if waves_per_eu > 0:
    occupancy = ' __attribute__((amdgpu_waves_per_eu(%d, %d)))' % (waves_per_eu,
                                                                   waves_per_eu)
  line('GLOBAL WG_SIZE(%d)%s void %s(soda_hip_args a) {'
       % (WAVES_PER_BLOCK * LANES, occupancy, name))
  line('  const int lane = lane_id();')
  line('  const int wave = __builtin_amdgcn_workitem_id_x() >> 6;')
  line('  const i64 x_origin = a.box_lo[0] - a.box_lo[0] %% %d;' % C)
  line('  const i64 strip = (i64)__builtin_amdgcn_workgroup_id_x() * %d + wave;'
       % WAVES_PER_BLOCK)
  line('  const i64 xs = x_origin + strip * %d;' % w_out)
  line('  if (xs >= a.box_hi[0]) return;')
  line('  const i64 x = xs - %d + lane * %d;' % (halo_lo, C))
  line('  const i64 yb = a.box_lo[1] + (i64)__builtin_amdgcn_workgroup_id_y() * %d'
       ' - %d;' % (r_out, y_lo))
  line('  const i64 chunk = a.param[0] > 0 ? a.param[0] : %d;' % chunk_planes)
  line('  const i64 z0 = a.box_lo[2] + (i64)__builtin_amdgcn_workgroup_id_z() * chunk;')
  line('  const i64 z1 = z0 + chunk < a.box_hi[2] ? z0 + chunk : a.box_hi[2];')
  line('  const bool interior = xs - %d >= 0 && xs - %d + %d <= a.dims[0];'
       % (halo_lo, halo_lo, LANES * C))
  if nt & 4:
    types = specmod.tensor_c_types(spec)
    cell_bytes = sum(specmod.ELEM_SIZE[types[n]] for n in
                     [t['name'] for t in spec['inputs']] + list(spec['outputs']))
    line('  const bool streaming = (a.box_hi[0] - a.box_lo[0]) * (a.box_hi[1] - a.box_lo[1]) '
         '* (a.box_hi[2] - a.box_lo[2]) * %d > %dll;' % (
             cell_bytes, kernel_common.NT_STREAMING_BYTES))
    line('  if (interior && streaming) %s_tile<true, true>(a, xs, x, yb, z0, z1);' % name)
    line('  else if (interior) %s_tile<true>(a, xs, x, yb, z0, z1);' % name)
  else:
    line('  if (interior) %s_tile<true>(a, xs, x, yb, z0, z1);' % name)
  line('  else %s_tile<false>(a, xs, x, yb, z0, z1);' % name)
  line('}')
  entry = dict(name=name, kind='fused', depth=depth, stage=-1,
               block=[WAVES_PER_BLOCK * LANES, 1, 1],
               tile=[WAVES_PER_BLOCK * w_out - C, r_out, chunk_planes, 1],
               fill_rows=L + lo[2], cols=C, rows=R, prefetch=prefetch,
               period=period, est_vgprs=est_vgprs, w_out=w_out, r_out=r_out)
  if nt:
    entry['nt'] = int(nt)
  return '\n'.join(o) + '\n', entry
