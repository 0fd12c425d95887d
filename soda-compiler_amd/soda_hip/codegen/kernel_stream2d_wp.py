"""Fused 2-D kernels, wave-pipelined form: the chain of stage instances of one
strip is cut into G consecutive groups and each group runs on its OWN wavefront
of the workgroup; group g hands the newest row of its last instance to group g+1
through a two-slot LDS buffer, one `s_barrier` per streamed row.

Why it exists: in the single-wave form (kernel_stream2d) a wavefront keeps the
windows of ALL `depth` levels, ~200 VGPRs at depth 12, so two waves fit a SIMD.
Here a wavefront keeps depth/G levels (~70 VGPRs at depth 12, G = 4) and six to
seven waves fit.  This is the reference's own structure one level up: SODA chains
compute modules with FIFOs (reference src/soda/dataflow.py:122-346); the modules
are wavefronts and the FIFOs are LDS rows.

Status: `kernel.generate` chooses this form where the single-wave form needs more
than 200 VGPRs or cannot be built (seidel2d from depth 12, blur from depth 8), and
for the depth-16 kernel of plain float programs (packed pairs + LDS ring, see
emit()).  Measured on MI355X, 16384x16384 (tools/tune.py): jacobi2d depth 12
single-wave 543-555 us per launch, this form 570-590 us, packed 545-570 us, packed
with the LDS ring 524 us; depth 16 packed with the ring 632 us = 39.5 us per
iteration against 46.0.  All forms are bit-exact against the oracle
(tests/test_gpu_parity.py, tools/check_variant.py).

Everything else is as in kernel_stream2d: lane l holds C consecutive columns,
x-neighbours by DPP wave shifts, windows rotated by unrolling, overlapped
strips/chunks, run-time chunk length.  A hand-off costs each side one 16-byte
LDS access per row; the barrier keeps the G wavefronts of a strip in lock step
(they do equal work by construction).

Scope: chains in which every group reads only itself and the LAST instance of
the previous group (jacobi2d, seidel2d, blur ...); other programs use the
single-wave form.
"""
import re

from . import spec as specmod
from .kernel_common import builtin_type, cell_assignment, device_expr, tensor_index
from .kernel_stream2d import LANES, Instance, NotFusable, geometry, kernel_name


def build_groups(spec, depth, prefetch, groups, loader=False, split=None, sync=1,
                 ring_lag=1):
  """Stage instances of `depth` iterations cut into `groups` wavefronts; the
  streamed dimension is the last one (rows in 2-D, planes in 3-D).

  loader=True: the program input does not come through the registers of group
  0 but through an LDS ring that a separate loader wavefront fills (role
  'ring_in' of the copy that group 0 reads, `ring_lag` steps behind the ring: 1
  when another wavefront fills it and a barrier lies in between, 0 when group 0
  fills the ring itself)."""
  axis = spec['dim'] - 1
  if len(spec['outputs']) != 1 or len(spec['inputs']) != 1:
    raise NotFusable('one input, one output')
  types = specmod.tensor_c_types(spec)
  if len({specmod.ELEM_SIZE[t] for t in types.values()}) != 1:
    raise NotFusable('mixed element widths')
  in_name = spec['inputs'][0]['name']
  source = Instance('in_%s' % in_name, in_name, 0, types[in_name])
  source.role, source.group = 'global_in', 0
  stages = []
  current = {in_name: source}
  for it in range(depth):
    for stage in spec['stages']:
      inst = Instance('k%d_%s' % (it, stage['name']), stage['name'], it,
                      stage['c_type'], stage)
      inst.role = 'compute'
      for tensor, rel in stage['loads']:
        inst.reads.append((current[tensor], tuple(rel), tensor))
      stages.append(inst)
      current[stage['name']] = inst
    current[in_name] = current[spec['outputs'][0]]
  final = current[spec['outputs'][0]]
  final.final = True
  if len(stages) < groups:
    raise NotFusable('%d stage instance(s) cannot feed %d wavefronts'
                     % (len(stages), groups))
  # contiguous groups of (nearly) equal size: equal VALU work per wavefront
  bounds = [round(i * len(stages) / groups) for i in range(groups + 1)]
  if split:
    # explicit number of stage instances per wavefront (the first and the last
    # wavefront also load / store: fewer levels there balance the four)
    if len(split) != groups or sum(split) != len(stages) or min(split) < 1:
      raise NotFusable('split %r does not cut %d stage instances into %d groups'
                       % (split, len(stages), groups))
    bounds = [sum(split[:i]) for i in range(groups + 1)]
  for g in range(groups):
    for inst in stages[bounds[g]:bounds[g + 1]]:
      inst.group = g
  last_of = [stages[bounds[g + 1] - 1] for g in range(groups)]
  # consumer-side copies of the hand-off instances
  copies = {}
  for g in range(1, groups):
    src = last_of[g - 1]
    copy = Instance('h%d_%s' % (g, src.ident), src.tensor, src.iteration, src.c_type)
    copy.role, copy.group, copy.origin = 'lds_in', g, src
    src.role = 'lds_out'
    src.handoff = g - 1
    copy.handoff = g - 1
    copies[id(src)] = copy
  ring = None
  if loader:
    ring = Instance('r_%s' % source.ident, source.tensor, 0, source.c_type)
    ring.role, ring.group, ring.origin = 'ring_in', 0, source
  for inst in stages:
    reads = []
    for src, rel, name in inst.reads:
      if src is source:
        if inst.group != 0:
          raise NotFusable('%s reads the program input from group %d'
                           % (inst.ident, inst.group))
        if loader:
          src = ring
      elif src.group != inst.group:
        if src is not last_of[inst.group - 1] or src.group != inst.group - 1:
          raise NotFusable('%s reads %s across wavefront groups'
                           % (inst.ident, src.ident))
        src = copies[id(src)]
      reads.append((src, rel, name))
    inst.reads = reads
  # lags, in execution order; a hand-off arrives one barrier interval (`sync`
  # steps) after it was produced
  source.lag = 0
  for inst in stages:
    lag = None
    for src, rel, _ in inst.reads:
      if src.role in ('lds_in', 'ring_in'):
        src.lag = src.origin.lag + (sync if src.role == 'lds_in' else ring_lag)
      v = src.lag + rel[axis] + (prefetch if src is source else 0)
      lag = v if lag is None else max(lag, v)
    inst.lag = lag
  everything = [source] + stages + list(copies.values()) + ([ring] if loader else [])
  for inst in stages:
    for src, rel, _ in inst.reads:
      src.keep = max(src.keep, inst.lag - rel[axis] - src.lag + 1)
  for inst in stages:
    if inst.role == 'compute' and not inst.final and inst.keep == 0:
      raise NotFusable('stage %s is never read' % inst.tensor)
  per_wave = []
  for g in range(groups):
    mine = [ring if loader else source] if g == 0 else [copies[id(last_of[g - 1])]]
    mine += stages[bounds[g]:bounds[g + 1]]
    per_wave.append(mine)
  return everything, per_wave, final


PLAIN_FLOAT_EXPR = re.compile(
    r'^(?:\{[^}]*\}|(?:\d+\.?\d*|\.\d+)(?:[eE][+-]?\d+)?f|\d+|[-+*/() ])*$')


def packable(spec):
  """True when every value of the program is a float and every expression is
  made of loads, + - * /, float-suffixed or integer literals only: such a program
  computes the same bits on <2 x float> operands (v_pk_add_f32 / v_pk_mul_f32 are
  IEEE like their scalar forms; a double literal would promote a scalar
  expression but not a vector one, so it disqualifies)."""
  if any(t != 'float' for t in specmod.tensor_c_types(spec).values()):
    return False
  for stage in spec['stages']:
    if stage['lets']:
      return False
    if not PLAIN_FLOAT_EXPR.match(device_expr(stage['expr'])):
      return False
  return True


def emit(spec, depth, cols=None, chunk_rows=256, prefetch=3, groups=4,
         max_period=12, vgpr_budget=120, skip_fill=1, pairs=0, align='none',
         ring=0, waves_per_eu=0, split=None, prio=None, seam_probe=0):
  """Returns (text, kernel table entry).

  pairs=2 (needs the ring): ONE strip of 2 x 64 x C columns per wavefront; a
  lane holds 2C consecutive columns, the first C in the low halves of its pairs
  and the last C in the high halves.  Same instruction count as pairs=1 with
  dppadd (the two operands per row that cross a half are two scalar adds, one of
  them DPP), but the x halo is paid once per 512 columns instead of per 256
  (jacobi2d depth 16: 480 of 512 columns kept against 448).

  With pairs=1 the lane-crossing operands stay two scalars so that each shift folds
  into a scalar add (kernel_common: pk2_shifted; jacobi2d depth 16: 627 -> 610 us).

  Measured and removed (docs/DESIGN_HISTORY.md 4.1a, 4.1d; all bit-exact, none a gain twice): one
  barrier per 2 or 3 rows instead of one per row (607 vs 602 us per depth-16 launch),
  wavefront roles rotated per workgroup (549 vs 543), non-temporal ring loads and
  stores of the deep kernels (+-0.4 % of cfg4), the ring read as two ds_read_b128 plus
  eight moves instead of four ds_read2_b32 (+0.5 %), per-lane store predicates per row
  instead of one decision per strip (589 vs 581).

  ring=N (a divisor of the rotation period, >= 3): the first wavefront does not
  prefetch input rows into registers; it streams them into an N-slot LDS ring
  with LDS-direct loads (global_load_lds_dwordx4, N-2 rows in flight, no VGPRs)
  and reads each row back when its turn comes.  LDS-direct loads cannot be
  guarded per element, so strips that would overhang the array are moved inside
  it (they still store only their own columns) and the kernel declares the
  narrowest array it accepts (min_extent).

  seam_probe=1 (wide form only; a TIMING PROBE, never shipped - tools/chunk_sweep.py
  '24;seam_probe=1', profiles/r06_seam_probe.txt): what exchanging the x halo between
  neighbouring strips instead of recomputing it would ADD per level-row, executed for real
  but feeding nothing: at the start of a step one ds_read_b32 per level (the neighbour's
  edge value of the row before, written before the barrier) and one wait; per level one
  VALU operation that stands for inserting it into lane 0 / 63 behind the DPP shift, one
  v_cndmask that picks this lane's edge column (column 0 in lane 0, column 2C-1 in lane 63:
  two different registers) and one ds_write_b32.  Four wave-instructions per level-row on
  top of 22; the results are the unchanged kernel's (the strips still overlap), so the
  probe prices the cost side only - the gain side is 512 / 464 columns at depth 24.

  pairs=1 (float programs, see packable()): a wavefront streams TWO adjacent
  strips at once, element c of strip A and element c of strip B sharing one
  64-bit register pair, so that every add and multiply is a packed
  v_pk_add_f32 / v_pk_mul_f32 (two results per issue slot: measured 68 T
  lane-ops/s against 37.5 T for the scalar forms, tools/microbench.hip).  Both
  halves have the same neighbours, hence no operand ever straddles a pair."""
  types = specmod.tensor_c_types(spec)
  index = tensor_index(spec)
  in_type = spec['inputs'][0]['c_type']
  out_name = spec['outputs'][0]
  elem = specmod.ELEM_SIZE[in_type]
  if cols is None:
    cols = max(1, 16 // elem)
  C = cols
  if pairs and not packable(spec):
    raise NotFusable('packed form: float programs of + - * / only')
  if pairs and (C * elem) % 16:
    raise NotFusable('packed form: whole 16-byte vectors per lane')
  P = 2 if pairs else 1
  wide = int(pairs) == 2
  if spec['dim'] != 2:
    raise NotFusable('2-D programs only')
  RS = int(ring)
  PF = RS - 2
  if RS and (RS < 3 or C * elem != 16):
    raise NotFusable('input ring: >= 3 slots, 16-byte lanes')
  if wide and not RS:
    raise NotFusable('wide strips come through the input ring')
  if isinstance(split, str):
    split = [int(v) for v in split.split('/')]
  everything, per_wave, final = build_groups(spec, depth, 0 if RS else prefetch,
                                             groups, split=split)
  S = 1       # rows per workgroup barrier (and per hand-off slot)
  dppadd = int(pairs) == 1
  geo = geometry(spec, depth, C, chunk_rows, align)
  if wide:      # the second half adds 64 x C columns, all of them output
    geo['w_out'] += LANES * C
  strip_cols = LANES * C * (2 if wide else 1)
  for inst in everything:
    for src, rel, _ in inst.reads:
      if abs(rel[0]) > C:
        raise NotFusable('x offset %d exceeds the %d columns a lane holds'
                         % (rel[0], C))
  # one rotation period for the whole workgroup (the barrier count per loop
  # trip must be the same in every wavefront); even, for the two LDS slots
  best = None
  for candidate in range(2, max_period + 1, 2):
    if max(i.keep for i in everything) > candidate:
      continue
    if RS and candidate % RS:
      continue
    if candidate % (2 * S):
      continue
    divisors = [d for d in range(1, candidate + 1) if candidate % d == 0]
    padded = [min(d for d in divisors if d >= i.keep) if i.keep else 0
              for i in everything]
    cost = (sum(padded), candidate)
    if best is None or cost < best[0]:
      best = (cost, candidate, padded)
  if best is None:
    raise NotFusable('windows exceed the rotation period limit')
  period = best[1]
  for inst, keep in zip(everything, best[2]):
    inst.keep = keep
  per_elem = max(1, elem // 4)
  est_vgprs = max(sum(i.keep for i in mine) for mine in per_wave) * C * per_elem * P + \
      4 * C * P + 20
  if est_vgprs > vgpr_budget:
    raise NotFusable('a wavefront would need about %d VGPRs (budget %d)'
                     % (est_vgprs, vgpr_budget))
  # first step at which an instance can matter (see kernel_stream2d.emit)
  below = {id(final): 0}
  order = [i for mine in per_wave for i in mine]
  for inst in reversed(order):
    need = below.get(id(inst))
    if need is None:
      continue
    if inst.role == 'lds_in':
      below[id(inst.origin)] = max(below.get(id(inst.origin), -10**9), need)
    for src, rel, _ in inst.reads:
      below[id(src)] = max(below.get(id(src), -10**9), need - rel[1])
  for inst in everything:
    inst.first_step = max(0, inst.lag + geo['y_lo'] - below.get(id(inst), 0))
  name = kernel_name(spec, depth)
  L = final.lag
  T_in = builtin_type(in_type)
  T_out = builtin_type(types[out_name])
  vec = 'vec_%s' % name
  o = []
  line = o.append
  line('// fused depth-%d kernel, wave-pipelined: %d wavefronts per strip, rotation '
       'period %d,' % (depth, groups, period))
  line('// strip = %d columns (%d out + halo %d/%d), prefetch %d rows, ~%d VGPRs'
       % (strip_cols, geo['w_out'], geo['halo_lo'], geo['halo_hi'], prefetch,
          est_vgprs))
  for g, mine in enumerate(per_wave):
    for inst in mine:
      line('//   wave %d  %-20s lag %2d keep %2d  %s' % (
          g, inst.ident, inst.lag, inst.keep,
          '-> HBM' if inst.final else inst.role))
  line('typedef %s %s __attribute__((ext_vector_type(%d), aligned(%d)));'
       % (T_in, vec, C, elem))
  line('typedef %s %s_lds __attribute__((ext_vector_type(%d), aligned(%d)));'
       % (T_in, vec, C, C * elem))
  if pairs:
    line('typedef float pk2 __attribute__((ext_vector_type(2)));')
  # LDS hand-off rows: 16-byte pieces, piece q of lane l at [q][l] (conflict-free)
  pieces = C * P * elem // 16 if pairs else 1
  per_piece = C * P // pieces

  def slot(inst, u, back):
    return (u - back) % inst.keep

  def operand(reader, src, rel, u, c):
    back = reader.lag - src.lag - rel[1]
    assert 0 <= back < src.keep, (reader.ident, src.ident, rel, back, src.keep)
    row = '%s[%d]' % (src.ident, slot(src, u, back))
    j = c + rel[0]
    if 0 <= j < C:
      return '%s[%d]' % (row, j)
    if wide:
      return 'pk_wide_%s(%s[%d])' % (('below', row, C + j) if j < 0 else
                                     ('above', row, j - C))
    shift = 'pk_from_lane' if pairs and dppadd else 'from_lane'
    if j < 0:
      return '%s_below(%s[%d])' % (shift, row, C + j)
    return '%s_above(%s[%d])' % (shift, row, j - C)

  def vmcnt(n):      # s_waitcnt immediate: vmcnt(n), other counters untouched
    return (n & 15) | (7 << 4) | (15 << 8) | ((n >> 4) << 14)

  def ring_issue(row_expr, slot_index):
    line('        { i64 row = %s; if (row > H - 1) row = H - 1;' % row_expr)
    for h in range(P):
      line('          __builtin_amdgcn_global_load_lds((const __attribute__(('
           'address_space(1))) void*)(g_in + row * W + %s), (__attribute__(('
           'address_space(3))) void*)&in_ring[%d][%d][0], 16, 0, %d);'
           % (('x - lane * %d + %d' % (C, h * LANES * C)) if wide else
              ('xb' if h else 'x'), slot_index, h, 0))
    line('        }')

  def emit_body(mine, guarded):
    probed = [i for i in mine if i.role in ('compute', 'lds_out') or i.final] if seam_probe \
        else []
    for u in range(period):
      line('      {  // unrolled step %d' % u)
      for inst in mine:
        if inst.role == 'global_in' and RS:
          s = slot(inst, u, 0)
          ring_issue('head + %d' % (u + PF), (u + PF) % RS)
          # row head+u was issued PF rows ago
          line('        __builtin_amdgcn_s_waitcnt(%d);  // vmcnt(%d)'
               % (vmcnt(PF * P), PF * P))
          if wide and C == 4:
            # the four pairs straight from the LDS read (ds_read2_b32) instead of
            # two ds_read_b128 and eight register moves: 0.4-0.7 % per depth-16
            # launch in four paired runs (552 vs 555 us)
            line('        soda_lds_read_pairs4(&in_ring[%d][0][0] + lane * 8, %s);'
                 % (u % RS, ', '.join('%s[%d][%d]' % (inst.ident, s, c)
                                      for c in range(C))))
            continue
          for h in range(P):
            line('        const %s_lds ring_v%d = __builtin_bit_cast(%s_lds, '
                 'soda_lds_read_f4(&in_ring[%d][%d][0] + lane * %d + %d));'
                 % (vec, h, vec, u % RS, 0 if wide else h, C * (2 if wide else 1),
                    C * h if wide else 0))
          for c in range(C):
            line('        %s[%d][%d] = %s;' % (
                inst.ident, s, c,
                'pk2{ring_v0[%d], ring_v1[%d]}' % (c, c) if pairs else 'ring_v0[%d]' % c))
          continue
        if inst.role == 'global_in':
          s = slot(inst, u, 0)
          line('        { i64 row = head + %d; if (row > H - 1) row = H - 1;' % u)
          line('          const %s* p = g_in + row * W + x;' % T_in)
          if pairs:
            line('          if (INTERIOR) { const %s va = *(const %s*)p, vb = *(const %s*)'
                 '(p + %d);%s }' % (vec, vec, vec, geo['w_out'], ''.join(
                     ' %s[%d][%d] = pk2{va[%d], vb[%d]};' % (inst.ident, s, c, c, c)
                     for c in range(C))))
            line('          else {%s } }' % ''.join(
                ' %s[%d][%d] = pk2{(x + %d >= 0 && x + %d < W) ? p[%d] : 0.0f, '
                '(x + %d >= 0 && x + %d < W) ? p[%d] : 0.0f};'
                % (inst.ident, s, c, c, c, c, c + geo['w_out'], c + geo['w_out'],
                   c + geo['w_out']) for c in range(C)))
            continue
          line('          if (INTERIOR) { const %s v = *(const %s*)p;%s }' % (
              vec, vec, ''.join(' %s[%d][%d] = v[%d];' % (inst.ident, s, c, c)
                                for c in range(C))))
          line('          else {%s } }' % ''.join(
              ' %s[%d][%d] = (x + %d >= 0 && x + %d < W) ? p[%d] : (%s)0;'
              % (inst.ident, s, c, c, c, c, T_in) for c in range(C)))
          continue
        if inst.role == 'lds_in':
          s = slot(inst, u, 0)
          # written by the previous wavefront one step ago: the other slot
          if pairs:
            for q in range(pieces):
              line('        { const soda_f4 v = *(const soda_f4*)&handoff[%d][%d][%d][%d][lane * 4];%s }'
                   % (inst.handoff, (u // S + 1) % 2, u % S, q, ''.join(
                       ' %s[%d][%d] = pk2{v[%d], v[%d]};' % (
                           inst.ident, s, q * 2 + j, 2 * j, 2 * j + 1)
                       for j in range(2))))
            continue
          line('        { const %s_lds v = *(const %s_lds*)&handoff[%d][%d][%d][lane * %d];%s }'
               % (vec, vec, inst.handoff, (u // S + 1) % 2, u % S, C, ''.join(
                   ' %s[%d][%d] = v[%d];' % (inst.ident, s, c, c) for c in range(C))))
          continue
        stage = inst.stage
        ctype = 'pk2' if pairs else builtin_type(inst.c_type)
        by_name = {(n, rel): src for src, rel, n in inst.reads}
        skip = guarded and inst.first_step > u
        if skip:
          line('        if (n + %d >= %d) {' % (u, inst.first_step))
        direct = inst.keep == 0
        if direct:
          line('        %s out_row[%d];' % (ctype, C))
        for c in range(C):
          def load(tensor, rel, u=u, c=c, inst=inst, by_name=by_name):
            return operand(inst, by_name[(tensor, tuple(rel))], tuple(rel), u, c)
          target = ('out_row[%d]' % c) if direct else \
              '%s[%d][%d]' % (inst.ident, slot(inst, u, 0), c)
          cell_assignment(stage, target, load, line, '        ')
        if inst in probed:
          k = probed.index(inst)
          row = 'out_row' if direct else '%s[%d]' % (inst.ident, slot(inst, u, 0))
          # the neighbour strip's edge value of the row before (written before the barrier):
          # read, waited for and consumed in one go - an asm statement's outputs must be
          # complete when it ends (a read left in flight past its statement is a register
          # the compiler may have given to something else: the first version of this probe
          # faulted), so the probe pays the LDS latency where a real implementation would
          # issue the read a level ahead and pay a register for it
          line('        { float seam_in;')
          line('          asm volatile("ds_read_b32 %%0, %%1 offset:%d\\n\\ts_waitcnt lgkmcnt(0)" '
               ': "=v"(seam_in) : "v"(seam_rd + %du));' % (k * 256, ((u + 1) % 2) * 2048))
          line('          seam_acc += seam_in; }')
          line('        { const float e = seam_first ? %s[0][0] : %s[%d][1];' % (row, row, C - 1))
          line('          asm volatile("ds_write_b32 %%0, %%1 offset:%d" :: "v"(seam_wr + %du), '
               '"v"(e)); }' % (k * 256, (u % 2) * 2048))
        if inst.role == 'lds_out' and pairs:
          for q in range(pieces):
            line('        { soda_f4 v;%s *(soda_f4*)&handoff[%d][%d][%d][%d][lane * 4] = v; }' % (
                ''.join(' v[%d] = out_row[%d][0]; v[%d] = out_row[%d][1];' % (
                    2 * j, q * 2 + j, 2 * j + 1, q * 2 + j) for j in range(2)),
                inst.handoff, (u // S) % 2, u % S, q))
        elif inst.role == 'lds_out':
          line('        { %s_lds v;%s *(%s_lds*)&handoff[%d][%d][%d][lane * %d] = v; }' % (
              vec, ''.join(' v[%d] = out_row[%d];' % (c, c) for c in range(C)),
              vec, inst.handoff, (u // S) % 2, u % S, C))
        if inst.final:
          line('        { const i64 y = head + %d;' % (u - L))
          line('          if (y >= y0 && y < y1) {')
          # common case, decided once per strip: no lane's vector straddles an
          # edge of the store range, so a lane stores its whole vector or nothing
          line('            if (!ragged) {')
          for half in range(P):
            sel = '[%d]' % half if pairs else ''
            xv = 'xb' if half else 'x'
            line('              if (full_%d) { %s v;%s *(%s*)(g_out + y * W + %s) = v; }'
                 % (half, vec, ''.join(' v[%d] = out_row[%d]%s;' % (c, c, sel)
                                       for c in range(C)), vec, xv))
          line('            } else {')
          for half in range(P):
            sel = '[%d]' % half if pairs else ''
            sfx = 'b' if half and not wide else ''
            xv = 'xb' if half else 'x'
            line('            { %s* q = g_out + y * W + %s;' % (T_out, xv))
            line('            if (%s >= st_lo%s && %s + %d <= st_hi%s) { %s v;%s '
                 '*(%s*)q = v; }'
                 % (xv, sfx, xv, C, sfx, vec, ''.join(
                     ' v[%d] = out_row[%d]%s;' % (c, c, sel) for c in range(C)), vec))
            line('            else {%s } }' % ''.join(
                ' if (%s + %d >= st_lo%s && %s + %d < st_hi%s) q[%d] = out_row[%d]%s;'
                % (xv, c, sfx, xv, c, sfx, c, c, sel) for c in range(C)))
          line('            }')
          line('          } }')
        if skip:
          line('        }')
      line('      }')
      if (u + 1) % S == 0:
        line('      %s();' % ('soda_lds_barrier' if RS else 'soda_block_barrier'))

  line('template <bool INTERIOR>')
  line('DEV void %s_strip(const soda_hip_args& a, const i64 xs, const i64 x, '
       'const i64 xb, const i64 y0, const i64 y1, const int wave, const int lane,'
       % name)
  ring_dims = (P, LANES * C) if RS else (1, 1)
  if pairs and seam_probe:
    line('    float (*handoff)[2][%d][%d][%d], %s (*in_ring)[%d][%d], float* seam_lds) {'
         % (S, pieces, LANES * 4, T_in, ring_dims[0], ring_dims[1]))
  elif pairs:
    line('    float (*handoff)[2][%d][%d][%d], %s (*in_ring)[%d][%d]) {'
         % (S, pieces, LANES * 4, T_in, ring_dims[0], ring_dims[1]))
  else:
    line('    %s (*handoff)[2][%d][%d], %s (*in_ring)[%d][%d]) {'
         % (T_in, S, LANES * C, T_in, ring_dims[0], ring_dims[1]))
  line('  const i64 W = a.dims[0], H = a.dims[1];')
  line('  const i64 st_lo = xs > a.box_lo[0] ? xs : a.box_lo[0];')
  line('  const i64 st_hi = xs + %d < a.box_hi[0] ? xs + %d : a.box_hi[0];'
       % (geo['w_out'], geo['w_out']))
  if pairs and not wide:   # the second strip starts where the first one ends
    line('  const i64 st_lob = xs + %d;' % geo['w_out'])
    line('  const i64 st_hib = xs + %d < a.box_hi[0] ? xs + %d : a.box_hi[0];'
         % (2 * geo['w_out'], 2 * geo['w_out']))
    line('  (void)st_lob; (void)st_hib;')
  line('  const %s* __restrict__ g_in = (const %s*)a.tensor[%d];'
       % (T_in, T_in, index[spec['inputs'][0]['name']]))
  line('  %s* __restrict__ g_out = (%s*)a.tensor[%d];' % (T_out, T_out,
                                                           index[out_name]))
  line('  (void)g_in; (void)g_out; (void)st_lo; (void)st_hi; (void)W; (void)H; '
       '(void)xb; (void)in_ring;')
  parts = []
  for half in range(P):
    sfx = 'b' if half and pairs and not wide else ''
    xv = 'xb' if half else 'x'
    line('  const bool full_%d = %s >= st_lo%s && %s + %d <= st_hi%s; (void)full_%d;'
         % (half, xv, sfx, xv, C, sfx, half))
    parts.append('(!full_%d && %s + %d > st_lo%s && %s < st_hi%s)'
                 % (half, xv, C, sfx, xv, sfx))
  line('  const bool ragged = __builtin_amdgcn_ballot_w64(%s) != 0; (void)ragged;'
       % ' || '.join(parts))
  line('  const i64 steps = (y1 - y0) + %d;' % (L + geo['y_lo']))
  if seam_probe:
    # per wavefront 2 parities x 8 levels x 64 lanes of floats; a lane reads what lane 63 - l
    # wrote (the other edge)
    line('  float seam_acc = 0.0f;')
    line('  const bool seam_first = lane == 0;')
    line('  const unsigned seam_wr = (unsigned)(unsigned long long)seam_lds + wave * 4096u + '
         'lane * 4u;')
    line('  const unsigned seam_rd = (unsigned)(unsigned long long)seam_lds + wave * 4096u + '
         '(63 - lane) * 4u;')
  prologue_steps = max(i.first_step for i in everything)
  prologue_steps = -(-prologue_steps // period) * period if skip_fill else 0
  for g, mine in enumerate(per_wave):
    line('  %sif (wave == %d) {' % ('' if g == 0 else 'else ', g))
    # Issue priority per wavefront ('a/b/c/d'; default: the first one raised).
    # With equal priorities the four wavefronts of a SIMD (one per workgroup, all
    # at the same pipeline stage) compete round-robin and stall together; any
    # inequality staggers them.  Measured per launch on 16384^2: jacobi2d depth
    # 16 561 -> 550 us (the same for 3/0/0/0, 0/0/0/3, 3/2/1/0; 3/3/3/3 = no
    # change), seidel2d depth 16 619 -> 605, blur depth 12 606 -> 577.
    levels = [3] + [0] * (groups - 1) if prio is None else \
        ([int(v) for v in (prio if isinstance(prio, (list, tuple)) else str(prio).split('/'))] + [0] * groups)[:groups]
    if levels[g]:
      line('    __builtin_amdgcn_s_setprio(%d);' % levels[g])
    for inst in mine:
      if inst.keep:
        line('    %s %s[%d][%d];' % ('pk2' if pairs else builtin_type(inst.c_type),
                                     inst.ident, inst.keep, C))
        for r in range(inst.keep):
          line('    ' + ' '.join('%s[%d][%d] = %s;' % (
              inst.ident, r, c, 'pk2{0.0f, 0.0f}' if pairs else '0')
                                 for c in range(C)))
    line('    i64 head = y0 - %d;' % geo['y_lo'])
    line('    i64 n = 0;')
    if RS and g == 0:      # the first PF rows of the ring
      for k in range(PF):
        ring_issue('head + %d' % k, k)
    if prologue_steps:
      line('    for (; n < %d && n < steps; n += %d, head += %d) {'
           % (prologue_steps, period, period))
      emit_body(mine, True)
      line('    }')
    line('    for (; n < steps; n += %d, head += %d) {' % (period, period))
    emit_body(mine, False)
    line('    }')
    if RS and g == 0:
      line('    __builtin_amdgcn_s_waitcnt(%d);  // no load may outlive the LDS'
           % vmcnt(0))
    line('  }')
  if seam_probe:      # keeps the probe's values alive; never true
    line('  if (seam_acc == 1.2345e30f) g_out[0] = seam_acc;')
  line('}')
  line('')
  occupancy = ''
  if waves_per_eu > 0:
    occupancy = ' __attribute__((amdgpu_waves_per_eu(%d, %d)))' % (waves_per_eu,
                                                                   waves_per_eu)
  line('GLOBAL WG_SIZE(%d)%s void %s(soda_hip_args a) {' % (groups * LANES, occupancy,
                                                             name))
  if pairs:
    line('  __attribute__((shared)) float handoff[%d][2][%d][%d][%d];' % (
        max(1, groups - 1), S, pieces, LANES * 4))
  else:
    line('  __attribute__((shared)) %s handoff[%d][2][%d][%d];' % (
        T_in, max(1, groups - 1), S, LANES * C))
  line('  __attribute__((shared)) %s in_ring[%d][%d][%d];' % (
      T_in, RS if RS else 1, ring_dims[0], ring_dims[1]))
  if seam_probe:
    if not wide:
      raise NotFusable('seam_probe: the wide packed form only')
    line('  __attribute__((shared)) float seam_lds[%d];' % (groups * 1024))
  line('  const int lane = lane_id();')
  line('  const int wave = __builtin_amdgcn_readfirstlane('
       '__builtin_amdgcn_workitem_id_x() >> 6);')
  line('  const i64 x_origin = a.box_lo[0] - a.box_lo[0] %% %d;' % geo['origin_align'])
  tile_cols = geo['w_out'] * (1 if wide or not pairs else 2)
  line('  const i64 xs = x_origin + (i64)__builtin_amdgcn_workgroup_id_x() * %d;'
       % tile_cols)
  line('  if (xs >= a.box_hi[0]) return;')
  line('  const i64 chunk = a.param[0] > 0 ? a.param[0] : %d;' % chunk_rows)
  line('  const i64 y0 = a.box_lo[1] + (i64)__builtin_amdgcn_workgroup_id_y() * chunk;')
  line('  const i64 y1 = y0 + chunk < a.box_hi[1] ? y0 + chunk : a.box_hi[1];')
  if RS:
    # strips that would overhang the array are moved inside it (no guarded loads)
    for v, start in (('wx', 'xs - %d' % geo['halo_lo']),
                     ('wxb', 'xs + %d' % (geo['w_out'] - geo['halo_lo']))):
      if wide and v == 'wxb':
        continue
      line('  i64 %s = %s;' % (v, start))
      line('  if (%s + %d > a.dims[0]) %s = a.dims[0] - %d;' % (v, strip_cols, v, strip_cols))
      line('  if (%s < 0) %s = 0;' % (v, v))
    if wide:    # the lane's 2C columns; xb = where its high halves start
      line('  const i64 x = wx + lane * %d, xb = x + %d;' % (2 * C, C))
    else:
      line('  const i64 x = wx + lane * %d, xb = wxb + lane * %d;' % (C, C))
    line('  %s_strip<true>(a, xs, x, xb, y0, y1, wave, lane, handoff, in_ring%s);' % (
        name, ', seam_lds' if seam_probe else ''))
  else:
    line('  const i64 x = xs - %d + lane * %d;' % (geo['halo_lo'], C))
    line('  const i64 xb = x + %d;' % geo['w_out'])
    line('  const bool interior = xs - %d >= 0 && xs - %d + %d <= a.dims[0];'
         % (geo['halo_lo'], geo['halo_lo'], LANES * C + (P - 1) * geo['w_out']))
    line('  if (interior) %s_strip<true>(a, xs, x, xb, y0, y1, wave, lane, handoff, '
         'in_ring);' % name)
    line('  else %s_strip<false>(a, xs, x, xb, y0, y1, wave, lane, handoff, in_ring);'
         % name)
  line('}')
  entry = dict(name=name, kind='fused', depth=depth, stage=-1,
               block=[groups * LANES, 1, 1],
               tile=[tile_cols, chunk_rows, 1, 1], pairs=int(pairs),
               ring=RS, min_extent=[strip_cols, 1] if RS else [0, 0],
               origin_align=geo['origin_align'],
               fill_rows=L + geo['y_lo'], cols=C, prefetch=prefetch, period=period,
               est_vgprs=est_vgprs, groups=groups, w_out=geo['w_out'])
  return '\n'.join(o) + '\n', entry
