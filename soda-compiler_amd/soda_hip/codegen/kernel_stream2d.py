"""Fused 2-D kernels: all stages of `depth` iterations in one launch, streaming
along the outer dimension with every intermediate kept in registers.

This is the GPU counterpart of the micro-architecture the reference synthesises
for an FPGA (line buffers + one compute module per stage per iteration, chained;
reference src/soda/core.py:612-777, src/soda/dataflow.py:122-346, README
"iterate" = replicated pipeline stages): the chain of stage instances is the
same, but the "line buffers" are per-lane register windows and the "FIFOs between
modules" are the program order of one wavefront.

Mapping (gfx950, wave64):
  * one wavefront owns a strip of 64*C columns; lane l holds C consecutive
    columns of every live row, so a row load is one coalesced `C*sizeof(T)`-byte
    vector load per lane (1 KiB per wave for 16-byte vectors);
  * x-neighbours inside a lane are other registers; across lanes they come from
    lane-1 / lane+1 through DPP wave shifts fused into the consuming VALU op --
    no LDS, no barrier, nothing shared between wavefronts;
  * the wavefront walks down its strip one input row per step.  Stage instance S
    trails the load head by lag(S) rows and keeps keep(S) rows for its readers
    (the register analogue of the reference's reuse-buffer lengths); the y loop
    is unrolled by the rotation period so that "shifting" a window is renaming;
  * strips overlap by the composed x-window of `depth` iterations and y-chunks
    by the composed y-window (overlapped tiling: halo cells are recomputed, never
    exchanged), so HBM sees one read and one write per cell per `depth` updates.

Cost model per cell-update (jacobi2d): 5 VALU lane-ops, 8/depth bytes of HBM.
"""

from . import kernel_common
from . import spec as specmod
from .kernel_common import builtin_type, cell_assignment, device_expr, tensor_index

WAVES_PER_BLOCK = 4
LANES = 64


class NotFusable(Exception):
  """The program is outside what this generator covers; per-stage kernels
  remain available."""


class Instance:
  """One tensor of one iteration inside the fused pipeline."""

  def __init__(self, ident, tensor, iteration, c_type, stage=None):
    self.ident = ident          # C identifier stem
    self.tensor = tensor
    self.iteration = iteration
    self.c_type = c_type
    self.stage = stage          # spec stage dict, None for a loaded input
    self.reads = []             # [(Instance, (dx, dy), name in expr)], unique
    self.lag = 0                # rows behind the load head
    self.keep = 0               # rows retained for readers (0 = stored directly)
    self.final = False          # last iteration's output: goes to HBM


def build_pipeline(spec, depth, prefetch):
  if spec['dim'] != 2:
    raise NotFusable('stream2d handles 2-D programs')
  if len(spec['outputs']) != 1:
    raise NotFusable('stream2d handles single-output programs')
  ins = [t['name'] for t in spec['inputs']]
  if depth > 1 and len(ins) != 1:
    raise NotFusable('depth > 1 needs one input feeding one output')
  types = specmod.tensor_c_types(spec)
  for name, ctype in types.items():
    if specmod.ELEM_SIZE[ctype] not in (1, 2, 4, 8):
      raise NotFusable('element size of %s' % name)
  insts = []
  current = {}
  for name in ins:
    inst = Instance('in_%s' % name, name, 0, types[name])
    insts.append(inst)
    current[name] = inst
  for it in range(depth):
    for stage in spec['stages']:
      inst = Instance('k%d_%s' % (it, stage['name']), stage['name'], it,
                      stage['c_type'], stage)
      for tensor, rel in stage['loads']:
        inst.reads.append((current[tensor], tuple(rel), tensor))
      insts.append(inst)
      current[stage['name']] = inst
    # output j feeds input j of the next iteration
    for i, o in zip(ins, spec['outputs']):
      if len(ins) == len(spec['outputs']):
        current[i] = current[o]
  final = current[spec['outputs'][0]]
  final.final = True
  # lags: a reader can produce row y once every row y+dy it reads exists;
  # rows of loaded inputs count as existing `prefetch` steps after their load
  for inst in insts:
    if inst.stage is None:
      inst.lag = 0
      continue
    inst.lag = max(src.lag + rel[1] + (prefetch if src.stage is None else 0)
                   for src, rel, _ in inst.reads)
  for inst in insts:
    for src, rel, _ in inst.reads:
      src.keep = max(src.keep, inst.lag - rel[1] - src.lag + 1)
  for inst in insts:
    if not inst.final and inst.keep == 0:
      raise NotFusable('stage %s is never read' % inst.tensor)
  return insts, final


def geometry(spec, depth, cols, chunk_rows, align='none'):
  """Strip/chunk geometry for `depth` fused iterations.

  align: 'none'  - strips as wide as the halo allows;
         'store' - every strip's OUTPUT columns start and end on a 128-byte line
                   (no line is written by two wavefronts);
         'full'  - the loaded columns start on a line as well;
         'exact' - strips do not overlap at all: a wavefront loads and stores exactly
                   its own 64 * cols columns, whole 128-byte lines on both sides, and
                   the columns its first and last lane need from beyond the strip come
                   from one extra vector load each per input row (`emit`, edge loads).
                   Only for kernels whose lane-crossing reads are all on loaded inputs
                   (depth 1 of the samples): a stage read across lanes would have to be
                   computed beyond the strip, which is what the overlap is for.
  Measured with the kernels' access pattern on a 16384x16384 float array
  (tools/copybench.hip, strips of 256 loaded columns): 248 out / halo 4 508 us,
  232/12 520 us, 224/16 487 us, 224/0 (both aligned) 445 us, 192/32 431 us,
  256/0 382 us - alignment is worth more than the extra halo reads for the
  kernels that HBM bounds."""
  margins = specmod.iteration_margins(spec, depth)
  if len(spec['inputs']) == 1 and len(spec['outputs']) == 1:
    lo, hi = margins[-1]
  else:
    lo, hi = margins[0]
  elem = specmod.ELEM_SIZE[spec['inputs'][0]['c_type']]
  line = max(cols, 128 // elem)
  line -= line % cols
  halo_lo = -(-lo[0] // cols) * cols      # padded up to whole vectors
  halo_hi = -(-hi[0] // cols) * cols
  if align == 'full':
    halo_lo = -(-lo[0] // line) * line
  if align == 'exact':
    halo_lo = halo_hi = 0
  w_out = LANES * cols - halo_lo - halo_hi
  origin_align = cols
  if align == 'exact':
    origin_align = line
  elif align in ('store', 'full') and w_out >= line:
    w_out -= w_out % line
    halo_hi = LANES * cols - halo_lo - w_out
    origin_align = line
  elif align not in ('none', 'store', 'full', 'exact'):
    raise ValueError('align: %r' % (align,))
  if w_out < cols:
    raise NotFusable('depth %d leaves no output columns in a strip' % depth)
  return dict(x_lo=lo[0], x_hi=hi[0], y_lo=lo[1], y_hi=hi[1],
              halo_lo=halo_lo, halo_hi=halo_hi, w_out=w_out,
              chunk_rows=chunk_rows, origin_align=origin_align)


def kernel_name(spec, depth):
  return '%s_fused_k%d' % (spec['app_name'], depth)


def emit(spec, depth, cols=None, chunk_rows=256, prefetch=3, max_period=12,
         vgpr_budget=244, waves_per_eu=0, skip_fill=1, nontemporal=0, align='none',
         steady=None, stage_edges=0, stream_wgs=None):
  """Returns (text, kernel table entry) for one fused depth.

  steady (default: on for depth <= 2, the memory-bound kernels): the rows of a chunk
  whose loads need no clamping and whose results are all stored run in a loop of their
  own without the per-row range checks, and whether a strip has lanes that store only
  some of their columns is decided once per strip (template parameter) instead of per
  row - per row that is ~45 instead of ~66 instructions for jacobi2d (the hand-written
  tools/k1bench.hip, the same kernel without any of the guards, runs the 16384^2 sweep
  in 430 us where the guarded loop takes 463-484)."""
  if steady is None:
    steady = depth <= 2
  steady = bool(steady) and bool(skip_fill)
  types = specmod.tensor_c_types(spec)
  index = tensor_index(spec)
  in_type = spec['inputs'][0]['c_type']
  out_name = spec['outputs'][0]
  elem = specmod.ELEM_SIZE[in_type]
  if any(specmod.ELEM_SIZE[t['c_type']] != elem for t in spec['inputs']) or \
      specmod.ELEM_SIZE[types[out_name]] != elem:
    raise NotFusable('inputs and output of different widths')
  if cols is None:
    cols = max(1, 16 // elem)
  insts, final = build_pipeline(spec, depth, prefetch)
  geo = geometry(spec, depth, cols, chunk_rows, align)
  C = cols
  for inst in insts:
    for src, rel, _ in inst.reads:
      if abs(rel[0]) > cols:
        raise NotFusable('x offset %d exceeds the %d columns a lane holds'
                         % (rel[0], cols))
  exact = align == 'exact'
  # the steady-state loop without any branch (round 4): the row's store is a raw buffer
  # store whose lane offset is out of range in lanes that store nothing (halo lanes of
  # overlapping strips), a scheduling fence per row, loaded rows kept packed - see emit_body
  flat = steady and C * elem in (4, 8, 16)
  for inst in insts:
    inst.edges = dict(lo=0, hi=0)    # columns beyond the strip that readers ask for, per side
  if exact:
    # which instances are needed beyond the strip (C columns on either side, held by the
    # first and the last lane): those read across lanes - and what THEY are computed
    # from, which must then be read at x offset 0 only (blur: blur_y reads blur_x across
    # lanes, blur_x reads the input straight up: both get edge vectors, the input's
    # loaded, blur_x's computed from them)
    for inst in reversed(insts):
      for src, rel, _ in inst.reads:
        src.edges['lo'] = max(src.edges['lo'], -rel[0])
        src.edges['hi'] = max(src.edges['hi'], rel[0])
      if any(inst.edges.values()) and inst.stage is not None:
        if not stage_edges:
          # (blur: blur_y reads blur_x across lanes.  Computing blur_x beyond the strip
          # works - bit-exact, `stage_edges=1` - and is no faster than overlapping
          # strips: 16384^2 222-242 us against 220; so such programs keep the overlap)
          raise NotFusable('seam-free strips: stage %s is read across lanes' % inst.tensor)
        for src, rel, _ in inst.reads:
          if rel[0]:
            raise NotFusable('seam-free strips: stage %s is read across lanes and reads '
                             '%s across lanes itself' % (inst.tensor, src.tensor))
          for side in ('lo', 'hi'):
            src.edges[side] = max(src.edges[side], inst.edges[side])
  # Rotation period: the row loop is unrolled `period` times so that every
  # window's "shift" is a renaming; each keep must divide it.  Keeping MORE rows
  # than needed is always legal, so keeps are rounded up to divisors of the
  # period that costs the fewest registers (jacobi2d: 6,3,3.. -> period 6, no
  # padding; denoise2d: 6,7,1,2,3.. would need lcm 42 -> period 8 with 8,8,1,2,4).
  best = None
  for candidate in range(1, max_period + 1):
    if max(inst.keep for inst in insts) > candidate:
      continue
    divisors = [d for d in range(1, candidate + 1) if candidate % d == 0]
    padded = [min(d for d in divisors if d >= inst.keep) if inst.keep else 0
              for inst in insts]
    cost = (sum(k * max(1, specmod.ELEM_SIZE[i.c_type] // 4)
                for k, i in zip(padded, insts)), candidate)
    if best is None or cost < best[0]:
      best = (cost, candidate, padded)
  if best is None:
    raise NotFusable('windows of up to %d rows exceed the rotation period limit %d'
                     % (max(inst.keep for inst in insts), max_period))
  period = best[1]
  for inst, keep in zip(insts, best[2]):
    inst.keep = keep
  # register budget: every retained row costs C VGPRs per lane (2C for 8-byte
  # types); past ~224 the kernel drops below two waves per SIMD and then spills
  est_vgprs = sum(inst.keep * (cols + sum(inst.edges.values())) *
                  max(1, specmod.ELEM_SIZE[inst.c_type] // 4) for inst in insts) + \
      4 * cols + 16
  if est_vgprs > vgpr_budget:
    raise NotFusable('depth %d would need about %d VGPRs (budget %d)'
                     % (depth, est_vgprs, vgpr_budget))
  # First step at which each instance's row can matter.  Rows of the final
  # output below the chunk's first row y0 are never stored, so walking back
  # through the readers gives, per instance, how many rows below y0 it is still
  # needed (`below`); it then first matters at step lag + y_lo - below.  During
  # the pipeline fill the generated prologue skips instances before that step.
  below = {id(final): 0}
  for inst in reversed(insts):
    need = below.get(id(inst))
    if need is None:
      continue
    for src, rel, _ in inst.reads:
      below[id(src)] = max(below.get(id(src), -10**9), need - rel[1])
  for inst in insts:
    inst.first_step = max(0, inst.lag + geo['y_lo'] - below.get(id(inst), 0))
  name = kernel_name(spec, depth)
  C = cols
  L = final.lag
  vec_bytes = C * elem
  T_in = builtin_type(in_type)
  T_out = builtin_type(types[out_name])

  o = []
  emit_line = o.append
  emit_line('// fused depth-%d kernel: %d stage instance(s), rotation period %d,'
            % (depth, len([i for i in insts if i.stage]), period))
  emit_line('// strip = %d columns (%d out + halo %d/%d), chunk = %d rows, '
            'prefetch %d rows'
            % (LANES * C, geo['w_out'], geo['halo_lo'], geo['halo_hi'],
               chunk_rows, prefetch))
  emit_line('//   instance            lag keep')
  for inst in insts:
    emit_line('//   %-18s %4d %4d%s' % (inst.ident, inst.lag, inst.keep,
                                        '  -> HBM' if inst.final else ''))
  vec_in = 'vec_%s_in' % name
  vec_out = 'vec_%s_out' % name
  emit_line('typedef %s %s __attribute__((ext_vector_type(%d), aligned(%d)));'
            % (T_in, vec_in, C, elem))
  emit_line('typedef %s %s __attribute__((ext_vector_type(%d), aligned(%d)));'
            % (T_out, vec_out, C, elem))
  emit_line('typedef unsigned soda_u2 __attribute__((ext_vector_type(2)));')
  emit_line('typedef unsigned soda_u4 __attribute__((ext_vector_type(4)));')
  # nontemporal: 1 = loads, 2 = stores, 4 = the stores of launches whose box does not
  # fit the Infinity Cache (kernel_common.NT_STREAMING_BYTES), as an instantiation of
  # the steady interior path chosen per launch.  Per launch at 16384^2, stores always /
  # never non-temporal: blur 222 vs 233 us, jacobi2d depth 1 452 vs 470, depth 2 464 vs
  # 476, sobel2d 263 vs 268, seidel2d depth 4 516 vs 511 (profiles/r03_k1bench.txt)
  nt_auto = bool(nontemporal & 4) and steady
  emit_line('template <bool INTERIOR%s%s>' % (', bool RAGGED' if steady else '',
                                              ', bool NT = false' if nt_auto else ''))
  emit_line('DEV void %s_strip(const soda_hip_args& a, const i64 xs, const i64 x,'
            ' const i64 y0, const i64 y1) {' % name)
  emit_line('  const i64 W = a.dims[0], H = a.dims[1];')
  emit_line('  const i64 st_lo = xs > a.box_lo[0] ? xs : a.box_lo[0];')
  emit_line('  const i64 st_hi = xs + %d < a.box_hi[0] ? xs + %d : a.box_hi[0];'
            % (geo['w_out'], geo['w_out']))
  for t in spec['inputs']:
    emit_line('  const %s* __restrict__ g_%s = (const %s*)a.tensor[%d];' % (
        builtin_type(t['c_type']), t['name'], builtin_type(t['c_type']),
        index[t['name']]))
  emit_line('  %s* __restrict__ g_out = (%s*)a.tensor[%d];' % (
      T_out, T_out, index[out_name]))
  if steady:
    emit_line('  const bool st_full = x >= st_lo && x + %d <= st_hi;' % C)
  if flat:
    emit_line('  const unsigned st_voff = st_full ? (unsigned)(x * %d) : 0xfffffff0u;' % elem)
  for inst in insts:
    # seam-free strips: a loaded row stays the VECTOR it arrived as until its cells are
    # used (sub-dword elements are unpacked there; unpacked at the load - behind the
    # scheduling fence of its step - every row would be waited for as soon as issued)
    inst.packed = flat and inst.stage is None and inst.c_type == in_type
    if inst.keep and inst.packed:
      emit_line('  %s %s[%d];' % (vec_in, inst.ident, inst.keep))
    elif inst.keep:
      emit_line('  %s %s[%d][%d];' % (builtin_type(inst.c_type), inst.ident,
                                     inst.keep, C))
    if any(inst.edges.values()):
      # seam-free strips: column k left of the strip (edge_lo: column xs - 1 - k) and
      # right of it (edge_hi: xs + 64 C + k) of every live row - what a lane below the
      # first / above the last one would hold - in every lane (lanes 0 and 63 use them).
      # Two registers rather than one shared by the halves of the wavefront: a register
      # read by two DPP moves is copied first, and the scheduler hoists that copy to the
      # top of the row loop, where it waits for the load issued last
      for side in ('lo', 'hi'):
        if inst.edges[side]:
          emit_line('  %s edge_%s_%s[%d][%d];' % (builtin_type(inst.c_type), side, inst.ident,
                                                 inst.keep, inst.edges[side]))
  # windows start as zeros so that the prologue computes on defined values
  for inst in insts:
    for r in range(inst.keep):
      emit_line('  ' + ' '.join('%s[%d][%d] = 0;' % (inst.ident, r, c)
                                for c in range(C)))
      for side in ('lo', 'hi'):
        if inst.edges[side]:
          emit_line('  ' + ' '.join('edge_%s_%s[%d][%d] = 0;' % (side, inst.ident, r, c)
                                    for c in range(inst.edges[side])))
  # (seam-free strips: EVERY lane loads the edge columns, one element per column and side
  # at a wave-uniform address: no branch in the row loop - under a branch the compiler
  # waits for all loads in flight at the join.  Clamped into the row where the array has
  # no such column: those values feed only cells outside every valid box.)
  emit_line('  // load head: first input row the chunk depends on')
  emit_line('  i64 head = y0 - %d;' % geo['y_lo'])
  emit_line('  const i64 steps = (y1 - y0) + %d;' % (L + geo['y_lo']))
  # the last input row any stored row depends on: the rows streamed after it only flush
  # the pipeline, and outside the branch-free loop they are not loaded (with the short
  # chunks of streaming launches - soda_hip_kernel.stream_chunk - they would be a
  # quarter of all row loads)
  emit_line('  const i64 load_last = y1 - 1 + %d;' % geo['y_hi'])
  prologue_steps = max(i.first_step for i in insts)
  prologue_steps = -(-prologue_steps // period) * period if skip_fill else 0
  emit_line('  i64 n = 0;')

  def slot(inst, u, back):
    """physical row of `inst`'s window holding its `back`-th newest row while
    unrolled copy `u` runs (after `inst` produced this step's row)."""
    return (u - back) % inst.keep

  def operand(reader, src, rel, u, c):
    back = reader.lag - src.lag - rel[1]
    assert 0 <= back < src.keep, (reader.ident, src.ident, rel, back, src.keep)
    row = '%s[%d]' % (src.ident, slot(src, u, back))
    j = c + rel[0]
    if 0 <= j < C:
      return '%s[%d]' % (row, j)
    if exact:       # the first / last lane take the column from the strip's edge loads
      assert src.edges['lo' if j < 0 else 'hi'] >= (-j if j < 0 else j - C + 1)
      if j < 0:
        return 'from_lane_below_or(%s[%d], edge_lo_%s[%d][%d])' % (
            row, C + j, src.ident, slot(src, u, back), -j - 1)
      return 'from_lane_above_or(%s[%d], edge_hi_%s[%d][%d])' % (
          row, j - C, src.ident, slot(src, u, back), j - C)
    if j < 0:
      return 'from_lane_below(%s[%d])' % (row, C + j)
    return 'from_lane_above(%s[%d])' % (row, j - C)

  def emit_body(guarded, calm=False):
    """calm: the steady-state copy - every row loaded lies inside the array, every
    row produced is stored."""
    for u in range(period):
      emit_line('    {  // unrolled step %d' % u)
      for inst in insts:
        if inst.stage is None:
          s = slot(inst, u, 0)
          # (only where the row loop proper is the branch-free copy: a branch around the
          # loads of the ONE loop of a deeper kernel makes the compiler wait for every
          # load in flight at its join - blur depth 4 298 -> 404 us per launch)
          emit_line('      %s{  // load row head+%d of %s' % (
              'if (head + %d <= load_last) ' % u if steady and not calm else '', u,
              inst.tensor))
          emit_line('        i64 row = head + %d;%s' % (
              u, '' if calm else ' if (row > H - 1) row = H - 1;'))
          emit_line('        const %s* p = g_%s + row * W + x;' % (
              builtin_type(inst.c_type), inst.tensor))
          emit_line('        if (INTERIOR) {')
          if nontemporal & 1:
            emit_line('          const %s v = __builtin_nontemporal_load((const %s*)p);'
                      % (vec_in, vec_in))
          else:
            emit_line('          const %s v = *(const %s*)p;' % (vec_in, vec_in))
          if inst.packed:
            emit_line('          %s[%d] = v;' % (inst.ident, s))
          for c in range(0 if inst.packed else C):
            emit_line('          %s[%d][%d] = v[%d];' % (inst.ident, s, c, c))
          emit_line('        } else {')
          for c in range(C):
            emit_line('          %s[%d][%d] = (x + %d >= 0 && x + %d < W) ? p[%d] : '
                      '(%s)0;' % (inst.ident, s, c, c, c, c,
                                  builtin_type(inst.c_type)))
          emit_line('        }')
          for side, k in [('lo', k) for k in range(inst.edges['lo'])] + \
              [('hi', k) for k in range(inst.edges['hi'])]:
            for ex in ('xs - %d' % (k + 1) if side == 'lo' else 'xs + %d' % (LANES * C + k),):
              emit_line('        { i64 ex = %s;' % ex)
              emit_line('          if (INTERIOR) { if (ex < 0) ex = 0; if (ex > W - 1) ex = W - 1; '
                        'edge_%s_%s[%d][%d] = g_%s[row * W + ex]; }' % (
                            side, inst.ident, s, k, inst.tensor))
              emit_line('          else edge_%s_%s[%d][%d] = (ex >= 0 && ex < W) ? '
                        'g_%s[row * W + ex] : (%s)0; }' % (side, inst.ident, s, k, inst.tensor,
                                                           builtin_type(inst.c_type)))
          emit_line('      }')
          continue
        stage = inst.stage
        ctype = builtin_type(inst.c_type)
        skip = guarded and inst.first_step > u
        if skip:
          emit_line('      if (n + %d >= %d) {' % (u, inst.first_step))
        by_name = {}
        for src, rel, load_name in inst.reads:
          by_name[(load_name, rel)] = src
        if inst.final:
          emit_line('      %s out_row[%d];' % (ctype, C))
        for c in range(C):
          def load(tensor, rel, u=u, c=c, inst=inst, by_name=by_name):
            return operand(inst, by_name[(tensor, tuple(rel))], tuple(rel), u, c)
          target = ('out_row[%d]' % c) if inst.final else \
              '%s[%d][%d]' % (inst.ident, slot(inst, u, 0), c)
          cell_assignment(stage, target, load, emit_line, '      ')
        for side in ('lo', 'hi'):
          # the same row beyond the strip, from the sources' edge columns
          for c in range(inst.edges[side]):
            def load_edge(tensor, rel, u=u, c=c, inst=inst, by_name=by_name, side=side):
              src = by_name[(tensor, tuple(rel))]
              back = inst.lag - src.lag - rel[1]
              return 'edge_%s_%s[%d][%d]' % (side, src.ident, slot(src, u, back), c)
            cell_assignment(stage, 'edge_%s_%s[%d][%d]' % (side, inst.ident,
                                                           slot(inst, u, 0), c),
                            load_edge, emit_line, '      ')
        if inst.final:
          emit_line('      {  // store row head+%d-%d' % (u, L))
          emit_line('        const i64 y = head + %d;' % (u - L))
          emit_line('        %s{' % ('' if calm else 'if (y >= y0 && y < y1) '))
          emit_line('          %s* q = g_out + y * W + x;' % T_out)
          if calm and flat:
            # !RAGGED (decided once per strip): a lane stores its whole vector or nothing -
            # ONE raw buffer store on the row, the lane's offset out of range where it
            # stores nothing: no branch, so the compiler's waits in this loop are exact
            # counts (behind a branch, and behind __builtin_nontemporal_store, it waits
            # for every access in flight: vmcnt(0) once per row or per trip)
            bits = {4: ('b32', 'unsigned'), 8: ('b64', 'soda_u2'),
                    16: ('b128', 'soda_u4')}[vec_bytes]
            emit_line('          if (!RAGGED) {')
            emit_line('            %s v;' % vec_out)
            for c in range(C):
              emit_line('            v[%d] = out_row[%d];' % (c, c))
            emit_line('            __builtin_amdgcn_raw_buffer_store_%s(__builtin_bit_cast(%s, v), '
                      '__builtin_amdgcn_make_buffer_rsrc((void*)(g_out + y * W), 0, '
                      '(int)(W * %d), 0x27000), st_voff, 0, %s);' % (
                          bits[0], bits[1], elem, 'NT ? 2 : 0' if nt_auto else
                          '2' if nontemporal & 2 else '0'))
            emit_line('          } else if (x >= st_lo && x + %d <= st_hi) {' % C)
          elif calm:
            emit_line('          if (RAGGED ? (x >= st_lo && x + %d <= st_hi) : st_full) {' % C)
          else:
            emit_line('          if (x >= st_lo && x + %d <= st_hi) {' % C)
          emit_line('            %s v;' % vec_out)
          for c in range(C):
            emit_line('            v[%d] = out_row[%d];' % (c, c))
          if nt_auto:
            emit_line('            if (NT) __builtin_nontemporal_store(v, (%s*)q); '
                      'else *(%s*)q = v;' % (vec_out, vec_out))
          elif nontemporal & 2:
            emit_line('            __builtin_nontemporal_store(v, (%s*)q);' % vec_out)
          else:
            emit_line('            *(%s*)q = v;' % vec_out)
          emit_line('          } else %s{' % ('if (RAGGED) ' if calm else ''))
          for c in range(C):
            emit_line('            if (x + %d >= st_lo && x + %d < st_hi) q[%d] = '
                      'out_row[%d];' % (c, c, c, c))
          emit_line('          }')
          emit_line('        }')
          emit_line('      }')
        if skip:
          emit_line('      }')
      emit_line('    }')
      if calm and flat:
        # a scheduling fence per row: nothing that uses a row moves up across the loads
        # issued after it (a use hoisted to the top of the loop waits for the load
        # issued LAST - vmcnt(0) once per trip - and the pipeline of rows in flight
        # drains there)
        emit_line('    __builtin_amdgcn_sched_barrier(0);')

  if prologue_steps:
    emit_line('  // pipeline fill: instances start as their windows become useful')
    emit_line('  for (; n < %d && n < steps; n += %d, head += %d) {'
              % (prologue_steps, period, period))
    emit_body(True)
    emit_line('  }')
  if steady and prologue_steps:
    # after the fill every row produced is stored (the prologue ends at a multiple of
    # the period past the first stored row); rows are loaded unclamped while
    # head + period - 1 <= H - 1
    emit_line('  {')
    emit_line('    const i64 in_array = H - y0 + %d;      // steps whose load row exists'
              % geo['y_lo'])
    emit_line('    const i64 calm_end = steps < in_array ? steps : in_array;')
    emit_line('    for (; n + %d <= calm_end; n += %d, head += %d) {' % (period, period, period))
    emit_body(False, calm=True)
    emit_line('    }')
    emit_line('  }')
  emit_line('  for (; n < steps; n += %d, head += %d) {' % (period, period))
  emit_body(False)
  emit_line('  }')
  emit_line('}')
  emit_line('')
  occupancy = ''
  if waves_per_eu > 0:
    occupancy = ' __attribute__((amdgpu_waves_per_eu(%d, %d)))' % (
        waves_per_eu, waves_per_eu)
  emit_line('GLOBAL WG_SIZE(%d)%s void %s(soda_hip_args a) {'
            % (WAVES_PER_BLOCK * LANES, occupancy, name))
  emit_line('  const int lane = lane_id();')
  emit_line('  const int wave = __builtin_amdgcn_workitem_id_x() >> 6;')
  emit_line('  const i64 x_origin = a.box_lo[0] - a.box_lo[0] %% %d;' % geo['origin_align'])
  # (Re-dealing the workgroups so that every XCD - ids b and b + 8 share one L2 - works
  # on a contiguous run of tiles was measured: jacobi2d 16384^2 depths 12 / 8 and blur
  # depth 1 within +-1 %, blur -3 %; the halo re-reads are a few percent of the traffic.)
  emit_line('  const unsigned block_x = __builtin_amdgcn_workgroup_id_x();')
  emit_line('  const unsigned block_y = __builtin_amdgcn_workgroup_id_y();')
  emit_line('  const i64 strip = (i64)block_x * %d + wave;' % WAVES_PER_BLOCK)
  emit_line('  const i64 xs = x_origin + strip * %d;' % geo['w_out'])
  emit_line('  if (xs >= a.box_hi[0]) return;')
  emit_line('  const i64 x = xs - %d + lane * %d;' % (geo['halo_lo'], C))
  emit_line('  const i64 chunk = a.param[0] > 0 ? a.param[0] : %d;' % chunk_rows)
  emit_line('  const i64 y0 = a.box_lo[1] + (i64)block_y * chunk;')
  emit_line('  const i64 y1 = y0 + chunk < a.box_hi[1] ? y0 + chunk : a.box_hi[1];')
  emit_line('  const bool interior = xs - %d >= 0 && xs - %d + %d <= a.dims[0];'
            % (geo['halo_lo'], geo['halo_lo'], LANES * C))
  if steady:
    # lanes that store only some of their columns: strips at the box's x edges when
    # the box does not start / end on a lane boundary
    emit_line('  const i64 st_lo = xs > a.box_lo[0] ? xs : a.box_lo[0];')
    emit_line('  const i64 st_hi = xs + %d < a.box_hi[0] ? xs + %d : a.box_hi[0];'
              % (geo['w_out'], geo['w_out']))
    if exact:      # seam-free strips: no halo lanes - `ragged` = not every lane stores
      emit_line('  const bool partial = !(x >= st_lo && x + %d <= st_hi);' % C)
    else:
      emit_line('  const bool partial = !(x >= st_lo && x + %d <= st_hi) && (%s);' % (
          C, ' || '.join('(x + %d >= st_lo && x + %d < st_hi)' % (c, c) for c in range(C))))
    # (rows of 2 GiB or more: the branch-free store's buffer resource takes a signed
    # 32-bit size and 32-bit lane offsets; such rows keep the guarded path)
    emit_line('  const bool ragged = __builtin_amdgcn_ballot_w64(partial) != 0 || '
              'a.dims[0] * %d >= 0x7ffffff0ll;' % elem)
    emit_line('  if (!interior) %s_strip<false, true>(a, xs, x, y0, y1);' % name)
    emit_line('  else if (ragged) %s_strip<true, true>(a, xs, x, y0, y1);' % name)
    if nt_auto:
      types = specmod.tensor_c_types(spec)
      cell_bytes = sum(specmod.ELEM_SIZE[types[n]] for n in
                       [t['name'] for t in spec['inputs']] + list(spec['outputs']))
      emit_line('  else if ((a.box_hi[0] - a.box_lo[0]) * (a.box_hi[1] - a.box_lo[1]) * %d > '
                '%dll) %s_strip<true, false, true>(a, xs, x, y0, y1);' % (
                    cell_bytes, kernel_common.NT_STREAMING_BYTES, name))
    emit_line('  else %s_strip<true, false>(a, xs, x, y0, y1);' % name)
  else:
    emit_line('  if (interior) %s_strip<true>(a, xs, x, y0, y1);' % name)
    emit_line('  else %s_strip<false>(a, xs, x, y0, y1);' % name)
  emit_line('}')
  entry = dict(name=name, kind='fused', depth=depth, stage=-1,
               block=[WAVES_PER_BLOCK * LANES, 1, 1],
               tile=[WAVES_PER_BLOCK * geo['w_out'], chunk_rows, 1, 1],
               origin_align=geo['origin_align'],
               fill_rows=L + geo['y_lo'],
               cols=C, prefetch=prefetch, period=period, est_vgprs=est_vgprs,
               halo=[geo['halo_lo'], geo['halo_hi']], w_out=geo['w_out'],
               steady=int(steady))
  if exact:
    # seam-free strips: straight-line row loop with counted waits - the kernel that
    # gains from fewer, longer chunks on arrays beyond the caches (16384^2, single
    # launches, workgroups per CU uncapped / 4 / 2 / 1: jacobi2d 436 / 436 / 414 / 466 us,
    # sobel2d 251 / 244 / 253 / 290; under the bench protocol, uncapped against 2:
    # jacobi2d 0.465 / 0.441 ms, sobel2d 0.258 / 0.257; the overlapping form LOSES with a
    # cap: sobel2d 0.272 / 0.314 ms; profiles/r04_k1_ab.txt)
    entry['exact'] = 1
  if stream_wgs is None:
    stream_wgs = 2 if exact else 0
  if stream_wgs:
    entry['stream_wgs_per_cu'] = int(stream_wgs)
  if nontemporal:
    entry['nontemporal'] = int(nontemporal)
  return '\n'.join(o) + '\n', entry
