"""Fused 3-D kernels, block form: ALL stage instances of `depth` iterations in
every wavefront, the plane tile of a workgroup cut into horizontal bands, one per
wavefront, and only the bands' EDGE ROWS exchanged through LDS.

Why (docs/DESIGN_HISTORY.md 4.1b): the wave-pipelined form (kernel_stream3d_wp: one level per
wavefront, whole plane tiles handed from wavefront to wavefront) is latency-bound -
its wavefronts wait for each other at the per-plane barrier 59 % of the time, a
level's whole tile makes an LDS round trip per plane, and 48 KiB + 168 VGPRs per
workgroup cap the CU at three workgroups.  Here a wavefront is a vertical stack of
levels for ITS rows (the single-wave form of kernel_stream3d, which fits two
levels, cut to R = 8 rows so that four fit), and what a band needs from its
neighbours is one or two rows per level and plane:

  * lane l of wavefront w holds C columns x R rows of every live plane of every
    level: columns wx + l*C.., rows wy + w*R..; x-neighbours by DPP wave shifts,
    y-neighbours inside the band are registers;
  * after producing a plane of level k a wavefront publishes its first and last
    rows in LDS (double-buffered by step parity); rows above / below the band are
    read from the neighbours' published rows ONE STEP LATER (one `s_barrier` per
    streamed plane) - for a 7-point stencil that costs no extra latency, because a
    level trails its parent by one plane anyway (it reads plane z+1);
  * the workgroup's tile is (64*C) x (G*R) cells: 128 x 64 with C = 2, R = 8 and
    G = 8 wavefronts keeps 120 x 56 = 82 % of itself at depth 4, where the 64 x 32
    tile of the wave-pipelined form keeps 66 %;
  * every wavefront runs the same code (no per-wavefront branches): a quarter of
    the instruction footprint, equal work per wavefront by construction;
  * tiles that would overhang the array are moved inside it and store only their
    own cells; loads and stores are raw buffer instructions on a per-plane resource
    (SGPR row offsets, one VGPR lane offset, out-of-range offsets for lanes that
    must not store); placement is XCD-aware (see kernel_stream3d_wp, xcd_tiles).

Reference correspondence: SODA's chain of compute modules (reference
src/soda/dataflow.py:122-346) replicated across `unroll factor` PEs; the PEs are
the wavefronts of a workgroup here, and what the FPGA's line buffers forward
between neighbouring PEs are the published edge rows.

Scope: one input, one output, 4-byte elements, x offsets within C columns,
y offsets within R rows.
"""
from . import kernel_common
from . import spec as specmod
from .kernel_common import builtin_type, cell_assignment, device_expr, tensor_index
from .kernel_stream2d import LANES, Instance, NotFusable
from .kernel_stream2d_wp import packable
from .kernel_stream3d import kernel_name


def build_chain(spec, depth, prefetch):
  """Stage instances with lags along z.  A read with dy != 0 may need a
  neighbouring wavefront's rows, which are visible one step after they were
  produced."""
  if len(spec['outputs']) != 1 or len(spec['inputs']) != 1:
    raise NotFusable('one input, one output')
  types = specmod.tensor_c_types(spec)
  in_name = spec['inputs'][0]['name']
  source = Instance('in_%s' % in_name, in_name, 0, types[in_name])
  insts = [source]
  current = {in_name: source}
  for it in range(depth):
    for stage in spec['stages']:
      inst = Instance('k%d_%s' % (it, stage['name']), stage['name'], it,
                      stage['c_type'], stage)
      inst.reads = [(current[t], tuple(rel), t) for t, rel in stage['loads']]
      insts.append(inst)
      current[stage['name']] = inst
    current[in_name] = current[spec['outputs'][0]]
  final = current[spec['outputs'][0]]
  final.final = True
  source.lag = 0
  source.ready = prefetch          # steps after its load at which a plane exists
  for inst in insts[1:]:
    inst.ready = 0
    inst.lag = max(src.lag + src.ready + rel[2] + (1 if rel[1] else 0)
                   for src, rel, _ in inst.reads)
  for inst in insts:
    inst.up = inst.down = 0        # rows the bands above / below need from this one
    inst.age = 0                   # steps a published row must stay readable
  for inst in insts[1:]:
    for src, rel, _ in inst.reads:
      src.keep = max(src.keep, inst.lag - rel[2] - src.lag + 1)
      if rel[1] < 0:               # the band below reads our LAST rows
        src.down = max(src.down, -rel[1])
      if rel[1] > 0:               # the band above reads our FIRST rows
        src.up = max(src.up, rel[1])
      if rel[1]:
        src.age = max(src.age, inst.lag - rel[2] - src.lag - src.ready)
  for inst in insts[1:]:
    if not inst.final and inst.keep == 0:
      raise NotFusable('stage %s is never read' % inst.tensor)
  return insts, final


MAX_PERIOD = 12     # longest unrolled rotation of the register windows
ALIGN_OUT = 16      # cells: output tiles start and end on 64-byte pieces


def emit(spec, depth, cols=2, rows=8, stack=8, chunk_planes=64, prefetch=0,
         vgpr_budget=250, ring=0, pairs=0, stamps=0, mask_loads=0, nt=0,
         wide_stores=0, lean_fill=0, edge=0):
  """Returns (text, kernel table entry).

  `prefetch` = input planes loaded ahead into REGISTERS (R*C VGPRs each);
  `ring` = N > 0: instead, every wavefront keeps N planes of its band in flight as
  LDS-direct loads (`global_load_lds_dwordx4`, no registers) into its own N-slot
  LDS ring: at step t it waits for exactly plane t (explicit vmcnt: loads and
  stores retire in issue order on gfx9, so the wait counts both - which is why a
  ring kernel issues the SAME number of stores every step, dropped by an
  out-of-range offset where nothing is to be stored), reads it into the input
  window's registers and refills the slot with plane t + N.
  `pairs` = 1 (float programs of + - * /, kernel_stream2d_wp.packable): band rows p
  and p + R/2 of a lane share a 64-bit register pair, so a level's plane is R/2
  pair-rows of v_pk_add_f32 / v_pk_mul_f32.  The y-neighbour of a pair-row is the
  next pair-row except at the seams (the low halves' row above the band and the
  high halves' row below it come from the neighbouring bands' edge rows, rows
  R/2-1 -> R/2 are a low half meeting its own high half): those operands, and the
  lane-crossing x-neighbours, are two scalars each (kernel_common: pk2_shifted,
  the DPP shift folded into a v_add_f32_dpp).  Same IEEE operations in the same
  order as the scalar form.
  Fixed parts of the design (they were options while they were being measured; the A/B
  figures are in docs/DESIGN_HISTORY.md 4.1d and profiles/r03_blk_variants.txt, r03_blk_stamps.txt):
  * the tiles' output columns start on multiples of ALIGN_OUT = 16 cells and are a
    multiple of 16 wide, so that the row segments of neighbouring workgroups meet on
    64-byte boundaries (tools/tile3dbench.hip, this kernel's access pattern without the
    arithmetic, 512^3: row segments that start 16 bytes into a line cost a pure copy
    +41 % - a line shared by two workgroups leaves L2 as masked partial writes - 64
    bytes into it +14 %; 120 of 128 columns kept 294 us, 112 kept 247 us);
  * each XCD (workgroup id mod 8) takes a RUN of consecutive tiles (x fastest, then y,
    then z chunks), so that tiles which share halo cells and cache lines run on one L2
    (the copy 247 -> 236 us);
  * no branch inside a step: the band function is instantiated for ragged and for full
    tiles (decided once per workgroup), a lane's part of the store predicate is a
    loop-invariant offset, a row's or plane's part the record count of the row's buffer
    resource (0 drops the store) - the same eight store instructions every step, which
    is what the ring's counted wait needs (loads and stores share one in-order counter;
    behind stores under branches the compiler waits for all of them).  Skipping the
    levels and loads a chunk's fill steps do not need was measured with branches (+14 %:
    they cut the step's one basic block) and without (nothing); so was ending the
    wavefronts of bands outside every stored cell's cone (nothing: those launches are
    bound by memory);
  * a scheduling fence between the parts of a step (input plane, each stage instance,
    barrier): left alone, the scheduler's order of the ~550 instructions was 25 % slower;
  * a level's reads of its neighbours' edge rows (LDS) are issued one part of the step
    ahead of its arithmetic (register-prefetch form only; the ring form reads them where
    they are used).
  A hand-ordered instruction stream for the arithmetic (one `asm` per VALU instruction,
  cells interleaved) was built and measured twice: it paid only for heavy programs and
  packed pair-rows beat it there, so it is gone.
  `lean_fill` = 1 (ring form): a chunk of n planes walks n + fill steps, and in the first
  and last of them some levels compute planes no stored cell depends on (depth 4: 4 n +
  32 level-planes computed, 4 n + 12 needed).  The row loop is emitted twice: trips in
  which every level is needed at every step run the branch-free body; the trips at a
  chunk's two ends run a copy whose levels sit under wave-uniform conditions (the parts
  of a step are fenced apart anyway, so a scalar branch per part costs little there).
  Memory operations are NOT skipped - the ring's counted waits need the same loads and
  stores every step (stores of skipped planes are dropped by the record count as always)
  - and a skipped level's edge rows are read only by cells that are skipped or unneeded
  themselves.  Round 3 measured level skipping with the branches in EVERY step: +14 %.
  `stamps` = device address of a debug buffer (tools/blk_stamps.py only): the
  wavefront sums the shader cycles (s_memtime) it spends in each part of a step -
  input plane, each stage instance, the barrier - and lane 0 writes the sums there;
  scheduling fences keep the parts apart, so a stamped build is a diagnostic, not
  the shipped kernel.
  `nt` (bit mask): 2 = the output stores are non-temporal, 1 = the input loads, 4 = the
  stores of launches whose box (input + output) is larger than the Infinity Cache
  (kernel_common.NT_STREAMING_BYTES): a second instantiation of the band function,
  chosen by the kernel's entry from the box it is given.  Stores: -6..-10 % per
  launch beyond the cache, +5..+15 % inside it; loads: up to +35 % on the largest
  boxes.  4 ships.
  `mask_loads` = 1: a ragged tile fetches only the box's reach - columns right of it
  have an out-of-range lane offset (loop-invariant), rows below it lie past the record
  count of the plane's buffer resource (the rows ride in the scalar offset, which
  gfx950 includes in the range check: tools/soffset_check.hip); in the ring the
  LDS-direct loads take their buffer form (raw_ptr_buffer_load_lds).  What is not
  fetched reads as zero and feeds only cells outside every stored cell's cone.  Ships.
  `wide_stores` = 1: row segments are stored in whole 64-byte pieces - the cells between
  the box and the next 64-byte boundary (outside the valid box of the level:
  unspecified by contract, read by no later launch, stored by no other tile) get the
  tile's values for them; 2 = only in the launches beyond the Infinity Cache (with the
  non-temporal instantiation).  A partial piece at a box's edge costs a masked write:
  cfg5 -2.5..-6 % under the bench protocol.  2 ships (not for packed pair-rows: heat3d
  +3 %).
  `edge` = 1: rounding the tile's width down to whole 64-byte pieces leaves `slack` valid
  columns a tile computes and does not store (128 - 8 = 120 at depth 4, 112 stored).  The
  FIRST and the LAST tile of a row of tiles store them (soda_hip_kernel.edge_slack): the
  launcher starts the tiles up to `slack` columns inside the box (param[1] >> 32), the first
  tile's window moves `slack` columns to the left, the last tile stores up to the box's
  end - nx tiles cover 112 nx + 8 columns wherever the box starts instead of 112 nx minus
  box_lo % 16.  cfg5: the boxes 464, 456, 336, 328, 240, 232 and 112 lose a tile column (45 -> 36,
  24 -> 18, 15 -> 10, 4 -> 2 tiles per plane).  Interior tiles are untouched; the moved
  window's edge pieces may be partial (one piece per row and side).  Ships."""
  if spec['dim'] != 3:
    raise NotFusable('3-D programs only')
  types = specmod.tensor_c_types(spec)
  index = tensor_index(spec)
  in_type = spec['inputs'][0]['c_type']
  out_name = spec['outputs'][0]
  elem = specmod.ELEM_SIZE[in_type]
  if any(specmod.ELEM_SIZE[t] != elem for t in types.values()) or elem != 4:
    raise NotFusable('the block form handles 4-byte elements')
  C, R, G = cols, rows, stack
  pairs = int(bool(pairs))
  if pairs and (R % 2 or not packable(spec)):
    raise NotFusable('packed pair-rows: an even number of rows and a plain float program')
  RP = R // 2 if pairs else R      # register rows per lane (pair-rows when packed)
  suffix = {4: 'b32', 8: 'b64', 16: 'b128'}.get(C * elem)
  if suffix is None:
    raise NotFusable('4-, 8- or 16-byte lanes')
  buf_type = {4: 'unsigned', 8: 'soda_u2', 16: 'soda_u4'}[C * elem]
  if ring and prefetch:
    raise NotFusable('ring and register prefetch exclude each other')
  insts, final = build_chain(spec, depth, prefetch)
  source = insts[0]
  # planes of an instance that the chunk's output planes [z0, z1) depend on:
  # [z0 - need_lo, z1 + need_hi)
  for inst in insts:
    inst.need_lo = inst.need_hi = None
  final.need_lo = final.need_hi = 0
  for inst in reversed(insts):
    if inst.need_lo is None:
      continue
    for src, rel, _ in inst.reads:
      lo_need, hi_need = inst.need_lo - rel[2], inst.need_hi + rel[2]
      src.need_lo = lo_need if src.need_lo is None else max(src.need_lo, lo_need)
      src.need_hi = hi_need if src.need_hi is None else max(src.need_hi, hi_need)
  lean_fill = bool(lean_fill) and bool(ring) and not stamps
  rows_per_load = 16 // (C * elem)     # a 16-byte-per-lane load covers this many rows
  if ring and (rows_per_load < 1 or R % rows_per_load):
    raise NotFusable('ring: %d rows per load do not divide %d rows' % (rows_per_load, R))
  ring_loads = R // max(1, rows_per_load)
  margins = specmod.iteration_margins(spec, depth)
  lo, hi = margins[-1]
  halo_lo = -(-lo[0] // C) * C
  halo_hi = -(-hi[0] // C) * C
  w_out = LANES * C - halo_lo - halo_hi
  align_out = max(C, ALIGN_OUT)
  if align_out % C:
    raise NotFusable('%d columns a lane do not divide the 64-byte store pieces' % C)
  if w_out >= align_out:
    w_out -= w_out % align_out
  else:
    align_out = C
  slack = LANES * C - halo_lo - halo_hi - w_out if edge and align_out > C else 0
  edge = int(slack > 0)
  TR = G * R
  y_lo, y_hi = lo[1], hi[1]
  r_out = TR - y_lo - y_hi
  if w_out < C or r_out < 1:
    raise NotFusable('depth %d leaves no output cells in a %dx%d tile'
                     % (depth, LANES * C, TR))
  for inst in insts:
    for src, rel, _ in inst.reads:
      if abs(rel[0]) > C:
        raise NotFusable('x offset %d exceeds the %d columns a lane holds'
                         % (rel[0], C))
      if abs(rel[1]) > RP:
        raise NotFusable('y offset %d exceeds the %d rows a band holds' % (rel[1], RP))
  slots = max([i.age for i in insts]) + 1
  slots = max(2, slots)
  best = None
  for candidate in range(1, MAX_PERIOD + 1):
    if max(inst.keep for inst in insts) > candidate or candidate % slots or \
        (ring and candidate % ring):
      continue
    divisors = [d for d in range(1, candidate + 1) if candidate % d == 0]
    padded = [min(d for d in divisors if d >= inst.keep) if inst.keep else 0
              for inst in insts]
    cost = (sum(padded), candidate)
    if best is None or cost < best[0]:
      best = (cost, candidate, padded)
  if best is None:
    raise NotFusable('windows exceed the rotation period limit')
  period = best[1]
  for inst, keep in zip(insts, best[2]):
    inst.keep = keep
  est_vgprs = sum(inst.keep * R * C for inst in insts) + R * C + 32
  if est_vgprs > vgpr_budget:
    raise NotFusable('a wavefront would need about %d VGPRs (budget %d)'
                     % (est_vgprs, vgpr_budget))
  publishers = [i for i in insts if i.up or i.down]
  edge_rows = max([i.up + i.down for i in publishers] or [1])
  lds_bytes = len(publishers) * slots * (G + 2) * edge_rows * LANES * C * elem
  ring_bytes = ring * G * R * LANES * C * elem
  if lds_bytes + ring_bytes > 160 * 1024:
    raise NotFusable('edge rows and ring need %d bytes of LDS' % (lds_bytes + ring_bytes))
  name = kernel_name(spec, depth) + 'b'    # next to the wave-pipelined kernel
  L = final.lag
  T = builtin_type(in_type)
  vec = 'vec_%s' % name
  o = []
  line = o.append
  line('// fused depth-%d 3-D kernel, block form: %d wavefronts x (%d x %d) bands ='
       % (depth, G, LANES * C, R))
  line('// tile %d x %d (%d x %d out), rotation period %d, prefetch %d, ~%d VGPRs, '
       '%d KiB LDS (edge rows, %d slots)' % (LANES * C, TR, w_out, r_out, period,
                                             prefetch, est_vgprs, lds_bytes // 1024,
                                             slots))
  for inst in insts:
    line('//   %-18s lag %2d keep %2d  publishes %d first / %d last rows%s' % (
        inst.ident, inst.lag, inst.keep, inst.up, inst.down,
        '  -> HBM' if inst.final else ''))
  line('typedef %s %s __attribute__((ext_vector_type(%d), aligned(%d)));'
       % (T, vec, C, C * elem))
  line('typedef unsigned soda_u2 __attribute__((ext_vector_type(2)));')
  # (by value: __builtin_bit_cast applied directly to the high element of a
  # <2 x float> lvalue stored the LOW element in the ragged-edge path)
  line('DEV unsigned %s_bits(%s v) { return __builtin_bit_cast(unsigned, v); }' % (name, T))
  # edge rows of a stage of another 4-byte type travel through the LDS array (typed as
  # the input) as the same BITS (by value, for the reason above)
  line('template <typename D, typename S> DEV D %s_pun(S v) { return __builtin_bit_cast(D, v); }'
       % name)
  line('typedef unsigned soda_u4 __attribute__((ext_vector_type(4)));')

  def slot(inst, u, back):
    return (u - back) % inst.keep

  def cell(ident, s, r, c):
    """Band row r, column c of window slot s as a scalar lvalue."""
    if pairs:
      return '%s[%d][%d][%d][%d]' % (ident, s, r % RP, c, r // RP)
    return '%s[%d][%d][%d]' % (ident, s, r, c)

  pub_index = {id(inst): k for k, inst in enumerate(publishers)}

  def edge(src, step, wave_expr, row):
    """LDS row `row` (0 .. up-1: the band's first rows, up .. : its last rows) of
    what `src` published at unrolled step `step`."""
    return 'edges[%d][%d][%s][%d]' % (pub_index[id(src)], step % slots, wave_expr, row)

  def publish(inst, u, s):
    """First `up` and last `down` rows of the plane in window slot s."""
    def bits(text):
      if pairs or inst.c_type == in_type:
        return text
      return '%s_pun<%s>(%s)' % (name, T, text)
    for k in range(inst.up):
      line('        { %s v;%s *(%s*)&%s[lane * %d] = v; }' % (
          vec, ''.join(' v[%d] = %s;' % (c, bits(cell(inst.ident, s, k, c)))
                       for c in range(C)), vec, edge(inst, u, 'wave + 1', k), C))
    for k in range(inst.down):
      line('        { %s v;%s *(%s*)&%s[lane * %d] = v; }' % (
          vec, ''.join(' v[%d] = %s;' % (
              c, bits(cell(inst.ident, s, R - inst.down + k, c))) for c in range(C)), vec,
          edge(inst, u, 'wave + 1', inst.up + k), C))

  def vmcnt(n):   # s_waitcnt vmcnt(N) only (expcnt and lgkmcnt left at their maxima)
    return (n & 15) | (7 << 4) | (15 << 8) | ((n >> 4) << 14)

  if ring:
    # a band's plane from the wavefront's ring slot in one burst of reads the
    # compiler does not see as LDS reads (in front of a visible read of memory
    # that LDS-direct loads write it would wait for ALL of them;
    # kernel_common: soda_lds_read_f4)
    width = {4: 'b32', 8: 'b64', 16: 'b128'}[C * elem]
    line('DEV void soda_ring_read_%s(const void* base, %s) {' % (
        name, ', '.join('%s& v%d_%d' % (T, r, c) for r in range(R) for c in range(C))))
    line('  %s;' % '; '.join('%s t%d' % (vec, r) for r in range(R)))
    line('  asm volatile(%s' % ''.join(
        '"ds_read_%s %%%d, %%%d offset:%d\\n\\t"\n               ' % (
            width, r, R, r * LANES * C * elem) for r in range(R)))
    line('               "s_waitcnt lgkmcnt(0)"')
    line('               : %s' % ', '.join('"=&v"(t%d)' % r for r in range(R)))
    line('               : "v"((unsigned)(unsigned long long)base) : "memory");')
    for r in range(R):
      line('  ' + ' '.join('v%d_%d = t%d[%d];' % (r, c, r, c) for c in range(C)))
    line('}')
  nt_auto = bool(nt & 4)
  stream_expr = ('(a.box_hi[0] - a.box_lo[0]) * (a.box_hi[1] - a.box_lo[1]) * '
                 '(a.box_hi[2] - a.box_lo[2]) * %d > %dll' % (
                     2 * elem, kernel_common.NT_STREAMING_BYTES))
  if wide_stores and align_out * elem % 64 == 0:
    # Row segments are stored in whole 64-byte pieces: the columns between the box and
    # the next 64-byte boundary on either side get the tile's values for them too.  They
    # lie outside the valid box of this level - unspecified cells by contract
    # (include/soda_hip.h: soda_hip_sweep), which no later launch reads: its box plus
    # reach is this box - and no other tile stores them.  A box that starts 16 bytes
    # into a line otherwise costs its edge tiles masked partial writes (4.1d).
    # wide_stores = 2: only for launches beyond the Infinity Cache (the non-temporal
    # instantiation; inside the caches a partial line costs nothing and the extra cells
    # are only more bytes: boxes of 232^3 .. 336^3 +1 %)
    seg = 64 // elem
    when = 'ST_WIDE' if wide_stores == 2 else None
    store_range = [
        '  i64 st_box_lo = a.box_lo[0], st_box_hi = a.box_hi[0];',
        '  %s{ st_box_lo -= st_box_lo %% %d; st_box_hi += %d - 1; st_box_hi -= st_box_hi %% %d; '
        'if (st_box_hi > a.dims[0]) st_box_hi = a.dims[0]; }' % (
            'if (%s) ' % when if when else '', seg, seg, seg),
        '  const i64 st_lo = xs - lo_ext > st_box_lo ? xs - lo_ext : st_box_lo;',
        '  const i64 st_hi = xs + %d + hi_ext < st_box_hi ? xs + %d + hi_ext : st_box_hi;' % (
            w_out, w_out)]
  else:
    wide_stores = 0
    store_range = [
        '  const i64 st_lo = xs - lo_ext > a.box_lo[0] ? xs - lo_ext : a.box_lo[0];',
        '  const i64 st_hi = xs + %d + hi_ext < a.box_hi[0] ? xs + %d + hi_ext : a.box_hi[0];' % (
            w_out, w_out)]
  line('template <bool RAGGED%s>' % (', bool NT' if nt_auto else ''))
  line('DEV void %s_band(const soda_hip_args& a, const i64 xs, const i64 lo_ext, '
       'const i64 hi_ext, const i64 yb, '
       'const i64 wx, const i64 wy, const i64 z0, const i64 z1, const int wave, '
       'const int lane, %s (*edges)[%d][%d][%d][%d], %s (*in_ring)[%d][%d][%d]) {'
       % (name, T, slots, G + 2, edge_rows, LANES * C, T, G if ring else 1,
          R if ring else 1, LANES * C if ring else 1))
  line('  const i64 W = a.dims[0], H = a.dims[1], D = a.dims[2];')
  line('  const i64 plane = W * H;')
  line('  const i64 plane_bytes = plane * %d;' % elem)
  line('  const i64 x = wx + lane * %d;' % C)
  line('  const i64 y_band = wy + wave * %d;     // first row of this band' % R)
  line('  const unsigned lane_byte = (unsigned)((y_band * W + x) * %d);' % elem)
  if wide_stores == 2:
    line('  const bool ST_WIDE = %s;' % ('NT' if nt_auto else stream_expr))
  for text in store_range:
    line(text)
  line('  const i64 st_ylo = a.box_lo[1] > yb + %d ? a.box_lo[1] : yb + %d;'
       % (y_lo, y_lo))
  line('  const i64 st_yhi = a.box_hi[1] < yb + %d ? a.box_hi[1] : yb + %d;'
       % (TR - y_hi, TR - y_hi))
  line('  const bool st_full = x >= st_lo && x + %d <= st_hi;' % C)
  for c in range(C):
    line('  const bool st_col%d = x + %d >= st_lo && x + %d < st_hi;' % (c, c, c))
  # no branch inside a step: raggedness is a template parameter, the lane's part of
  # the store predicate a loop-invariant offset, the row's and the plane's part the
  # record count of the row's buffer resource (scalar selects)
  line('  const bool st_ragged = RAGGED;')
  line('  const unsigned st_voff = st_full ? lane_byte : 0xfffffff0u;')
  for c in range(C):
    line('  const unsigned st_voff%d = st_col%d ? lane_byte + %d : 0xfffffff0u;'
         % (c, c, c * elem))
  line('  unsigned st_rows = 0;')
  for r in range(R):
    line('  if (y_band + %d >= st_ylo && y_band + %d < st_yhi) st_rows |= %du;'
         % (r, r, 1 << r))
  if mask_loads and ring:
    line('  const i64 ld_rows = a.box_hi[1] + %d < H ? a.box_hi[1] + %d : H;' % (hi[1], hi[1]))
    line('  const i64 ld_plane_bytes = ld_rows * W * %d;' % elem)
  elif mask_loads:
    # a ragged tile reads only what some stored cell depends on: lanes right of the
    # box's reach "load" out of range (a loop-invariant offset), rows below it lie
    # past the record count of the plane's resource - neither costs an instruction
    # in the loop, and what they would have fetched is never moved
    line('  const unsigned ld_lane_byte = x < a.box_hi[0] + %d ? lane_byte : 0xfffffff0u;'
         % hi[0])
    line('  const i64 ld_rows = a.box_hi[1] + %d < H ? a.box_hi[1] + %d : H;' % (hi[1], hi[1]))
    line('  const i64 ld_plane_bytes = ld_rows * W * %d;' % elem)
  else:
    line('  const unsigned ld_lane_byte = lane_byte;')
    line('  const i64 ld_plane_bytes = plane_bytes;')
  line('  const %s* __restrict__ g_in = (const %s*)a.tensor[%d];'
       % (T, T, index[spec['inputs'][0]['name']]))
  line('  %s* __restrict__ g_out = (%s*)a.tensor[%d];' % (T, T, index[out_name]))
  for inst in insts:
    if inst.keep:
      line('  %s %s[%d][%d][%d];' % ('pk2' if pairs else builtin_type(inst.c_type),
                                     inst.ident, inst.keep, RP, C))
      for k in range(inst.keep):
        for r in range(RP):
          line('  ' + ' '.join('%s[%d][%d][%d] = %s;' % (
              inst.ident, k, r, c, 'pk2{0.0f, 0.0f}' if pairs else '0')
                               for c in range(C)))
  line('  i64 head = z0 - %d;' % lo[2])
  line('  const i64 steps = (z1 - z0) + %d;' % (L + lo[2]))
  line('  const i64 span = z1 - z0;')

  # `nt`: the non-temporal bit (aux bit 1) on the output stores (2) / the input loads
  # (1); 4 = on the stores of launches whose box does not fit the Infinity Cache (an
  # instantiation of its own, chosen per launch by the kernel's entry)
  st_aux = 'NT ? 2 : 0' if nt_auto else '2' if nt & 2 else '0'
  ld_aux = 2 if nt & 1 else 0

  def ring_load(slot_index, plane_expr, indent):
    line(indent + '{ i64 zz = %s; if (zz > D - 1) zz = D - 1;' % plane_expr)
    if mask_loads:
      # the buffer form of the LDS-direct load: chunks right of the box's reach have an
      # out-of-range offset, rows below it lie past the record count - neither is
      # fetched (zeros arrive in the ring; only cells outside every stored cell's
      # cone read them)
      line(indent + '  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_'
           'rsrc((void*)(g_in + zz * plane), 0, (int)ld_plane_bytes, 0x27000);')
      for i in range(ring_loads):
        line(indent + '  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__(('
             'address_space(3))) void*)&in_ring[%d][wave][%d][0], 16, ld_dma_byte, '
             '(unsigned)(%d * W * %d), 0, %d);' % (
                 slot_index, i * rows_per_load, i * rows_per_load, elem, ld_aux))
      line(indent + '}')
      return
    line(indent + '  const %s* p = g_in + zz * plane + dma_lane;' % T)
    for i in range(ring_loads):
      line(indent + '  __builtin_amdgcn_global_load_lds((const __attribute__(('
           'address_space(1))) void*)(p + %d * W), (__attribute__((address_space(3)))'
           ' void*)&in_ring[%d][wave][%d][0], 16, 0, %d);' % (
               i * rows_per_load, slot_index, i * rows_per_load, ld_aux))
    line(indent + '}')

  if ring:
    # lane l of an LDS-direct load moves 16 bytes: row l / per_row of the group of
    # rows_per_load rows, 16-byte chunk l % per_row of it
    per_row = LANES * C * elem // 16
    line('  const i64 dma_lane = (y_band + lane / %d) * W + wx + (lane %% %d) * %d;'
         % (per_row, per_row, 16 // elem))
    if mask_loads:
      line('  const unsigned ld_dma_byte = wx + (lane %% %d) * %d < a.box_hi[0] + %d ? '
           '(unsigned)(dma_lane * %d) : 0xfffffff0u;' % (per_row, 16 // elem, hi[0], elem))
    for k in range(ring):
      ring_load(k, 'head + %d' % k, '  ')
    # the first planes are waited for outright; from step `ring` on the counted
    # wait below is exact
    line('  __builtin_amdgcn_s_waitcnt(%d);  // vmcnt(0)' % vmcnt(0))
  if stamps:
    line('  unsigned long long soda_t[%d];' % (len(insts) + 1))
    for k in range(len(insts) + 1):
      line('  soda_t[%d] = 0;' % k)
    line('  unsigned long long soda_prev = __builtin_readcyclecounter();')

  def stamp(k):
    if not stamps:
      # scheduling fence between the parts of a step: the input plane, each stage
      # instance, the barrier (what the stamped diagnostic build has as well)
      line('      __builtin_amdgcn_sched_barrier(0);')
    if stamps:
      line('      __builtin_amdgcn_sched_barrier(0);')
      line('      { const unsigned long long now = __builtin_readcyclecounter(); '
           'soda_t[%d] += now - soda_prev; soda_prev = now; }' % k)
      line('      __builtin_amdgcn_sched_barrier(0);')
  def out_cell(r, c):
    if pairs:
      return 'out_tile[%d][%d][%d]' % (r % RP, c, r // RP)
    return 'out_tile[%d][%d]' % (r, c)

  def operand_pk(reader, src, rel, u, p, c):
    """Packed form: the operand of pair-row p (band rows p and p + RP)."""
    back = reader.lag - src.lag - rel[2]
    assert 0 <= back < src.keep, (reader.ident, src.ident, rel, back, src.keep)
    s = slot(src, u, back)
    j = c + rel[0]
    jj = j if 0 <= j < C else (C + j if j < 0 else j - C)
    shift = None if 0 <= j < C else ('below' if j < 0 else 'above')
    pp = p + rel[1]
    if 0 <= pp < RP:
      whole = '%s[%d][%d][%d]' % (src.ident, s, pp, jj)
      return whole if shift is None else 'pk_from_lane_%s(%s)' % (shift, whole)
    if pp < 0:     # low half: the band above's last rows; high half: low halves
      lo = 'xa_%s_%d_%d[%d]' % (src.ident, s, src.down + pp, jj)
      hi = '%s[%d][%d][%d][0]' % (src.ident, s, RP + pp, jj)
    else:          # low half: the high halves; high half: the band below's first rows
      lo = '%s[%d][%d][%d][1]' % (src.ident, s, pp - RP, jj)
      hi = 'xb_%s_%d_%d[%d]' % (src.ident, s, pp - RP, jj)
    if shift is not None:
      lo, hi = ('from_lane_%s(%s)' % (shift, v) for v in (lo, hi))
    return 'pk2_shifted{%s, %s}' % (lo, hi)

  def operand(reader, src, rel, u, r, c):
    if pairs:
      return operand_pk(reader, src, rel, u, r, c)
    back = reader.lag - src.lag - rel[2]
    assert 0 <= back < src.keep, (reader.ident, src.ident, rel, back, src.keep)
    s = slot(src, u, back)
    rr = r + rel[1]
    if rr < 0:        # the band above: its last rows
      row = 'xa_%s_%d_%d' % (src.ident, s, src.down + rr)
    elif rr >= R:     # the band below: its first rows
      row = 'xb_%s_%d_%d' % (src.ident, s, rr - R)
    else:
      row = '%s[%d][%d]' % (src.ident, s, rr)
    j = c + rel[0]
    if 0 <= j < C:
      return '%s[%d]' % (row, j)
    if j < 0:
      return 'from_lane_below(%s[%d])' % (row, C + j)
    return 'from_lane_above(%s[%d])' % (row, j - C)

  def edge_read_lines(inst, u, visible):
    """The rows of the neighbouring bands a stage instance reads at step u: loads
    from the published edge rows (LDS) into registers.  `visible` = the (instance, slot)
    keys already declared in the enclosing step's scope (two stages that read one
    tensor's rows ahead of their arithmetic would declare them twice); returns the
    lines and the keys they declare."""
    out = []
    wanted = {}
    for src, rel, _ in inst.reads:
      if not rel[1]:
        continue
      back = inst.lag - src.lag - rel[2]
      key = (src.ident, slot(src, u, back))
      age = inst.lag - rel[2] - src.lag - src.ready
      wanted[key] = (src, age)
    wanted = {key: v for key, v in wanted.items() if key not in visible}
    for (ident, s), (src, age) in sorted(wanted.items()):
      def elem_of(row_vec, c, src=src):
        if pairs or src.c_type == in_type:
          return '%s[%d]' % (row_vec, c)
        return '%s_pun<%s>(%s[%d])' % (name, builtin_type(src.c_type), row_vec, c)
      for k in range(src.down):     # last rows of the band above
        out.append('        const %s xa_v_%s_%d_%d = *(const %s*)&%s[lane * %d];' % (
            vec, ident, s, k, vec, edge(src, u - age, 'wave', src.up + k), C))
        out.append('        const %s xa_%s_%d_%d[%d] = {%s};' % (
            builtin_type(src.c_type), ident, s, k, C, ', '.join(
                elem_of('xa_v_%s_%d_%d' % (ident, s, k), c) for c in range(C))))
      for k in range(src.up):       # first rows of the band below
        out.append('        const %s xb_v_%s_%d_%d = *(const %s*)&%s[lane * %d];' % (
            vec, ident, s, k, vec, edge(src, u - age, 'wave + 2', k), C))
        out.append('        const %s xb_%s_%d_%d[%d] = {%s};' % (
            builtin_type(src.c_type), ident, s, k, C, ', '.join(
                elem_of('xb_v_%s_%d_%d' % (ident, s, k), c) for c in range(C))))
    return out, set(wanted)

  def needed(inst, u):
    """Wave-uniform condition under which `inst` has to run at unrolled step u of the
    trip that starts at step n: its plane head + u - lag lies in [z0 - need_lo,
    z1 + need_hi)."""
    first = lo[2] + inst.lag - inst.need_lo
    last = lo[2] + inst.lag + inst.need_hi          # n + u < span + last
    parts = ['n + %d >= %d' % (u, first)] if first > 0 else []
    parts.append('n + %d < span + %d' % (u, last))
    return ' && '.join(parts), first, last

  def emit_trip(guarded):
    for u in range(period):
      line('    {  // unrolled step %d' % u)
      edges_done = set()
      in_step_scope = set()      # edge rows declared at the step's own level
      if not ring:      # the first level's rows, ahead of the input plane's part
        ahead = [i for i in insts if i.stage is not None]
        for inst in ahead[:1]:
          edges_done.add(id(inst))
          lines, keys = edge_read_lines(inst, u, in_step_scope)
          in_step_scope |= keys
          for text in lines:
            line(text)
      for inst_index, inst in enumerate(insts):
        if inst_index:
          stamp(inst_index - 1)
        if inst.stage is None and ring:
          s = slot(inst, u, 0)
          # plane head+u was issued `ring` steps ago; since then this wavefront has
          # issued ring-1 planes of loads and `ring` steps of (at least) R stores
          wait = (ring - 1) * ring_loads + ring * R
          # (the ragged instantiation stores column by column)
          ragged_wait = (ring - 1) * ring_loads + ring * R * C
          line('      __builtin_amdgcn_s_waitcnt(RAGGED ? %d : %d);  // vmcnt(%d / %d)' % (
              vmcnt(min(63, ragged_wait)), vmcnt(wait), min(63, ragged_wait), wait))
          line('      { %s t[%d][%d];' % (T, R, C))
          line('        soda_ring_read_%s(&in_ring[%d][wave][0][lane * %d], %s);' % (
              name, u % ring, C, ', '.join('t[%d][%d]' % (r, c)
                                           for r in range(R) for c in range(C))))
          for r in range(R):
            line('        ' + ' '.join('%s = t[%d][%d];' % (cell(inst.ident, s, r, c), r, c)
                                       for c in range(C)))
          line('      }')
          ring_load(u % ring, 'head + %d' % (u + ring), '      ')
          if inst.up or inst.down:
            publish(inst, u, s)
          continue
        if inst.stage is None:
          s = slot(inst, u, 0)
          line('      { i64 zz = head + %d; if (zz > D - 1) zz = D - 1;' % u)
          line('        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_'
               'rsrc((void*)(g_in + zz * plane), 0, (int)ld_plane_bytes, 0x27000);')
          for r in range(R):
            line('        { const %s v = __builtin_bit_cast(%s, __builtin_amdgcn_raw_'
                 'buffer_load_%s(rs, %s, (unsigned)(%d * W * %d), %d));%s }' % (
                     vec, vec, suffix, 'ld_lane_byte', r, elem, ld_aux,
                     ''.join(
                         ' %s = v[%d];' % (cell(inst.ident, s, r, c), c)
                         for c in range(C))))
          line('      }')
          if inst.up or inst.down:    # the plane that "exists" from this step on
            publish(inst, u, slot(inst, u, inst.ready))
          continue
        stage = inst.stage
        ctype = builtin_type(inst.c_type)
        by_name = {(n, rel): src for src, rel, n in inst.reads}
        if not ring:      # the next level's rows, ahead of this one's
          later = [i for i in insts[insts.index(inst) + 1:] if i.stage is not None]
          if later:                           # (step scope: the next block uses them)
            edges_done.add(id(later[0]))
            lines, keys = edge_read_lines(later[0], u, in_step_scope)
            in_step_scope |= keys
            for text in lines:
              line(text)
        line('      {')
        if pairs:
          ctype = 'pk2'
        if inst.final:
          line('        %s out_tile[%d][%d];' % (ctype, RP, C))
        if guarded:
          # (a skipped final level still issues its stores - the counted waits need them
          # - with a record count of 0: its plane lies outside [z0, z1))
          if inst.final:
            for r in range(RP):
              line('        ' + ' '.join('out_tile[%d][%d] = %s;' % (
                  r, c, 'pk2{0.0f, 0.0f}' if pairs else '0') for c in range(C)))
          line('        if (%s) {' % needed(inst, u)[0])
        if id(inst) not in edges_done:
          for text in edge_read_lines(inst, u, in_step_scope)[0]:
            line(text)
        for r in range(RP):
          for c in range(C):
            def load(tensor, rel, u=u, r=r, c=c, inst=inst, by_name=by_name):
              return operand(inst, by_name[(tensor, tuple(rel))], tuple(rel), u, r, c)
            target = ('out_tile[%d][%d]' % (r, c)) if inst.final else \
                '%s[%d][%d][%d]' % (inst.ident, slot(inst, u, 0), r, c)
            cell_assignment(stage, target, load, line, '        ')
        if inst.up or inst.down:
          publish(inst, u, slot(inst, u, 0))
        if guarded:
          line('        }')
        if inst.final:
          line('        const i64 z = head + %d;' % (u - L))
          line('        const bool z_ok = z >= z0 && z < z1;')
          line('        const unsigned rows_now = z_ok ? st_rows : 0u;')
          line('        %s* const out_plane = g_out + (z_ok ? z : z0) * plane;' % T)
          for r in range(R):
            line('        { const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_'
                 'rsrc((void*)out_plane, 0, (rows_now >> %d) & 1u ? (int)plane_bytes : 0, '
                 '0x27000);' % r)
            line('          if (!RAGGED) { %s v;%s __builtin_amdgcn_raw_buffer_store_%s('
                 '__builtin_bit_cast(%s, v), rs, st_voff, (unsigned)(%d * W * %d), %s); }' % (
                     vec, ''.join(' v[%d] = %s;' % (c, out_cell(r, c))
                                  for c in range(C)), suffix, buf_type, r, elem, st_aux))
            line('          else {%s }' % ''.join(
                ' __builtin_amdgcn_raw_buffer_store_b32(%s_bits(%s), rs, st_voff%d, '
                '(unsigned)(%d * W * %d), %s);' % (name, out_cell(r, c), c, r, elem, st_aux)
                for c in range(C)))
            line('        }')
        line('      }')
      stamp(len(insts) - 1)
      line('    }')
      line('    %s();' % ('soda_lds_barrier' if ring else 'soda_block_barrier'))
      stamp(len(insts))

  stages = [i for i in insts if i.stage is not None]
  line('  i64 n = 0;')
  if lean_fill:
    firsts = [needed(i, 0)[1] for i in stages]
    lasts = [needed(i, 0)[2] for i in stages]
    lean_from = -(-max(firsts + [0]) // period) * period
    line('  // trips at the chunk\'s start: levels run from the step their plane is needed at')
    line('  for (; n < %d && n < steps; n += %d, head += %d) {' % (lean_from, period, period))
    emit_trip(True)
    line('  }')
    line('  // every level needed at every step of the trip: the branch-free body')
    line('  for (; n + %d <= span + %d; n += %d, head += %d) {' % (period, min(lasts), period,
                                                                 period))
    emit_trip(False)
    line('  }')
    line('  // trips at the chunk\'s end')
  line('  for (; n < steps; n += %d, head += %d) {' % (period, period))
  emit_trip(lean_fill)
  line('  }')
  if stamps:
    line('  if (lane == 0) {')
    line('    unsigned long long* dbg = (unsigned long long*)%dull + '
         '((i64)__builtin_amdgcn_workgroup_id_x() * %d + wave) * 16;' % (int(stamps), G))
    for k in range(len(insts) + 1):
      line('    dbg[%d] = soda_t[%d];' % (k, k))
    line('    dbg[15] = (unsigned long long)steps;')
    line('  }')
  line('}')
  line('')
  line('GLOBAL WG_SIZE(%d) void %s(soda_hip_args a) {' % (G * LANES, name))
  line('  __attribute__((shared)) %s edges[%d][%d][%d][%d][%d];' % (
      T, max(1, len(publishers)), slots, G + 2, edge_rows, LANES * C))
  if ring:
    line('  __attribute__((shared)) %s in_ring[%d][%d][%d][%d];' % (T, ring, G, R,
                                                                   LANES * C))
  line('  const int lane = lane_id();')
  line('  const int wave = __builtin_amdgcn_readfirstlane('
       '__builtin_amdgcn_workitem_id_x() >> 6);')
  # (a 1-D grid: the kernel places its tiles itself, XCD by XCD)
  line('  const unsigned L = __builtin_amdgcn_workgroup_id_x();')
  line('  const unsigned SX = (unsigned)a.param[1] & 0xffffu, '
       'SY = (unsigned)a.param[1] >> 16;')
  line('  const unsigned nsx = (unsigned)a.param[2] & 0xffffu, '
       'nsy = (unsigned)a.param[2] >> 16;')
  # param[3] = P > 0: runs - XCD x = L mod 8 takes tiles [x P, (x + 1) P)
  line('  const unsigned P = (unsigned)a.param[3];')
  line('  const unsigned S = SX * SY, i = L >> 3, g = P ? (L & 7u) * P + i : '
       '(i / S) * 8u + (L & 7u), within = P ? 0u : i % S;')
  line('  const unsigned block_x = (g % nsx) * SX + within % SX;')
  line('  const unsigned block_y = ((g / nsx) % nsy) * SY + within / SX;')
  line('  const unsigned block_z = g / (nsx * nsy);')
  if edge:
    # where the tiles start comes from the launcher (up to `slack` columns inside the box:
    # soda_hip_kernel.edge_slack); the first tile of a row reaches back to the box's start
    # with its window moved `slack` columns to the left, the last one stores up to its end
    line('  const i64 x_origin = (i64)((unsigned long long)a.param[1] >> 32);')
    line('  const i64 xs = x_origin + (i64)block_x * %d;' % w_out)
    line('  if (xs >= a.box_hi[0]) return;')
    line('  const bool shifted = block_x == 0 && a.box_lo[0] < xs;')
    line('  const i64 lo_ext = shifted ? %d : 0;' % slack)
    line('  const i64 hi_ext = block_x + 1 == nsx && !shifted ? %d : 0;' % slack)
  else:
    line('  const i64 x_origin = a.box_lo[0] - a.box_lo[0] %% %d;' % align_out)
    line('  const i64 xs = x_origin + (i64)block_x * %d;' % w_out)
    line('  if (xs >= a.box_hi[0]) return;')
    line('  const i64 lo_ext = 0, hi_ext = 0;')
  line('  const i64 yb = a.box_lo[1] + (i64)block_y * %d - %d;' % (r_out, y_lo))
  line('  const i64 chunk = a.param[0] > 0 ? a.param[0] : %d;' % chunk_planes)
  line('  const i64 z0 = a.box_lo[2] + (i64)block_z * chunk;')
  line('  if (yb + %d >= a.box_hi[1] || z0 >= a.box_hi[2]) return;' % y_lo)
  line('  const i64 z1 = z0 + chunk < a.box_hi[2] ? z0 + chunk : a.box_hi[2];')
  # a tile that would overhang the array is moved inside it: every load is
  # unguarded and the tile still stores only its own cells
  line('  i64 wx = xs - %d - lo_ext;' % halo_lo)
  line('  if (wx + %d > a.dims[0]) wx = a.dims[0] - %d;' % (LANES * C, LANES * C))
  line('  if (wx < 0) wx = 0;')
  line('  i64 wy = yb;')
  line('  if (wy + %d > a.dims[1]) wy = a.dims[1] - %d;' % (TR, TR))
  line('  if (wy < 0) wy = 0;')
  # does any lane of this tile store only some of its columns?  (tiles at the box's
  # x edges when the box does not start or end on a lane boundary)
  line('  const i64 x = wx + lane * %d;' % C)
  if nt_auto or wide_stores == 2:
    line('  const bool streaming = %s;' % stream_expr)
    line('  const bool ST_WIDE = streaming; (void)ST_WIDE;')
  for text in store_range:
    line(text)
  line('  const bool partial = !(x >= st_lo && x + %d <= st_hi) && (%s);' % (
      C, ' || '.join('(x + %d >= st_lo && x + %d < st_hi)' % (c, c) for c in range(C))))
  call = '%s_band<%%s>(a, xs, lo_ext, hi_ext, yb, wx, wy, z0, z1, wave, lane, edges, %s);' % (
      name, 'in_ring' if ring else 'nullptr')
  if nt_auto:
    # a box (input + output) beyond the Infinity Cache: its stores bypass the caches
    # (kernel_common.NT_STREAMING_BYTES; jacobi3d per launch inside a 512^3 array,
    # always / never: box 496 222 vs 246 us, 448 161 vs 176, 400 141 vs 153, 344 106
    # vs 111, 320 85 vs 81, 224 46 vs 40, 112 35 vs 35)
    line('  const bool ragged = __builtin_amdgcn_ballot_w64(partial) != 0;')
    line('  if (streaming) { if (ragged) %s else %s }' % (call % 'true, true',
                                                         call % 'false, true'))
    line('  else { if (ragged) %s else %s }' % (call % 'true, false',
                                                call % 'false, false'))
  else:
    line('  if (__builtin_amdgcn_ballot_w64(partial) != 0)')
    line('    ' + call % 'true')
    line('  else')
    line('    ' + call % 'false')
  line('}')
  entry = dict(name=name, kind='fused', depth=depth, stage=-1,
               block=[G * LANES, 1, 1], tile=[w_out, r_out, chunk_planes, 1],
               origin_align=align_out, fill_rows=L + lo[2], cols=C, rows=R, stack=G,
               prefetch=prefetch, period=period, est_vgprs=est_vgprs, w_out=w_out,
               r_out=r_out, lds_bytes=lds_bytes + ring_bytes, ring=ring, pairs=pairs,
               xcd_tiles=-1, min_extent=[LANES * C, TR])
  if nt:                    # (only when set: the shipped kernels' calibration keys stay)
    entry['nt'] = int(nt)
  if mask_loads:
    entry['mask_loads'] = 1
  if lean_fill:
    entry['lean_fill'] = 1
  if wide_stores:
    entry['wide_stores'] = int(wide_stores)
  if edge:
    entry['edge_slack'] = int(slack)
  return '\n'.join(o) + '\n', entry
