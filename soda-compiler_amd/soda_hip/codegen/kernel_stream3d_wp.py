"""Fused 3-D kernels, wave-pipelined form: the plane-streaming construction of
kernel_stream3d with the chain of stage instances cut into G groups, one
wavefront per group, plane tiles handed from group to group through LDS.

Why: a level of a 7-point stencil keeps three plane tiles in registers (3*R*C
per lane, 96 VGPRs at R = 16, C = 2), so one wavefront holds two levels at most
and a launch advances two iterations.  The 3-D kernels are bounded by memory
(jacobi3d 512^3: 156 us per depth-2 launch against ~60 us of VALU work), so what
they need is DEPTH.  With one level per wavefront a workgroup of four advances
four iterations per launch at the same registers per wavefront; the price is an
LDS round trip per plane tile (R*C values per lane, two slots per hand-off) and
one barrier per streamed plane.

Reference correspondence: this is SODA's chain of compute modules connected by
FIFOs (reference src/soda/dataflow.py:122-346) with wavefronts for modules and
LDS plane tiles for FIFOs.

Scope: single-input single-output chains in which a group reads only itself
and the last instance of the previous group; anything else keeps the
single-wave form.
"""
from . import kernel_common
from . import spec as specmod
from .kernel_common import builtin_type, cell_assignment, device_expr, tensor_index
from .kernel_stream2d import LANES, NotFusable
from .kernel_stream2d_wp import build_groups, packable
from .kernel_stream3d import kernel_name


def emit(spec, depth, cols=2, rows=16, chunk_planes=64, prefetch=0, groups=4,
         max_period=12, vgpr_budget=200, lds_budget=64 * 1024, split=2,
         waves_per_eu=3, loader=0, ring_prefetch=2, sched_fence=1, pairs=0,
         xcd_remap=0, prio='3', xcd_tiles=1, buffer_io=1, nt=0):
  """Returns (text, kernel table entry).

  split=2: the wavefront is a 32 x 2 grid of lanes; lane (lx, ly) holds columns
  lx*C.. of rows ly*R.. so the tile is (32*C) x (2*R) instead of (64*C) x R.  The
  composed halo of `depth` iterations is the same on every side, so the squarer
  tile keeps more of itself: depth 4, C = 2, R = 16: 56x24 of 64x32 (66 %)
  against 120x8 of 128x16 (47 %) - less redundant arithmetic and fewer halo
  cells fetched.  Rows that cross the halves come from one v_permlane32_swap per
  register (tools/permlane_test.hip).

  sched_fence=1, waves_per_eu=3: left alone, the scheduler hoists the NEXT
  planes' loads above the current plane's arithmetic in the loading wavefront
  (its own prefetch, 32 fresh registers per plane): jacobi3d 252 VGPRs, heat3d
  274 (one workgroup per CU, 814 us per launch).  A scheduling fence after each
  plane's loads keeps them in place (176 / 229 VGPRs) and the occupancy hint asks
  for three workgroups per CU (168 VGPRs; 9 / 135 spilled registers):
  jacobi3d 415 -> 356 us per full-size launch, heat3d 601 -> 509 us.

  pairs=1 (split=2, float programs of + - * /, kernel_stream2d_wp.packable): tile
  rows r and r + R/2 of a lane share a 64-bit register pair, so the arithmetic is
  v_pk_add_f32 / v_pk_mul_f32 on R/2 pair-rows.  y-neighbours of a pair-row are
  the next pair-row except at the two seams (row R/2 - 1 -> R/2 is the low half
  meeting its own high half, rows -1 and R come from the other 32-lane half as
  before): those operands are two scalars (kernel_common: pk2_shifted) and so are
  the lane-crossing x-neighbours, each shift folded into its scalar add.  Loads
  and stores address rows, so the first wavefront spreads each loaded row over
  the halves of R/2 pairs and the last one gathers them again (register moves
  only there); hand-offs keep the pair layout (16 bytes per lane and pair-row).

  buffer_io=1 (default, two row blocks, 4- or 8- or 16-byte lanes): the plane
  tiles are loaded and stored with raw buffer instructions on a per-plane
  resource (base = the plane, in SGPRs; row offset = an SGPR; lane offset = one
  VGPR), so no 64-bit address is computed per row, and the stores need no
  branches: a lane that must not store a row gets an offset beyond the resource
  and the hardware drops the access.  Before, the storing wavefront spent 346
  VALU + 270 SALU instructions and ~100 branches per plane (row and column
  predicates as exec-mask regions) against 230 VALU in the wavefronts between -
  and every wavefront of the workgroup waits for the slowest at the barrier.

  xcd_tiles=1 (default): XCD-aware placement.  Workgroups are dealt round-robin
  over the 8 XCDs (ids b and b + 8 share one XCD and its L2), so under the plain
  3-D grid the tiles that read each other's halo sit on DIFFERENT XCDs and every
  halo cell - and every 128-byte line a 64-float row shares with the next tile -
  is fetched from HBM twice: the PMC read bytes of jacobi3d were 2.5x the written
  ones (tile halo 1.52 x line sharing 1.34 x z-chunk fill).  Here the grid is 1-D
  and the launcher cuts the plane of tiles into super-tiles of SX x SY tiles
  (param[1]); super-tile g goes to XCD g mod 8 and its tiles are CONSECUTIVE
  workgroups of that XCD, hence resident together and streaming the same planes
  at the same time: what one fetches, its neighbours find in L2.  Placement only:
  any placement gives the same results (MI355X_MICROARCH.md: dispatch order is not
  a contract).  Every XCD gets the same number of super-tiles; tiles beyond the
  grid's edge exit at once.

  xcd_remap=1 (off): workgroups re-dealt so that each XCD (own L2) works on a
  contiguous run of tiles.  Measured on MI355X: jacobi3d 512^3 -1.5 % per launch,
  heat3d +6 %, cfg5 (box shrinking to 112^3) +12 % - the runs unbalance the XCDs
  when a launch has few workgroups.

  loader=1 (with split=2, EXPERIMENTAL, off): an extra wavefront does nothing but
  stream input plane tiles into an LDS ring with LDS-direct loads
  (global_load_lds_dwordx4: no registers, `ring_prefetch` planes in flight per
  workgroup); the first compute group reads the ring like any other hand-off.
  Bit-exact, but measured slower on MI355X (jacobi3d 512^3, depth 4: 690-790 us
  per launch against 417 us with the first group loading through its own
  registers, at the same two workgroups per CU), so it is not the default."""
  if spec['dim'] != 3:
    raise NotFusable('3-D programs only')
  types = specmod.tensor_c_types(spec)
  index = tensor_index(spec)
  in_type = spec['inputs'][0]['c_type']
  out_name = spec['outputs'][0]
  elem = specmod.ELEM_SIZE[in_type]
  if any(specmod.ELEM_SIZE[t] != elem for t in types.values()):
    raise NotFusable('mixed element widths')
  if elem != 4:
    raise NotFusable('the wave-pipelined 3-D form handles 4-byte elements')
  C, R = cols, rows
  BUF_SUFFIX = {4: 'b32', 8: 'b64', 16: 'b128'}.get(C * elem)
  BUF_TYPE = {4: 'unsigned', 8: 'soda_u2', 16: 'soda_u4'}.get(C * elem)
  if BUF_SUFFIX is None:
    buffer_io = 0
  if split not in (1, 2):
    raise NotFusable('split: 1 or 2 row blocks per wavefront')
  LX = LANES // split          # lanes along x
  TR = split * R               # tile rows
  loader = int(loader) if (loader and split == 2) else 0
  self_load = loader == 2        # the first compute wavefront fills the ring itself
  extra = 1 if loader == 1 else 0          # a wavefront that only loads
  pairs = int(bool(pairs))
  if pairs and (split != 2 or R % 2 or loader or C * 2 * elem != 16):
    raise NotFusable('packed 3-D form: two row blocks, even R, 2 columns, no loader')
  if pairs and not packable(spec):
    raise NotFusable('packed form: float programs of + - * / only')
  RP = R // 2 if pairs else R      # register rows per lane (pair-rows when packed)

  def cell(ident, s, r, c):
    """Register holding tile row r, column c of window slot s."""
    if pairs:
      return '%s[%d][%d][%d][%d]' % (ident, s, r % RP, c, r // RP)
    return '%s[%d][%d][%d]' % (ident, s, r, c)
  PF = ring_prefetch
  # ring slots: PF in flight, one being read; one more free when another wavefront
  # fills the ring (it runs a barrier interval ahead)
  RS = PF + (1 if self_load else 2)
  everything, per_wave, final = build_groups(spec, depth, prefetch, groups,
                                             loader=bool(loader),
                                             ring_lag=0 if self_load else 1)
  margins = specmod.iteration_margins(spec, depth)
  lo, hi = margins[-1]
  halo_lo = -(-lo[0] // C) * C
  halo_hi = -(-hi[0] // C) * C
  w_out = LX * C - halo_lo - halo_hi
  y_lo, y_hi = lo[1], hi[1]
  r_out = TR - y_lo - y_hi
  if w_out < C or r_out < 1:
    raise NotFusable('depth %d leaves no output cells in a %dx%d tile'
                     % (depth, LX * C, TR))
  if split == 2:
    for inst in everything:
      for src, rel, _ in inst.reads:
        if abs(rel[1]) > R:
          raise NotFusable('y offset %d exceeds the %d rows a lane holds'
                           % (rel[1], R))
  for inst in everything:
    for src, rel, _ in inst.reads:
      if abs(rel[0]) > C:
        raise NotFusable('x offset %d exceeds the %d columns a lane holds'
                         % (rel[0], C))
  # one even rotation period for the whole workgroup (two LDS slots, the same
  # number of barriers per loop trip in every wavefront)
  best = None
  for candidate in range(2, max_period + 1, 2):
    if max(i.keep for i in everything) > candidate:
      continue
    if loader and candidate % RS:
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
  est_vgprs = max(sum(i.keep for i in mine) for mine in per_wave) * R * C + \
      R * C + 28
  if est_vgprs > vgpr_budget:
    raise NotFusable('a wavefront would need about %d VGPRs (budget %d)'
                     % (est_vgprs, vgpr_budget))
  tile_elems = R * LANES * C
  lds_bytes = max(1, groups - 1) * 2 * tile_elems * elem
  if loader:
    if (TR * LX * C * elem) % (LANES * 16) or (LX * C * elem) % 16:
      raise NotFusable('tile rows do not split into 16-byte LDS-direct loads')
    lds_bytes += RS * tile_elems * elem
    lds_budget = max(lds_budget, 80 * 1024)
  if lds_bytes > lds_budget:
    raise NotFusable('hand-off tiles need %d bytes of LDS (budget %d)'
                     % (lds_bytes, lds_budget))
  stage_boxes = specmod.iteration_boxes(spec, depth)
  name = kernel_name(spec, depth)
  L = final.lag
  T_in = builtin_type(in_type)
  T_out = builtin_type(types[out_name])
  vec = 'vec_%s' % name
  o = []
  line = o.append
  line('// fused depth-%d 3-D kernel, wave-pipelined: %d wavefronts per tile column,'
       % (depth, groups))
  line('// tile %d x %d (%d x %d out), rotation period %d, prefetch %d planes, '
       '~%d VGPRs, %d KiB LDS' % (LX * C, TR, w_out, r_out, period, prefetch,
                                  est_vgprs, lds_bytes // 1024))
  for g, mine in enumerate(per_wave):
    for inst in mine:
      line('//   wave %d  %-20s lag %2d keep %2d  %s' % (
          g, inst.ident, inst.lag, inst.keep,
          '-> HBM' if inst.final else inst.role))
  line('typedef %s %s __attribute__((ext_vector_type(%d), aligned(%d)));'
       % (T_in, vec, C, elem))
  line('typedef %s %s_lds __attribute__((ext_vector_type(%d), aligned(%d)));'
       % (T_in, vec, C, C * elem))
  if buffer_io:
    line('typedef unsigned soda_u2 __attribute__((ext_vector_type(2)));')
    line('typedef unsigned soda_u4 __attribute__((ext_vector_type(4)));')

  def slot(inst, u, back):
    return (u - back) % inst.keep

  def operand_pk(reader, src, rel, u, p, c):
    """Packed form: the operand of pair-row p (tile rows p and p + RP)."""
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
    if pp < 0:     # low half: rows above the lane's block; high half: low halves
      lo = 'xa_%s_%d_%d[%d]' % (src.ident, s, R + pp, jj)
      hi = '%s[%d][%d][%d][0]' % (src.ident, s, RP + pp, jj)
    else:          # low half: the high halves; high half: rows below the block
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
    rr = r + rel[1]
    if split == 1:
      rr = min(max(rr, 0), R - 1)      # clamped rows are halo rows
    if rr < 0:        # last rows of the half above
      row = 'xa_%s_%d_%d' % (src.ident, slot(src, u, back), R + rr)
    elif rr >= R:     # first rows of the half below
      row = 'xb_%s_%d_%d' % (src.ident, slot(src, u, back), rr - R)
    else:
      row = '%s[%d][%d]' % (src.ident, slot(src, u, back), rr)
    j = c + rel[0]
    if 0 <= j < C:
      return '%s[%d]' % (row, j)
    if j < 0:
      return 'from_lane_below(%s[%d])' % (row, C + j)
    return 'from_lane_above(%s[%d])' % (row, j - C)

  def row_off(r):
    # offset of the lane's tile row r inside a plane, without the column.  One
    # row block: wave-uniform, rows clamped into the array (clamped rows only
    # feed halo cells).  Two row blocks: a per-lane base plus a uniform r*W, no
    # clamping - the guarded path tests the row instead.
    return 'row_y[%d]' % r if split == 1 else '(lane_row + %d * W)' % r

  def row_inside(r):
    return '' if split == 1 else ' && yt + %d >= 0 && yt + %d < H' % (r, r)

  if loader:
    # LDS-direct loads: lane l moves 16 bytes, an instruction 1 KiB = a few tile
    # rows; nothing comes back to registers
    rows_per_load = LANES * 16 // (LX * C * elem)
    loads = TR // rows_per_load
    per_row = LX * C * elem // 16

  def ring_load(slot_expr, plane_expr, indent='      '):
    line(indent + '{ i64 zz = %s; if (zz > D - 1) zz = D - 1;' % plane_expr)
    line(indent + '  const %s* p = g_in + zz * plane + dma_lane;' % T_in)
    for i in range(loads):
      line(indent + '  __builtin_amdgcn_global_load_lds((const __attribute__(('
           'address_space(1))) void*)(p + %d * W), (__attribute__((address_space(3)))'
           ' void*)&in_ring[%s][%d][0], 16, 0, 0);' % (i * rows_per_load, slot_expr,
                                                      i * rows_per_load))
    line(indent + '}')

  def vmcnt(n):   # s_waitcnt vmcnt(N) only (expcnt and lgkmcnt left at their maxima)
    return (n & 15) | (7 << 4) | (15 << 8) | ((n >> 4) << 14)

  barrier = 'soda_lds_barrier' if self_load else 'soda_block_barrier'

  def emit_body(mine):
    for u in range(period):
      line('      {  // unrolled step %d' % u)
      for inst in mine:
        if inst.role == 'global_in':
          s = slot(inst, u, 0)
          line('        { i64 zz = head + %d; if (zz > D - 1) zz = D - 1;' % u)
          line('          const %s* p = g_in + zz * plane;' % T_in)
          line('          if (INTERIOR) {')
          if buffer_io and split == 2:
            line('            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_'
                 'buffer_rsrc((void*)p, 0, (int)plane_bytes, 0x27000);')
          for r in range(R):
            if buffer_io and split == 2:
              line('            { const %s v = __builtin_bit_cast(%s, __builtin_amdgcn_raw_'
                   'buffer_load_%s(rs, lane_byte, (unsigned)(%d * W * %d), 0));%s }' % (
                       vec, vec, BUF_SUFFIX, r, elem, ''.join(' %s = v[%d];' % (
                           cell(inst.ident, s, r, c), c) for c in range(C))))
              continue
            address = '(p + %s + x)' % row_off(r) if split == 1 else \
                '((const char*)(p + %d * W) + lane_byte)' % r
            line('            { const %s v = *(const %s*)%s;%s }' % (
                vec, vec, address, ''.join(' %s = v[%d];' % (
                    cell(inst.ident, s, r, c), c) for c in range(C))))
          line('          } else {')
          for r in range(R):
            for c in range(C):
              line('            %s = (x + %d >= 0 && x + %d < W%s) ? '
                   'p[%s + x + %d] : (%s)0;' % (cell(inst.ident, s, r, c), c, c,
                                                row_inside(r), row_off(r), c, T_in))
          line('          } }')
          if sched_fence:
            # keep this plane's loads where they are: the scheduler otherwise
            # hoists the NEXT planes' loads above them into fresh registers
            line('          __builtin_amdgcn_sched_barrier(0);')
          continue
        if inst.role == 'ring_in' and self_load:
          s = slot(inst, u, 0)
          # this wavefront keeps PF planes in flight: issue plane head+u+PF, wait
          # for exactly plane head+u (issued PF steps ago), read its rows back
          ring_load('%d' % ((u + PF) % RS), 'head + %d' % (u + PF), '        ')
          line('        __builtin_amdgcn_s_waitcnt(%d);  // vmcnt(%d)'
               % (vmcnt(PF * loads), PF * loads))
          line('        soda_ring_read_%s(&in_ring[%d][ly * %d][(lane & %d) * %d], %s);'
               % (name, u % RS, R, LX - 1, C, ', '.join(
                   '%s[%d][%d][%d]' % (inst.ident, s, r, c)
                   for r in range(R) for c in range(C))))
          continue
        if inst.role == 'ring_in':
          s = slot(inst, u, 0)
          # the plane the loader made visible at the previous barrier
          for r in range(R):
            line('        { const %s_lds v = *(const %s_lds*)&in_ring[%d][ly * %d + %d]'
                 '[(lane & %d) * %d];%s }' % (
                     vec, vec, (u + RS - 1) % RS, R, r, LX - 1, C, ''.join(
                         ' %s[%d][%d][%d] = v[%d];' % (inst.ident, s, r, c, c)
                         for c in range(C))))
          continue
        if inst.role == 'lds_in' and pairs:
          s = slot(inst, u, 0)
          for q in range(RP):
            line('        { const soda_f4 v = *(const soda_f4*)&handoff[%d][%d][%d]'
                 '[lane * 4];%s }' % (inst.handoff, (u + 1) % 2, q, ''.join(
                     ' %s[%d][%d][%d] = pk2{v[%d], v[%d]};' % (
                         inst.ident, s, q, c, 2 * c, 2 * c + 1) for c in range(C))))
          continue
        if inst.role == 'lds_in':
          s = slot(inst, u, 0)
          for r in range(R):
            line('        { const %s_lds v = *(const %s_lds*)&handoff[%d][%d][%d]'
                 '[lane * %d];%s }' % (
                     vec, vec, inst.handoff, (u + 1) % 2, r, C, ''.join(
                         ' %s[%d][%d][%d] = v[%d];' % (inst.ident, s, r, c, c)
                         for c in range(C))))
          continue
        stage = inst.stage
        ctype = 'pk2' if pairs else builtin_type(inst.c_type)
        by_name = {(n, rel): src for src, rel, n in inst.reads}
        direct = inst.keep == 0
        if direct:
          line('        %s out_tile[%d][%d];' % (ctype, RP, C))
          if inst.role == 'lds_out':   # rows outside the dependency cone: zeros
            for r in range(RP):
              line('        ' + ' '.join('out_tile[%d][%d] = %s;' % (
                  r, c, 'pk2{0.0f, 0.0f}' if pairs else '0') for c in range(C)))
        # rows whose whole dependency cone lies inside the tile (with two row
        # blocks every lane computes all its rows: the halves run in lock step)
        blo, bhi = stage_boxes[inst.iteration][stage['name']]
        rows_needed = range(-blo[1], R - bhi[1]) if split == 1 else range(RP)
        line('        {')
        if split == 2:
          above, below = set(), set()
          for src, rel, _ in inst.reads:
            sl = slot(src, u, inst.lag - src.lag - rel[2])
            for r in rows_needed:
              if r + rel[1] < 0:
                above.add((src.ident, sl, R + r + rel[1]))
              elif r + rel[1] >= RP:     # (packed: the high halves go past row R)
                below.add((src.ident, sl, r + rel[1] - RP))
          above, below = sorted(above), sorted(below)
          for n in range(max(len(above), len(below))):
            ia, sa, ka = above[n] if n < len(above) else (None, 0, 0)
            ib, sb, kb = below[n] if n < len(below) else (None, 0, 0)
            if pairs:     # the rows as scalars: halves of the pairs that hold them
              last = '%s[%d][%d]' % (ia, sa, ka % RP) if ia else \
                  '%s[%d][%d]' % (ib, sb, kb % RP)
              first = '%s[%d][%d]' % (ib, sb, kb % RP) if ib else last
              half_last = '[%d]' % ((ka if ia else kb) // RP)
              half_first = '[%d]' % (kb // RP) if ib else half_last
            else:
              last = '%s[%d][%d]' % (ia, sa, ka) if ia else '%s[%d][%d]' % (ib, sb, kb)
              first = '%s[%d][%d]' % (ib, sb, kb) if ib else last
              half_last = half_first = ''
            na = 'xa_%s_%d_%d' % (ia, sa, ka) if ia else 'xa_unused%d' % n
            nb = 'xb_%s_%d_%d' % (ib, sb, kb) if ib else 'xb_unused%d' % n
            line('        %s %s[%d], %s[%d];' % (T_in, na, C, nb, C))
            for c in range(C):
              line('        rows_across_halves(%s[%d]%s, %s[%d]%s, %s[%d], %s[%d]);'
                   % (first, c, half_first, last, c, half_last, na, c, nb, c))
            if not ia:
              line('        (void)%s;' % na)
            if not ib:
              line('        (void)%s;' % nb)
        for r in rows_needed:
          for c in range(C):
            def load(tensor, rel, u=u, r=r, c=c, inst=inst, by_name=by_name):
              return operand(inst, by_name[(tensor, tuple(rel))], tuple(rel), u, r, c)
            target = ('out_tile[%d][%d]' % (r, c)) if direct else \
                '%s[%d][%d][%d]' % (inst.ident, slot(inst, u, 0), r, c)
            cell_assignment(stage, target, load, line, '        ')
        line('        }')
        if inst.role == 'lds_out' and pairs:
          for q in range(RP):
            src_row = ('out_tile[%d]' % q) if direct else \
                '%s[%d][%d]' % (inst.ident, slot(inst, u, 0), q)
            line('        { soda_f4 v;%s *(soda_f4*)&handoff[%d][%d][%d][lane * 4] = v; }'
                 % (''.join(' v[%d] = %s[%d][0]; v[%d] = %s[%d][1];' % (
                     2 * c, src_row, c, 2 * c + 1, src_row, c) for c in range(C)),
                    inst.handoff, u % 2, q))
        elif inst.role == 'lds_out':
          for r in range(R):
            src_row = ('out_tile[%d]' % r) if direct else \
                '%s[%d][%d]' % (inst.ident, slot(inst, u, 0), r)
            line('        { %s_lds v;%s *(%s_lds*)&handoff[%d][%d][%d][lane * %d] = v; }'
                 % (vec, ''.join(' v[%d] = %s[%d];' % (c, src_row, c)
                                 for c in range(C)), vec, inst.handoff, u % 2, r, C))
        if inst.final and buffer_io and split == 2:
          def out(r, c):
            return 'out_tile[%d][%d][%d]' % (r % RP, c, r // RP) if pairs else \
                'out_tile[%d][%d]' % (r, c)
          line('        { const i64 z = head + %d;' % (u - L))
          line('          if (z >= z0 && z < z1) {')
          line('            %s* q = g_out + z * plane;' % T_out)
          line('            if (!st_ragged) {')
          line('              const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_'
               'buffer_rsrc(q, 0, (int)plane_bytes, 0x27000);')
          # nt = 4: launches whose box does not fit the Infinity Cache store around
          # the caches (a wave-uniform branch next to the ones this store path has;
          # kernel_common.NT_STREAMING_BYTES); 2: always
          for aux in ((2, 0) if nt & 4 else (2,) if nt & 2 else (0,)):
            if nt & 4:
              line('              %s {' % ('if (st_streaming)' if aux else 'else'))
            for r in range(R):
              line('              { %s v;%s __builtin_amdgcn_raw_buffer_store_%s('
                   '__builtin_bit_cast(%s, v), rs, (st_rows >> %d) & 1u ? lane_byte : '
                   '0xfffffff0u, (unsigned)(%d * W * %d), %d); }' % (
                       vec, ''.join(' v[%d] = %s;' % (c, out(r, c)) for c in range(C)),
                       BUF_SUFFIX, BUF_TYPE, r, r, elem, aux))
            if nt & 4:
              line('              }')
          line('            } else {')
          for r in range(R):
            line('            if (%d >= st_r0 && %d < st_r1) {' % (r, r))
            line('              if (st_full) { %s v;%s *(%s*)((char*)(q + %d * W) + '
                 'lane_byte) = v; }' % (
                     vec, ''.join(' v[%d] = %s;' % (c, out(r, c))
                                  for c in range(C)), vec, r))
            line('              else {%s }' % ''.join(
                ' if (st_col%d) q[%s + x + %d] = %s;'
                % (c, row_off(r), c, out(r, c)) for c in range(C)))
            line('            }')
          line('            }')
          line('          } }')
        elif inst.final:
          line('        { const i64 z = head + %d;' % (u - L))
          line('          if (z >= z0 && z < z1) {')
          line('            %s* q = g_out + z * plane;' % T_out)
          for r in (range(y_lo, R - y_hi) if split == 1 else range(R)):
            if split == 1:
              line('            if (yb + %d >= a.box_lo[1] && yb + %d < a.box_hi[1]) {'
                   % (r, r))
              line('              if (x >= st_lo && x + %d <= st_hi) { %s v;%s '
                   '*(%s*)(q + %s + x) = v; }' % (
                       C, vec, ''.join(' v[%d] = out_tile[%d][%d];' % (c, r, c)
                                       for c in range(C)), vec, row_off(r)))
              line('              else {%s }' % ''.join(
                  ' if (x + %d >= st_lo && x + %d < st_hi) q[%s + x + %d] = '
                  'out_tile[%d][%d];' % (c, c, row_off(r), c, r, c)
                  for c in range(C)))
              line('            }')
            else:   # per-lane predicates computed once, outside the plane loop
              line('            if (%d >= st_r0 && %d < st_r1) {' % (r, r))
              def out(r, c):
                return 'out_tile[%d][%d][%d]' % (r % RP, c, r // RP) if pairs else \
                    'out_tile[%d][%d]' % (r, c)
              line('              if (st_full) { %s v;%s *(%s*)((char*)(q + %d * W) + '
                   'lane_byte) = v; }' % (
                       vec, ''.join(' v[%d] = %s;' % (c, out(r, c))
                                    for c in range(C)), vec, r))
              line('              else {%s }' % ''.join(
                  ' if (st_col%d) q[%s + x + %d] = %s;'
                  % (c, row_off(r), c, out(r, c)) for c in range(C)))
              line('            }')
          line('          } }')
      line('      }')
      line('      %s();' % barrier)

  if self_load:
    # all rows of a lane's plane tile from the ring in one burst of reads the
    # compiler does not see as LDS reads (it would wait for ALL LDS-direct loads
    # in front of a visible one, kernel_common: soda_lds_read_f4)
    width = {8: 'b64', 16: 'b128'}.get(C * elem)
    if width is None:
      raise NotFusable('self-loading ring: 8- or 16-byte lanes')
    line('typedef %s %s_row __attribute__((ext_vector_type(%d)));' % (T_in, vec, C))
    line('DEV void soda_ring_read_%s(const void* base, %s) {' % (
        name, ', '.join('%s& v%d_%d' % (T_in, r, c) for r in range(R) for c in range(C))))
    line('  %s;' % '; '.join('%s_row t%d' % (vec, r) for r in range(R)))
    line('  asm volatile(%s' % ''.join(
        '"ds_read_%s %%%d, %%%d offset:%d\\n\\t"\n               ' % (
            width, r, R, r * LX * C * elem) for r in range(R)))
    line('               "s_waitcnt lgkmcnt(0)"')
    line('               : %s' % ', '.join('"=&v"(t%d)' % r for r in range(R)))
    line('               : "v"((unsigned)(unsigned long long)base) : "memory");')
    for r in range(R):
      line('  ' + ' '.join('v%d_%d = t%d[%d];' % (r, c, r, c) for c in range(C)))
    line('}')

  line('template <bool INTERIOR>')
  line('DEV void %s_tile(const soda_hip_args& a, const i64 xs, const i64 x, '
       'const i64 yb, const i64 wx, const i64 wy, const i64 z0, const i64 z1, '
       'const int wave, const int lane,' % name)
  line('    %s (*handoff)[2][%d][%d], %s (*in_ring)[%d][%d]) {'
       % (T_in, RP, LANES * C * (2 if pairs else 1), T_in, TR if loader else 1,
          LX * C if loader else 1))
  line('  const i64 W = a.dims[0], H = a.dims[1], D = a.dims[2];')
  line('  const i64 plane = W * H;')
  line('  const i64 st_lo = xs > a.box_lo[0] ? xs : a.box_lo[0];')
  line('  const i64 st_hi = xs + %d < a.box_hi[0] ? xs + %d : a.box_hi[0];'
       % (w_out, w_out))
  if split == 1:
    line('  i64 row_y[%d];' % R)
    for r in range(R):
      line('  { i64 y = yb + %d; y = y < 0 ? 0 : (y > H - 1 ? H - 1 : y); '
           'row_y[%d] = y * W; }' % (r, r))
    line('  (void)row_y;')
  if split == 2:
    line('  const int ly = lane >> 5;')
    line('  const i64 yt = wy + ly * %d;   // first tile row of this lane' % R)
    line('  const i64 lane_row = yt * W; (void)lane_row;')
    # fast path: one 32-bit byte offset per lane on top of wave-uniform row
    # pointers (the launcher takes it only when a plane is below 4 GiB)
    line('  const unsigned lane_byte = (unsigned)((lane_row + x) * %d); '
         '(void)lane_byte;' % elem)
    line('  const i64 st_ylo = a.box_lo[1] > yb + %d ? a.box_lo[1] : yb + %d;'
         % (y_lo, y_lo))
    line('  const i64 st_yhi = a.box_hi[1] < yb + %d ? a.box_hi[1] : yb + %d;'
         % (TR - y_hi, TR - y_hi))
    line('  const int st_r0 = (int)(st_ylo - yt), st_r1 = (int)(st_yhi - yt);')
    line('  const bool st_full = x >= st_lo && x + %d <= st_hi;' % C)
    for c in range(C):
      line('  const bool st_col%d = x + %d >= st_lo && x + %d < st_hi;' % (c, c, c))
    line('  (void)ly; (void)st_r0; (void)st_r1; (void)st_full;%s'
         % ''.join(' (void)st_col%d;' % c for c in range(C)))
    if buffer_io:
      # rows this lane stores, as a bit mask (empty when the lane's columns are
      # not all inside the strip's store range); ragged = some lane straddles an
      # edge of the range: such a tile takes the predicated path
      line('  unsigned st_rows = 0;')
      line('  for (int r = 0; r < %d; ++r) if (st_full && r >= st_r0 && r < st_r1) '
           'st_rows |= 1u << r;' % R)
      if nt & 4:
        line('  const bool st_streaming = (a.box_hi[0] - a.box_lo[0]) * (a.box_hi[1] - '
             'a.box_lo[1]) * (a.box_hi[2] - a.box_lo[2]) * %d > %dll;' % (
                 2 * elem, kernel_common.NT_STREAMING_BYTES))
      line('  const bool st_ragged = __builtin_amdgcn_ballot_w64(!st_full && (%s)) != 0;'
           % ' || '.join('st_col%d' % c for c in range(C)))
      line('  const i64 plane_bytes = W * H * %d; (void)plane_bytes; (void)st_rows; '
           '(void)st_ragged;' % elem)
  line('  const %s* __restrict__ g_in = (const %s*)a.tensor[%d];'
       % (T_in, T_in, index[spec['inputs'][0]['name']]))
  line('  %s* __restrict__ g_out = (%s*)a.tensor[%d];' % (T_out, T_out,
                                                           index[out_name]))
  line('  (void)g_in; (void)g_out; (void)st_lo; (void)st_hi; (void)W; (void)D; '
       '(void)H; (void)plane; (void)wx; (void)wy; (void)in_ring;')
  line('  const i64 steps = (z1 - z0) + %d;' % (L + lo[2]))
  if extra:
    # The loader wavefront: %d instructions per plane tile, nothing comes back to
    # registers.
    line('  if (wave == 0) {')
    line('    const i64 dma_lane = (wy + (lane / %d)) * W + wx + (lane %% %d) * %d;'
         % (per_row, per_row, 16 // elem))
    line('    i64 head = z0 - %d;' % lo[2])
    for k in range(PF):
      ring_load('%d' % k, 'head + %d' % k)
    line('    for (i64 n = 0; n < steps; n += %d, head += %d) {' % (period, period))
    for u in range(period):
      ring_load('%d' % ((u + PF) % RS), 'head + %d' % (u + PF))
      # plane head+u (issued PF steps ago) has landed once at most the %d most
      # recent loads are still in flight
      line('      __builtin_amdgcn_s_waitcnt(%d);  // vmcnt(%d)'
           % (vmcnt(PF * loads), PF * loads))
      # a bare barrier: a release fence here would make the compiler wait for
      # ALL LDS-direct loads (vmcnt(0)) and undo the prefetch; the explicit
      # wait above is what makes plane head+u visible before the barrier
      line('      __builtin_amdgcn_s_barrier();')
    line('    }')
    line('    __builtin_amdgcn_s_waitcnt(%d);  // drain before the LDS is released'
         % vmcnt(0))
    line('  }')
  for g, mine in enumerate(per_wave):
    line('  %sif (wave == %d) {' % ('' if g == 0 and not extra else 'else ',
                                   g + extra))
    levels = ([int(v) for v in (prio if isinstance(prio, (list, tuple)) else str(prio).split('/'))] + [0] * groups)[:groups]
    # issue priority per wavefront, as in kernel_stream2d_wp (first one raised:
    # heat3d 380 -> 374 us per launch, cfg5 6.77 -> 6.67 ms)
    if levels[g]:
      line('    __builtin_amdgcn_s_setprio(%d);' % levels[g])
    for inst in mine:
      if inst.keep:
        line('    %s %s[%d][%d][%d];' % ('pk2' if pairs else builtin_type(inst.c_type),
                                         inst.ident, inst.keep, RP, C))
        for k in range(inst.keep):
          for r in range(RP):
            line('    ' + ' '.join('%s[%d][%d][%d] = %s;' % (
                inst.ident, k, r, c, 'pk2{0.0f, 0.0f}' if pairs else '0')
                                   for c in range(C)))
    line('    i64 head = z0 - %d;' % lo[2])
    if self_load and g == 0:      # the first PF planes of the ring
      line('    const i64 dma_lane = (wy + (lane / %d)) * W + wx + (lane %% %d) * %d;'
           % (per_row, per_row, 16 // elem))
      for k in range(PF):
        ring_load('%d' % k, 'head + %d' % k, '    ')
    line('    for (i64 n = 0; n < steps; n += %d, head += %d) {' % (period, period))
    emit_body(mine)
    line('    }')
    if self_load and g == 0:
      line('    __builtin_amdgcn_s_waitcnt(%d);  // no load may outlive the LDS'
           % vmcnt(0))
    line('  }')
  line('}')
  line('')
  occupancy = ''
  if waves_per_eu > 0:
    occupancy = ' __attribute__((amdgpu_waves_per_eu(%d, %d)))' % (waves_per_eu,
                                                                   waves_per_eu)
  line('GLOBAL WG_SIZE(%d)%s void %s(soda_hip_args a) {'
       % ((groups + extra) * LANES, occupancy, name))
  line('  __attribute__((shared)) %s handoff[%d][2][%d][%d];' % (
      T_in, max(1, groups - 1), RP, LANES * C * (2 if pairs else 1)))
  line('  __attribute__((shared)) %s in_ring[%d][%d][%d];' % (
      T_in, RS if loader else 1, TR if loader else 1, LX * C if loader else 1))
  line('  const int lane = lane_id();')
  line('  const int wave = __builtin_amdgcn_readfirstlane('
       '__builtin_amdgcn_workitem_id_x() >> 6);')
  if xcd_tiles:
    # 1-D grid; L = 8 * i + c is the i-th workgroup of XCD c (round-robin deal).
    # The XCD's j-th super-tile (j = i / S) is global super-tile 8 j + c.
    line('  const unsigned L = __builtin_amdgcn_workgroup_id_x();')
    line('  const unsigned SX = (unsigned)a.param[1] & 0xffffu, '
         'SY = (unsigned)a.param[1] >> 16;')
    line('  const unsigned nsx = (unsigned)a.param[2] & 0xffffu, '
         'nsy = (unsigned)a.param[2] >> 16;')
    line('  const unsigned S = SX * SY, i = L >> 3, g = (i / S) * 8u + (L & 7u), '
         'within = i % S;')
    line('  const unsigned block_x = (g % nsx) * SX + within % SX;')
    line('  const unsigned block_y = ((g / nsx) % nsy) * SY + within / SX;')
    line('  const unsigned block_z = g / (nsx * nsy);')
  elif xcd_remap:
    # workgroups are dealt round-robin over the 8 XCDs (ids b and b+8 share an
    # L2): re-deal them so that every XCD works on a contiguous run of tiles (x
    # fastest, then y, then the z chunk) and neighbouring tiles, which read each
    # other's halo, share an L2.  Any placement is correct; bijective.
    line('  const unsigned gx = __builtin_amdgcn_grid_size_x() / %d;'
         % ((groups + extra) * LANES))
    line('  const unsigned gy = __builtin_amdgcn_grid_size_y();')
    line('  const unsigned total = gx * gy * __builtin_amdgcn_grid_size_z();')
    line('  const unsigned lin = __builtin_amdgcn_workgroup_id_x() + gx * ('
         '__builtin_amdgcn_workgroup_id_y() + gy * __builtin_amdgcn_workgroup_id_z());')
    line('  const unsigned xcd = lin & 7u, within = lin >> 3;')
    line('  const unsigned share = total >> 3, extra = total & 7u;')
    line('  const unsigned tile_id = xcd * share + (xcd < extra ? xcd : extra) + within;')
    line('  const unsigned block_x = tile_id % gx, block_y = (tile_id / gx) % gy, '
         'block_z = tile_id / (gx * gy);')
  else:
    line('  const unsigned block_x = __builtin_amdgcn_workgroup_id_x();')
    line('  const unsigned block_y = __builtin_amdgcn_workgroup_id_y();')
    line('  const unsigned block_z = __builtin_amdgcn_workgroup_id_z();')
  line('  const i64 x_origin = a.box_lo[0] - a.box_lo[0] %% %d;' % C)
  line('  const i64 xs = x_origin + (i64)block_x * %d;' % w_out)
  line('  if (xs >= a.box_hi[0]) return;')
  line('  const i64 yb = a.box_lo[1] + (i64)block_y * %d - %d;' % (r_out, y_lo))
  line('  const i64 chunk = a.param[0] > 0 ? a.param[0] : %d;' % chunk_planes)
  line('  const i64 z0 = a.box_lo[2] + (i64)block_z * chunk;')
  if xcd_tiles:   # padding tiles of the last super-tiles
    line('  if (yb + %d >= a.box_hi[1] || z0 >= a.box_hi[2]) return;' % y_lo)
  line('  const i64 z1 = z0 + chunk < a.box_hi[2] ? z0 + chunk : a.box_hi[2];')
  if split == 1:
    line('  const i64 x = xs - %d + lane * %d;' % (halo_lo, C))
    line('  const bool interior = xs - %d >= 0 && xs - %d + %d <= a.dims[0];'
         % (halo_lo, halo_lo, LX * C))
    line('  if (interior) %s_tile<true>(a, xs, x, yb, 0, yb, z0, z1, wave, lane, '
         'handoff, in_ring);' % name)
    line('  else %s_tile<false>(a, xs, x, yb, 0, yb, z0, z1, wave, lane, handoff, '
         'in_ring);' % name)
  else:
    # A tile that would overhang the array is moved inwards instead: its window
    # then lies inside the array, every load is unguarded, and it still stores
    # only its own output cells (which the moved window covers with the same
    # values).  The launcher uses this kernel only for arrays of at least one
    # tile (soda_hip_kernel.min_extent).
    line('  i64 wx = xs - %d;' % halo_lo)
    line('  if (wx + %d > a.dims[0]) wx = a.dims[0] - %d;' % (LX * C, LX * C))
    line('  if (wx < 0) wx = 0;')
    line('  i64 wy = yb;')
    line('  if (wy + %d > a.dims[1]) wy = a.dims[1] - %d;' % (TR, TR))
    line('  if (wy < 0) wy = 0;')
    line('  const i64 x = wx + (lane & %d) * %d;' % (LX - 1, C))
    line('  %s_tile<true>(a, xs, x, yb, wx, wy, z0, z1, wave, lane, handoff, in_ring);'
         % name)
  line('}')
  entry = dict(name=name, kind='fused', depth=depth, stage=-1,
               block=[(groups + extra) * LANES, 1, 1],
               tile=[w_out, r_out, chunk_planes, 1], origin_align=C,
               fill_rows=L + lo[2], cols=C, rows=R, prefetch=prefetch,
               period=period, est_vgprs=est_vgprs, w_out=w_out, r_out=r_out,
               groups=groups, lds_bytes=lds_bytes, split=split, loader=loader,
               pairs=pairs, xcd_tiles=4 * int(bool(xcd_tiles)), buffer_io=int(bool(buffer_io)),
               min_extent=[LX * C, TR] if split == 2 else [0, 0])
  if nt and buffer_io and split == 2:
    entry['nt'] = int(nt)
  return '\n'.join(o) + '\n', entry
