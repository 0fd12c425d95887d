"""Multi-GPU execution: one process per GPU, the grid cut into slabs along the
outermost (last, streamed) dimension, ghost rows exchanged between neighbours.

The reference has nothing distributed (single FPGA), so
this layer is new.  What it must preserve is the reference's semantics: no
boundary condition, the valid box shrinks by the stencil window every iteration
(reference core.py:794-835, host.py:1082-1091).  That makes the chain of slabs
OPEN: the first and last rank have one neighbour, nothing wraps around.

Per super-step of E iterations a rank
  1. sends its first E*r_hi own rows down and its last E*r_lo own rows up, and
     receives its neighbours' into its ghost rows  (one batched isend/irecv
     pair per neighbour: RCCL over xGMI on GPUs, gloo in the CPU tests);
  2. advances its slab E iterations in place of one: the sweep is told that the
     ghost sides are fully valid (`valid_lo/hi = 0` there) and that the global
     sides carry the margin of the iterations done so far, so after the sweep
     exactly the rank's own rows are defined again.
Halo cells are recomputed instead of exchanged inside a super-step, which trades
`E*r` redundant rows per side for E times fewer (latency-bound) exchanges.

The sweep itself is an injected `engine` (the HIP program in production; the
tests inject a CPU engine built on the oracle to exercise this logic under
gloo without a GPU).
"""
import os
import time

import numpy as np


def slab_bounds(extent, world):
  """Own rows [start, stop) of every rank: as even as possible."""
  base, extra = divmod(extent, world)
  out, start = [], 0
  for r in range(world):
    stop = start + base + (1 if r < extra else 0)
    out.append((start, stop))
    start = stop
  return out


class SlabPlan:
  """Geometry of one rank's slab for a program with per-iteration radius
  (r_lo, r_hi) along the last dimension."""

  def __init__(self, dims, rank, world, r_lo, r_hi, exchange):
    self.dims = list(dims)
    self.rank, self.world = rank, world
    self.r_lo, self.r_hi = r_lo, r_hi
    bounds = slab_bounds(dims[-1], world)
    self.start, self.stop = bounds[rank]
    self.own = self.stop - self.start
    self.has_lo = rank > 0
    self.has_hi = rank < world - 1
    smallest = min(b - a for a, b in bounds)
    reach = max(r_lo, r_hi, 1)
    if world > 1 and smallest < reach:
      # a ghost region of even ONE iteration would be deeper than a neighbour's
      # own rows: the exchange would ship rows the neighbour does not own
      raise ValueError(
          'cannot cut %d rows into %d slabs: the smallest slab (%d rows) is '
          'thinner than the stencil reach (%d)' % (dims[-1], world, smallest, reach))
    # a ghost region cannot be deeper than the neighbour's own rows
    self.exchange = max(1, min(exchange, smallest // reach)) if world > 1 else exchange
    self.ghost_lo = self.exchange * r_lo if self.has_lo else 0
    self.ghost_hi = self.exchange * r_hi if self.has_hi else 0
    self.local_extent = self.ghost_lo + self.own + self.ghost_hi
    self.local_dims = self.dims[:-1] + [self.local_extent]
    # rows the neighbours need from us
    self.send_up = self.exchange * r_lo if self.has_hi else 0    # our last rows
    self.send_down = self.exchange * r_hi if self.has_lo else 0  # our first rows

  def valid_margins(self, done, margins_of):
    """valid_lo/valid_hi of the slab's input after `done` iterations, given
    `margins_of(k)` -> (lo, hi) tuples of the global margins after k iterations."""
    lo, hi = margins_of(done)
    lo, hi = list(lo), list(hi)
    if self.has_lo:
      lo[-1] = 0
    else:
      lo[-1] = lo[-1]          # global edge: rows [0, margin) are undefined
    if self.has_hi:
      hi[-1] = 0
    return lo, hi


def _corrupt_received(rank, cut):
  """Test hook (SODA_HIP_TUNING=1 only, like the library's own): SODA_DIST_CORRUPT_GHOST=R
  makes rank R damage one cell of the first rows it receives in every exchange - what the
  self-check of bench_main must catch; SODA_DIST_CORRUPT_CUT=recut|static confines it to
  the exchanges of that cut (a fault of one cut only: the check falls back to the other)."""
  if os.environ.get('SODA_HIP_TUNING') != '1':
    return False
  who = os.environ.get('SODA_DIST_CORRUPT_GHOST')
  only = os.environ.get('SODA_DIST_CORRUPT_CUT')
  return who is not None and who.lstrip('-').isdigit() and int(who) == rank and \
      only in (None, '', cut)


def _damage(rows):
  # the middle cell of the FIRST and of the LAST row (plane) of the block: one of the two
  # lies next to the rank's own rows.  (A cell at the grid's edge feeds only cells that
  # leave the valid box anyway; and an averaging stencil damps a bump of 1.0 that lies 24
  # rows away below one float32 ulp within 48 iterations - measured: bit-identical rows.)
  import sys
  for k in {0, rows.shape[0] - 1}:
    block = rows[k]
    while block.dim() > 1:
      block = block[block.shape[0] // 2]
    block[block.shape[0] // 2:block.shape[0] // 2 + 1] += 1
  sys.stderr.write('soda_hip dist: TEST HOOK damaged the first and last of %d received rows\n'
                   % rows.shape[0])


def exchange_ghosts(array, plan, dist, backend_ops=None):
  """array: torch tensor of shape reversed(local_dims) (outer dim first).
  Fills the ghost rows from the neighbours' own rows."""
  if plan.world == 1:
    return
  ops = []
  g_lo, g_hi, own = plan.ghost_lo, plan.ghost_hi, plan.own
  first_own = g_lo
  last_own = g_lo + own
  # bytes are bytes: unsigned element types travel as the signed type of the same
  # width (RCCL has no uint16 / uint32 / uint64)
  signed = {'torch.uint16': 'int16', 'torch.uint32': 'int32', 'torch.uint64': 'int64'}
  if str(array.dtype) in signed:
    import torch
    array = array.view(getattr(torch, signed[str(array.dtype)]))

  def add(op, rows, peer):     # a one-sided window has nothing to ship one way:
    if rows.shape[0] > 0:      # both sides skip that (empty) message
      ops.append(dist.P2POp(op, rows, peer))
  if plan.has_lo:
    # lower neighbour: it needs our first send_down rows, we need its last rows
    add(dist.isend, array[first_own:first_own + plan.send_down], plan.rank - 1)
    add(dist.irecv, array[0:g_lo], plan.rank - 1)
  if plan.has_hi:
    add(dist.isend, array[last_own - plan.send_up:last_own], plan.rank + 1)
    add(dist.irecv, array[last_own:last_own + g_hi], plan.rank + 1)
  if ops:
    for req in dist.batch_isend_irecv(ops):
      req.wait()
    if _corrupt_received(plan.rank, 'static'):
      if plan.has_lo and g_lo:
        _damage(array[0:g_lo])
      elif plan.has_hi and g_hi:
        _damage(array[last_own:last_own + g_hi])


class SerialSchedule:
  """How run_slab orders the exchange against the sweeps: here, not at all (the
  CPU engines of the gloo tests, and the reference point on GPUs)."""
  overlapped = False
  # True: the exchanges are left out (stale ghost rows; results are garbage, launch
  # times are not - they do not depend on the data): what the compute of a rank's
  # slab costs by itself (bench_main: compute_only_ms_per_step)
  skip_exchange = False

  def before_super_step(self):
    pass

  def exchange(self, fn):
    if not self.skip_exchange:
      fn()

  def after_bands(self):
    pass


def band_plan(plan, step):
  """Row ranges (local indices, outermost dimension) of one super-step of `step`
  iterations cut into the two bands the neighbours are waiting for and the rest:
  [(r0, r1, cut_lo, cut_hi)] = sweep the sub-array of rows [r0, r1) whose lower /
  upper side is cut inside valid data (margin 0 there; an uncut side is the
  slab's own side and keeps the margin the caller computed).  The output rows of
  the pieces tile the slab's own rows exactly; every piece reads only rows that
  are valid before the super-step (own rows and freshly exchanged ghost rows)."""
  first_own, last_own = plan.ghost_lo, plan.ghost_lo + plan.own
  reach_lo, reach_hi = step * plan.r_lo, step * plan.r_hi
  bands, lo_edge, hi_edge = [], first_own, last_own
  if plan.has_lo:      # the lower neighbour's ghost rows: our first send_down rows
    bands.append((first_own - reach_lo, first_own + plan.send_down + reach_hi,
                  True, True))
    lo_edge = first_own + plan.send_down
  if plan.has_hi:
    bands.append((last_own - plan.send_up - reach_lo, last_own + reach_hi, True, True))
    hi_edge = last_own - plan.send_up
  interior = (lo_edge - reach_lo if plan.has_lo else 0,
              hi_edge + reach_hi if plan.has_hi else plan.local_extent,
              plan.has_lo, plan.has_hi)
  return bands, interior


def run_slab(engine, plan, arrays, iterate, margins_of, dist, ghosts_ready=False,
             schedule=None):
  """Advances the slab `iterate` iterations.  `arrays` = [A, B, C]: A holds the
  level-0 slab (own rows filled, ghost rows anything) and is not written; the
  result ends up in the returned array (B or C).  Returns (array, exchanges).

  ghosts_ready: A's ghost rows already hold the neighbours' level-0 rows (the
  input was distributed with its halo, or an earlier call exchanged them - A is
  never written, so they stay), and the first exchange is skipped.

  schedule: SerialSchedule (default) exchanges, then sweeps the whole slab.  An
  overlapping schedule (StreamSchedule on GPUs) makes every super-step but the
  last produce the rows its neighbours need FIRST, as two thin band sweeps, hands
  them to the exchange of the next super-step on a side stream, and sweeps the
  interior meanwhile."""
  a, b, c = arrays
  schedule = schedule or SerialSchedule()
  src, done, exchanges = a, 0, 0
  dst_cycle = [b, c]
  k = 0
  pending = ghosts_ready      # src's ghost rows are (being) filled already
  while done < iterate:
    if not pending:
      schedule.exchange(lambda src=src: exchange_ghosts(src, plan, dist))
      exchanges += 1
    schedule.before_super_step()       # ghost rows of src have landed
    pending = False
    step = min(plan.exchange, iterate - done)
    lo, hi = plan.valid_margins(done, margins_of)
    dst = dst_cycle[k % 2]
    more = done + step < iterate
    small = plan.own < 2 * (plan.send_down + plan.send_up) + 1
    if schedule.overlapped and more and plan.world > 1 and not small:
      bands, interior = band_plan(plan, step)

      def piece(r0, r1, cut_lo, cut_hi):
        plo, phi = list(lo), list(hi)
        if cut_lo:
          plo[-1] = 0
        if cut_hi:
          phi[-1] = 0
        # final_only: a piece's sub-array of dst overlaps rows another piece has
        # finished (and that are being sent): only the piece's LAST launch, whose
        # box is exactly its own rows, may write dst
        engine.sweep(src, dst, plan.local_dims, step, plo, phi, rows=(r0, r1),
                     final_only=True)
      for band in bands:
        piece(*band)
      schedule.after_bands()
      schedule.exchange(lambda dst=dst: exchange_ghosts(dst, plan, dist))
      exchanges += 1
      pending = True
      piece(*interior)
    else:
      engine.sweep(src, dst, plan.local_dims, step, lo, hi)
    src = dst
    done += step
    k += 1
  schedule.before_super_step()
  return src, exchanges


# ---------------------------------------------------------------------------
# slabs re-cut to the shrinking valid box
# ---------------------------------------------------------------------------
# The static cut above gives every rank the same rows for the whole run.  Under the
# reference's semantics the valid box shrinks by the window every iteration
# (core.py:794-835, host.py:1082-1091): the first and last ranks lose r rows per
# iteration, the ranks in the middle none, and a step costs what the slowest rank costs -
# jacobi2d 16384^2 x1000 on 4 or 8 ranks is capped at 93.9 % by that alone, jacobi3d
# 512^3 x200 on 8 ranks at 61 % (ranks 0 and 7 own planes 0-63 and 448-511 and have
# nothing left to do after iteration 64).  Here every super-step cuts the rows its
# OUTPUT level defines evenly again; rows that change owner travel in the same grouped
# send / recv as the ghost rows.
def _even_cut(lo, hi, world):
  """world + 1 cut points of [lo, hi) (an empty range: all equal to lo)."""
  pts, at = [lo], lo
  for a, b in slab_bounds(max(0, hi - lo), world):
    at += b - a
    pts.append(at)
  return pts


def _intersect(a, b):
  lo, hi = max(a[0], b[0]), min(a[1], b[1])
  return (lo, hi) if hi > lo else None


class RecutPlan:
  """Geometry of a run whose slabs are cut afresh every super-step.

  Global rows of the outermost dimension throughout.  Super-step s starts after
  `done` iterations and advances `step` = min(exchange, iterate - done) more:
    owned[s][r] = rows of its INPUT level rank r holds (it produced them; s = 0: the even
                  cut of the whole grid, how the input arrives),
    cuts[s]     = world + 1 cut points of the rows its OUTPUT level defines,
                  [(done + step) r_lo, rows - (done + step) r_hi), cut evenly,
    need[s][r]  = rows of the input level rank r reads: its output rows widened by the
                  reach of `step` iterations (None: the rank has no output rows).
  Before super-step s rank r sends need[s][q] & owned[s][r] to every other rank q and
  receives need[s][r] & owned[s][q] - ghost rows and rows changing owner alike; with
  thin slabs (cfg5 on 8 ranks: 14 planes each at the end, 16 or 32 of reach) the
  partner is not always the immediate neighbour.  A rank's arrays span the hull of
  everything it ever holds, [base, base + local_extent), at a fixed offset."""

  def __init__(self, dims, rank, world, r_lo, r_hi, exchange, iterate):
    self.dims = list(dims)
    self.rank, self.world = rank, world
    self.r_lo, self.r_hi = r_lo, r_hi
    self.exchange = max(1, min(exchange, iterate))
    self.iterate = iterate
    rows = dims[-1]
    self.steps = []                 # (done, step) per super-step
    self.owned, self.cuts, self.need = [], [], []
    level = [(a, b) for a, b in slab_bounds(rows, world)]
    done = 0
    while done < iterate:
      step = min(self.exchange, iterate - done)
      lo, hi = (done + step) * r_lo, rows - (done + step) * r_hi
      cut = _even_cut(lo, max(lo, hi), world)
      need = [(cut[r] - step * r_lo, cut[r + 1] + step * r_hi)
              if cut[r + 1] > cut[r] else None for r in range(world)]
      self.steps.append((done, step))
      self.owned.append(level)
      self.cuts.append(cut)
      self.need.append(need)
      level = [(cut[r], cut[r + 1]) for r in range(world)]
      done += step
    self.final = level              # rows of the result every rank ends up with
    self.start, self.stop = self.owned[0][rank]          # level-0 rows (the input)
    self.own = self.stop - self.start
    held = [self.owned[0][rank]] + [n[rank] for n in self.need if n[rank]]
    self.base = min(a for a, _ in held)
    self.local_extent = max(b for _, b in held) - self.base
    self.local_dims = self.dims[:-1] + [self.local_extent]
    self.ghost_lo = self.start - self.base       # where the level-0 rows go
    self.final_rows = self.final[rank]           # global; local = minus base

  def local(self, rows):
    return rows[0] - self.base, rows[1] - self.base

  def messages(self, s):
    """(sends, recvs) of this rank before super-step s: [(peer, (lo, hi) global rows)],
    ascending peers - both sides enumerate a pair's rows from the same tables."""
    sends, recvs = [], []
    mine = self.owned[s][self.rank]
    for q in range(self.world):
      if q == self.rank:
        continue
      if self.need[s][q] and mine[1] > mine[0]:
        rows = _intersect(self.need[s][q], mine)
        if rows:
          sends.append((q, rows))
      theirs = self.owned[s][q]
      if self.need[s][self.rank] and theirs[1] > theirs[0]:
        rows = _intersect(self.need[s][self.rank], theirs)
        if rows:
          recvs.append((q, rows))
    return sends, recvs

  def pieces(self, s):
    """Super-step s, bands first: ([band, ...], interior), each (out_lo, out_hi) global
    output rows - the bands are the rows other ranks need for super-step s + 1, the
    interior the rest of this rank's output.  None when there is nothing to gain: the last
    super-step, no output rows, nobody waiting, or bands that meet (a thin slab)."""
    if s + 1 >= len(self.steps):
      return None
    lo, hi = self.cuts[s][self.rank], self.cuts[s][self.rank + 1]
    if hi <= lo:
      return None
    below = [n[1] for n in self.need[s + 1][:self.rank] if n and n[1] > lo]
    above = [n[0] for n in self.need[s + 1][self.rank + 1:] if n and n[0] < hi]
    b_lo = min(hi, max(below)) if below else lo
    b_hi = max(lo, min(above)) if above else hi
    if (b_lo == lo and b_hi == hi) or b_lo >= b_hi:
      return None
    bands = ([(lo, b_lo)] if b_lo > lo else []) + ([(b_hi, hi)] if b_hi < hi else [])
    return bands, (b_lo, b_hi)

  def max_rows_per_iteration(self):
    """Sum over the super-steps of the rows the BUSIEST rank sweeps per iteration (its
    output rows plus the reach it recomputes, averaged over the step's iterations) - what
    a step costs when a row costs the same everywhere; for DESIGN.md's before / after
    table."""
    total = 0
    for (done, step), need in zip(self.steps, self.need):
      widest = max([b - a for n in need if n for a, b in [n]] or [0])
      # iteration i of the step sweeps the input rows minus i reaches
      total += sum(max(0, widest - i * (self.r_lo + self.r_hi)) for i in range(1, step + 1))
    return total


def static_max_rows_per_iteration(dims, world, r_lo, r_hi, exchange, iterate):
  """RecutPlan.max_rows_per_iteration for the static even cut: per super-step the rows
  the busiest rank sweeps (own rows + ghost rows, minus what has left the valid box at a
  global edge, minus one reach per iteration), summed over the run."""
  rows = dims[-1]
  plans = [SlabPlan(dims, r, world, r_lo, r_hi, exchange) for r in range(world)]
  total, done = 0, 0
  while done < iterate:
    step = min(plans[0].exchange, iterate - done)
    worst = 0
    for p in plans:
      lo = max(p.start - p.ghost_lo, done * r_lo)
      hi = min(p.stop + p.ghost_hi, rows - done * r_hi)
      worst = max(worst, sum(max(0, hi - lo - i * (r_lo + r_hi)) for i in range(1, step + 1)))
    total += worst
    done += step
  return total


def exchange_rows(array, plan, s, dist):
  """array: torch tensor of shape reversed(plan.local_dims).  The grouped send / recv
  of RecutPlan.messages(s)."""
  if plan.world == 1:
    return
  signed = {'torch.uint16': 'int16', 'torch.uint32': 'int32', 'torch.uint64': 'int64'}
  if str(array.dtype) in signed:
    import torch
    array = array.view(getattr(torch, signed[str(array.dtype)]))
  sends, recvs = plan.messages(s)
  ops = []
  for peer, rows in sends:
    a, b = plan.local(rows)
    ops.append(dist.P2POp(dist.isend, array[a:b], peer))
  for peer, rows in recvs:
    a, b = plan.local(rows)
    ops.append(dist.P2POp(dist.irecv, array[a:b], peer))
  if ops:
    for req in dist.batch_isend_irecv(ops):
      req.wait()
    if recvs and _corrupt_received(plan.rank, 'recut'):
      a, b = plan.local(recvs[0][1])
      _damage(array[a:b])


def run_recut(engine, plan, arrays, margins_of, dist, ghosts_ready=False, schedule=None):
  """run_slab for a RecutPlan: advances the rank's rows plan.iterate iterations;
  arrays = [A, B, C] of shape reversed(plan.local_dims), A holding the level-0 rows at
  [plan.ghost_lo, plan.ghost_lo + plan.own) and never written by a sweep.  Returns (array,
  exchanges): the rank's rows of the result, plan.final_rows, are at plan.local(...) of
  the returned array.

  Every sweep is a sweep of the SUB-ARRAY of the rows it reads with the outer sides
  declared valid (the rows there were produced or received): its output box is exactly
  the rank's rows of the output level, on every rank alike - the first and last rank read
  from the edge of the valid rows, which is where their sub-arrays start."""
  a, b, c = arrays
  schedule = schedule or SerialSchedule()
  src, exchanges = a, 0
  dst_cycle = [b, c]
  pending = ghosts_ready
  last = len(plan.dims) - 1
  for s, (done, step) in enumerate(plan.steps):
    if not pending:
      schedule.exchange(lambda src=src, s=s: exchange_rows(src, plan, s, dist))
      exchanges += 1
    schedule.before_super_step()
    pending = False
    lo, hi = margins_of(done)
    lo, hi = list(lo), list(hi)
    lo[last] = hi[last] = 0
    dst = dst_cycle[s % 2]

    def piece(out_rows, final_only):
      r0 = out_rows[0] - step * plan.r_lo - plan.base
      r1 = out_rows[1] + step * plan.r_hi - plan.base
      engine.sweep(src, dst, plan.local_dims, step, lo, hi, rows=(r0, r1),
                   final_only=final_only)
    out = (plan.cuts[s][plan.rank], plan.cuts[s][plan.rank + 1])
    pieces = plan.pieces(s) if schedule.overlapped and plan.world > 1 else None
    if pieces:
      bands, interior = pieces
      for band in bands:
        piece(band, True)
      schedule.after_bands()
      schedule.exchange(lambda dst=dst, s=s: exchange_rows(dst, plan, s + 1, dist))
      exchanges += 1
      pending = True
      piece(interior, True)
    elif out[1] > out[0]:
      piece(out, False)
    src = dst
  schedule.before_super_step()
  return src, exchanges


def auto_exchange(own_rows, reach, deepest, iterate):
  """Iterations between exchanges: a multiple of the deepest fused kernel, up to
  eight of them, but never more ghost rows than ~15 % of the slab.

  Measured on one MI355X at the slab sizes of 1/2/4/8 ranks of the 16384^2
  jacobi2d grid (tools/slab_cost.py: own rows + 2 E ghost rows, E iterations,
  depths up to 24): the cost per iteration hardly depends on E (8 ranks: 5.24 /
  5.41 / 5.42 / 5.56 us at E = 48 / 96 / 144 / 192; 4 ranks 9.11 / 9.73 / 9.65 /
  9.60), so E is chosen for the exchange: the ghost BYTES per iteration are fixed,
  the number of (latency-bound) exchanges is not."""
  e = deepest * 8
  while e > deepest and e * reach * 2 > max(1, own_rows) * 0.15:
    e -= deepest
  return max(1, min(e, iterate))


def exchange_candidates(smallest_own, reach, deepest, iterate):
  """Exchange periods worth timing: 1, 2, 4 and 8 times the deepest fused kernel,
  none with ghost regions deeper than the thinnest slab (SlabPlan's clamp) and none
  beyond the iteration count; duplicates dropped, ascending."""
  cap = max(1, min(iterate, smallest_own // max(1, reach)))
  out = []
  for multiple in (1, 2, 4, 8):
    e = max(1, min(deepest * multiple, cap))
    if e not in out:
      out.append(e)
  return out


EXCHANGE_REPEATS = 3       # timed steps per candidate
EXCHANGE_MARGIN = 0.02     # a candidate must beat the incumbent by this much


def choose_exchange(candidates, time_step, reduce_max, repeats=EXCHANGE_REPEATS,
                    incumbent=None, margin=EXCHANGE_MARGIN):
  """Times `repeats` steps of every (exchange period, overlapped) pair and returns
  (table, chosen): table = [dict(exchange=E, overlapped=bool, ms=the slowest rank's
  FASTEST step, repeats=n)], chosen = its fastest row (the first of equals) - unless
  `incumbent` (a pair of the table: the default period, serial order) is within `margin`
  of it: a single step on a shared node varies by more than the few percent that separate
  neighbouring candidates, and the default must not be traded for noise.

  `time_step(E, overlapped, repeats)` -> [seconds] on THIS rank; `reduce_max(list of
  seconds)` -> the element-wise maximum over all ranks.  Every rank sees the same table
  and therefore takes the same pair - a rank that chose another exchange period would
  wait for messages nobody sends."""
  local = []
  for e, overlapped in candidates:
    samples = list(time_step(e, overlapped, repeats))
    local.append(min(samples))
  slowest = reduce_max(local)
  table = [dict(exchange=e, overlapped=bool(o), ms=float(t) * 1e3, repeats=repeats)
           for (e, o), t in zip(candidates, slowest)]
  chosen = min(table, key=lambda row: row['ms'])
  if incumbent is not None:
    held = [row for row in table
            if (row['exchange'], row['overlapped']) == (incumbent[0], bool(incumbent[1]))]
    if held and chosen['ms'] >= held[0]['ms'] * (1.0 - margin):
      chosen = held[0]
  return table, chosen


# ---------------------------------------------------------------------------
# the HIP engine + bench driver (GPU only)
# ---------------------------------------------------------------------------
class HipEngine:
  """Runs the sweep of a `host.Program` on torch CUDA tensors, on torch's
  current stream so that it is ordered with RCCL traffic."""

  def __init__(self, program, torch):
    self.program, self.torch = program, torch
    self.final_only = False

  def sweep(self, src, dst, local_dims, iterations, valid_lo, valid_hi, rows=None,
            final_only=False):
    """rows = (r0, r1): sweep only the sub-array of those rows of the outermost
    dimension (a contiguous piece of memory: the same call on offset pointers).
    final_only: dst is written by the sweep's last launch only
    (soda_hip_plan_set_out_final_only)."""
    if final_only != self.final_only:
      self.program.set_out_final_only(final_only)
      self.final_only = final_only
    stream = self.torch.cuda.current_stream().cuda_stream
    dims = list(local_dims)
    sp, dp = src.data_ptr(), dst.data_ptr()
    if rows is not None:
      row_bytes = src.element_size()
      for n in dims[:-1]:
        row_bytes *= n
      sp, dp = sp + rows[0] * row_bytes, dp + rows[0] * row_bytes
      dims[-1] = rows[1] - rows[0]
    self.program.sweep([sp], [dp], dims, iterations, valid_lo, valid_hi,
                       stream=stream)


class StreamSchedule:
  """Exchange on a side stream: it starts when the band sweeps have finished
  (event) and the next super-step waits for it (event), so it runs beside the
  interior sweep.  RCCL work issued under torch.cuda.stream(side) is ordered
  after that stream's earlier work, and req.wait() makes the side stream (not
  the host) wait for it."""
  overlapped = True
  skip_exchange = False      # see SerialSchedule

  def __init__(self, torch, host_sync=False):
    """host_sync: the backend reads device memory from the host side without
    ordering itself against streams (gloo in the one-GPU rehearsal): wait for
    the side stream before handing it the arrays."""
    self.torch = torch
    self.host_sync = host_sync
    self.side = torch.cuda.Stream()
    self.bands_done = torch.cuda.Event()
    self.ghosts_landed = None
    self.spans = []                     # (start, end) events of every exchange

  def after_bands(self):
    self.bands_done.record(self.torch.cuda.current_stream())
    self.side.wait_event(self.bands_done)

  def exchange(self, fn):
    if self.skip_exchange:
      return
    main = self.torch.cuda.current_stream()
    # the rows to be sent were produced on the main stream (the input, the band
    # sweeps, or - small slabs - a whole-slab sweep): the side stream follows
    # everything enqueued there so far.  (after_bands() alone covered only the
    # banded super-steps.)
    self.side.wait_stream(main)
    t0 = self.torch.cuda.Event(enable_timing=True)
    t1 = self.torch.cuda.Event(enable_timing=True)
    with self.torch.cuda.stream(self.side):
      t0.record()
      if self.host_sync:
        self.side.synchronize()
      fn()
      t1.record()
    self.spans.append((t0, t1))
    self.ghosts_landed = t1

  def before_super_step(self):
    if self.ghosts_landed is not None:
      self.torch.cuda.current_stream().wait_event(self.ghosts_landed)
      self.ghosts_landed = None

  def exchange_ms(self):
    total = sum(a.elapsed_time(b) for a, b in self.spans)
    self.spans = []
    return total


class TimedSerialSchedule(SerialSchedule):
  """The serial order on a GPU, with the exchanges bracketed by events."""

  def __init__(self, torch, host_sync=False):
    self.torch = torch
    self.host_sync = host_sync
    self.spans = []

  def exchange(self, fn):
    if self.skip_exchange:
      return
    t0 = self.torch.cuda.Event(enable_timing=True)
    t1 = self.torch.cuda.Event(enable_timing=True)
    t0.record()
    if self.host_sync:
      self.torch.cuda.current_stream().synchronize()
    fn()
    t1.record()
    self.spans.append((t0, t1))

  exchange_ms = StreamSchedule.exchange_ms


def make_plan(static, dims, rank, world, r_lo, r_hi, exchange, iterate):
  """The static even cut (SlabPlan) or slabs re-cut every super-step (RecutPlan)."""
  if static:
    return SlabPlan(dims, rank, world, r_lo, r_hi, exchange)
  return RecutPlan(dims, rank, world, r_lo, r_hi, exchange, iterate)


def run_plan(engine, plan, arrays, iterate, margins_of, dist, schedule=None,
             ghosts_ready=False):
  if isinstance(plan, RecutPlan):
    return run_recut(engine, plan, arrays, margins_of, dist, ghosts_ready=ghosts_ready,
                     schedule=schedule)
  return run_slab(engine, plan, arrays, iterate, margins_of, dist,
                  ghosts_ready=ghosts_ready, schedule=schedule)


def result_rows(plan):
  """((first, last) global rows of the result this rank holds, their local offset)."""
  if isinstance(plan, RecutPlan):
    return plan.final_rows, plan.final_rows[0] - plan.base
  return (plan.start, plan.stop), plan.ghost_lo


def super_step_shapes(plan, iterate):
  """[(first local row, rows, iterations)] of the whole-slab sweep of every super-step."""
  if isinstance(plan, RecutPlan):
    out = []
    for (done, step), need in zip(plan.steps, plan.need):
      if need[plan.rank]:
        a, b = plan.local(need[plan.rank])
        out.append((a, b - a, step))
    return out
  out, done = [], 0
  while done < iterate:
    step = min(plan.exchange, iterate - done)
    out.append((0, plan.local_extent, step))
    done += step
  return out


def check_dims(dims, world, reach, exchange, iterate):
  """The small grid of the multi-rank self-check and its iteration count: the inner
  extents capped (4096 columns; 256 x 256 planes), two and a half exchange periods of
  iterations (or the whole run if shorter), enough rows that every rank still holds
  valid rows at the end."""
  it = max(1, min(iterate, 2 * exchange + max(1, exchange // 2)))
  cap = 4096 if len(dims) == 2 else 256
  inner = [min(int(n), cap) for n in dims[:-1]]
  it = min([it] + [(n - 8) // (2 * reach) for n in inner if n > 8 + 2 * reach])
  it = max(1, it)
  rows = 2 * reach * it + world * (64 if len(dims) == 2 else 16)
  return inner + [rows], it


def multi_rank_check(torch, dist, program, spec, make_input, margin_table_of, rank, world,
                     static, r_lo, r_hi, exchange, overlapped, dims, iterate, backend, dev):
  """Runs the chosen (cut, exchange period, order) once on a small grid cut the same
  way, gathers every rank's rows of the result on rank 0 and compares them, cell for
  cell, with the same grid swept by rank 0 alone (HIP against HIP: the one-rank sweep is
  what the parity suite checks against the oracle).  Returns the number of differing
  cells of the valid box, the same on every rank.  The driver's scaling run is the only
  time real RCCL with more than one rank executes this code; a ghost row that arrives
  wrong must not yield a throughput figure."""
  reach = max(r_lo, r_hi, 1)
  cdims, cit = check_dims(dims, world, reach, exchange, iterate)
  plan = make_plan(static, cdims, rank, world, r_lo, r_hi, exchange, cit)
  engine = HipEngine(program, torch)
  table = margin_table_of(cit)

  def margins_of(k):
    if k == 0:
      return (0,) * spec['dim'], (0,) * spec['dim']
    return table[k - 1]
  own = torch.from_numpy(make_input(spec, cdims, rows=(plan.start, plan.stop))[0]).to(dev)
  shape = tuple(reversed(plan.local_dims))
  a, b, c = (torch.zeros(shape, dtype=own.dtype, device=dev) for _ in range(3))
  a[plan.ghost_lo:plan.ghost_lo + plan.own].copy_(own)
  order = (StreamSchedule if overlapped and world > 1 else TimedSerialSchedule)(
      torch, host_sync=backend != 'nccl')
  result, _ = run_plan(engine, plan, [a, b, c], cit, margins_of, dist, schedule=order)
  # (the engine leaves final_only as the last piece set it)
  program.set_out_final_only(False)
  torch.cuda.synchronize()
  (first, last), offset = result_rows(plan)
  mine = result[offset:offset + (last - first)].contiguous()
  signed = {'torch.uint16': torch.int16, 'torch.uint32': torch.int32}
  wire = (lambda t: t.view(signed[str(t.dtype)]) if str(t.dtype) in signed else t)
  differing = 0
  if rank == 0:
    whole_in = torch.from_numpy(make_input(spec, cdims)[0]).to(dev)
    want = torch.zeros_like(whole_in)
    program.sweep([whole_in.data_ptr()], [want.data_ptr()], cdims, cit,
                  stream=torch.cuda.current_stream().cuda_stream)
    got = torch.zeros_like(whole_in)
    got[first:last].copy_(mine)
    ops = []
    for q in range(1, world):
      (qa, qb), _ = result_rows(make_plan(static, cdims, q, world, r_lo, r_hi, exchange, cit))
      if qb > qa:
        ops.append(dist.P2POp(dist.irecv, wire(got)[qa:qb], q))
    if ops:
      if backend != 'nccl':
        torch.cuda.synchronize()
      for req in dist.batch_isend_irecv(ops):
        req.wait()
    torch.cuda.synchronize()
    lo, hi = table[cit - 1]
    box = tuple(slice(lo[d], cdims[d] - hi[d]) for d in reversed(range(spec['dim'])))
    differing = int((got[box] != want[box]).sum().item())
    if got[box].numel() == 0:
      differing = -1
  elif last > first:
    if backend != 'nccl':
      torch.cuda.synchronize()
    for req in dist.batch_isend_irecv([dist.P2POp(dist.isend, wire(mine), 0)]):
      req.wait()
    torch.cuda.synchronize()
  verdict = torch.tensor([differing], dtype=torch.int64,
                         device=dev if backend == 'nccl' else 'cpu')
  dist.broadcast(verdict, 0)
  return int(verdict.item()), cdims, cit


def bench_main(args, open_program, make_input, per_iteration_updates,
               roofline_block, schedule_text, cpu_baseline=None):
  """`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`."""
  import torch
  import torch.distributed as dist
  from . import capi, host
  from ..codegen import spec as specmod

  rank = int(os.environ.get('RANK', '0'))
  world = int(os.environ.get('WORLD_SIZE', '1'))
  local_rank = int(os.environ.get('LOCAL_RANK', str(rank)))
  # SODA_DIST_BACKEND=gloo: rehearsal of the multi-rank path where the ranks
  # share ONE GPU (RCCL refuses two ranks on a device); ghost rows then travel
  # through the host.  Production is nccl (= RCCL over xGMI), one GPU per rank.
  backend = os.environ.get('SODA_DIST_BACKEND', 'nccl')
  if backend != 'nccl':
    local_rank = local_rank % max(1, torch.cuda.device_count())
  torch.cuda.set_device(local_rank)
  capi.check(capi.lib().soda_hip_set_device(local_rank))
  # a collective that never completes (a rank that died, a pairing that deadlocks) fails
  # after this long instead of torch's ten minutes: nothing here waits longer than the CPU
  # baseline rank 0 times while the others stand at a barrier (~12 s)
  import datetime
  patience = datetime.timedelta(seconds=int(os.environ.get('SODA_DIST_TIMEOUT_S', '240')))
  if backend == 'nccl':
    dist.init_process_group(backend='nccl', device_id=torch.device('cuda', local_rank),
                            timeout=patience)
  else:
    dist.init_process_group(backend=backend, timeout=patience)
  try:
    program, spec = open_program(args.app, args.iterate, args.jit)
    program.set_max_depth(args.max_depth)
    if len(spec['inputs']) != 1 or len(spec['outputs']) != 1:
      raise SystemExit('multi-GPU bench handles one-input one-output programs')
    dims = list(args.size)
    r_lo, r_hi = spec['radius']['lo'][-1], spec['radius']['hi'][-1]
    deepest = max([k['depth'] for k in program.kernels if k['kind'] == 'fused'
                   and (args.max_depth <= 0 or k['depth'] <= args.max_depth)] or [1])
    bounds = slab_bounds(dims[-1], world)
    start, stop = bounds[rank]
    smallest = min(b1 - b0 for b0, b1 in bounds)
    reach = max(r_lo, r_hi, 1)
    # slabs re-cut to the shrinking valid box every super-step unless --static-cut; with
    # neither --static-cut nor --recut the chosen (period, order) is timed under the static
    # cut too and the faster cut runs: the re-cut levels the ranks' work but ships the rows
    # that change owner on top of the ghost rows (cfg4 on 8 ranks: up to 252 rows per
    # message instead of 144), and which of the two weighs more is the links' to say
    static = bool(getattr(args, 'static_cut', False))
    cut_given = static or bool(getattr(args, 'recut', False))
    dt = program.in_dtypes[0]
    tdt = {'float32': torch.float32, 'float64': torch.float64,
           'uint16': torch.uint16, 'int16': torch.int16, 'uint8': torch.uint8,
           'int32': torch.int32}[dt.name]
    dev = torch.device('cuda', local_rank)
    given = bool(args.exchange) or bool(getattr(args, 'overlap', False)) or \
        bool(getattr(args, 'no_exchange_tune', False)) or world == 1
    # Which (exchange period, order) pairs to time: unless given on the command line,
    # E = 1, 2, 4, 8 x the deepest fused kernel, each serial and overlapped - the two
    # knobs that decide an N > 1 run and that one GPU cannot measure
    if given:
      pairs = [(args.exchange or auto_exchange(dims[-1] // world, reach, deepest,
                                               args.iterate),
                bool(getattr(args, 'overlap', False)) and world > 1)]
    else:
      pairs = [(e, o) for e in exchange_candidates(smallest, reach, deepest, args.iterate)
               for o in (False, True)]
    # this rank's rows of the global seeded input stay on the device; every candidate
    # builds its slab (own rows + ITS ghost rows) as a view of three arrays sized for
    # the deepest ghost regions
    own_rows = torch.from_numpy(make_input(spec, dims, rows=(start, stop))[0]).to(dev)
    extents = [make_plan(cut, dims, rank, world, r_lo, r_hi, e, args.iterate).local_extent
               for e in sorted({e for e, _ in pairs})
               for cut in ((static,) if cut_given else (False, True))]
    full_shape = tuple(reversed(dims[:-1] + [max(extents)]))
    storage = [torch.zeros(full_shape, dtype=tdt, device=dev) for _ in range(3)]
    engine = HipEngine(program, torch)
    margin_table = specmod.iteration_margins(spec, args.iterate)

    def margins_of(k):
      if k == 0:
        return (0,) * spec['dim'], (0,) * spec['dim']
      return margin_table[k - 1]

    def setup(exchange, overlapped, static_cut=None):
      """(plan, [a, b, c], order, step) of one candidate: a holds the own rows at
      level 0, ghost rows anything - every step exchanges them first."""
      plan = make_plan(static if static_cut is None else static_cut, dims, rank, world,
                       r_lo, r_hi, exchange, args.iterate)
      a, b, c = (t[:plan.local_extent] for t in storage)
      a[plan.ghost_lo:plan.ghost_lo + plan.own].copy_(own_rows)
      order = (StreamSchedule if overlapped else TimedSerialSchedule)(
          torch, host_sync=backend != 'nccl')

      # Every exchange is inside the timed region, the level-0 one included: a step
      # starts from own rows only, as a fresh input would arrive.
      def step():
        return run_plan(engine, plan, [a, b, c], args.iterate, margins_of, dist,
                        schedule=order)
      return plan, (a, b, c), order, step

    def tune_split(plan, arrays):
      # untimed: the candidate splits of a super-step on this rank's slab
      # (soda_hip_plan_tune; no communication inside)
      # (a re-cut run sweeps a slightly smaller sub-array every super-step: each
      # distinct shape is tuned - the tuned split is keyed by extents and iterations)
      if not getattr(args, 'no_tune', False) and args.iterate > 1:
        torch.cuda.synchronize()
        row_bytes = arrays[0][0].numel() * arrays[0].element_size()
        for first_row, rows, iterations in sorted(set(super_step_shapes(plan, args.iterate))):
          program.tune([arrays[0].data_ptr() + first_row * row_bytes],
                       [arrays[1].data_ptr() + first_row * row_bytes],
                       plan.local_dims[:-1] + [rows], iterations,
                       stream=torch.cuda.current_stream().cuda_stream)

    def reduce_max(seconds):
      t = torch.tensor(list(seconds), dtype=torch.float64,
                       device=dev if backend == 'nccl' else 'cpu')
      dist.all_reduce(t, op=dist.ReduceOp.MAX)
      return [float(v) for v in t.tolist()]

    def fenced(fn, repeats):
      """Seconds `repeats` calls of fn take on this rank, between two barriers."""
      torch.cuda.synchronize()
      dist.barrier()
      torch.cuda.synchronize()
      t0, out = time.perf_counter(), None
      for _ in range(repeats):
        out = fn()
      torch.cuda.synchronize()
      dist.barrier()
      torch.cuda.synchronize()
      return time.perf_counter() - t0, out

    # a true collective first: batched point-to-point calls may involve a subset of
    # the ranks only AFTER the group's first collective (torch.distributed docs)
    dist.barrier()
    # one untimed exchange in any case: RCCL builds its point-to-point channels
    # on first use (seconds), which must not land in a run started with --warmup 0
    plan, arrays, order, step = setup(*pairs[0])
    if isinstance(plan, RecutPlan):
      exchange_rows(arrays[0], plan, 0, dist)
    else:
      exchange_ghosts(arrays[0], plan, dist)
    table = None
    if len(pairs) > 1:
      tuned = set()

      def time_step(exchange, overlapped, repeats, static_cut=None):
        plan, arrays, order, step = setup(exchange, overlapped, static_cut)
        if (exchange, static_cut) not in tuned:
          tune_split(plan, arrays)
          tuned.add((exchange, static_cut))
        step()                                   # untimed: clocks, channels, scratch
        return [fenced(step, 1)[0] for _ in range(repeats)]
      # the incumbent: the default period (clamped like the candidates), serial order
      default = auto_exchange(dims[-1] // world, reach, deepest, args.iterate)
      near = min((e for e, _ in pairs), key=lambda e: abs(e - default))
      table, chosen = choose_exchange(pairs, time_step, reduce_max,
                                      incumbent=(near, False))
      for row in table:
        row['cut'] = 'static' if static else 'recut'
      if not cut_given:
        # the chosen pair under the static cut: kept only when it wins by the margin
        other, _ = choose_exchange(
            [(chosen['exchange'], chosen['overlapped'])],
            lambda e, o, repeats: time_step(e, o, repeats, static_cut=True), reduce_max)
        other[0]['cut'] = 'static'
        table.append(other[0])
        if other[0]['ms'] < chosen['ms'] * (1.0 - EXCHANGE_MARGIN):
          static, chosen = True, other[0]
      plan, arrays, order, step = setup(chosen['exchange'], chosen['overlapped'])
    overlap = order.overlapped
    a, b, c = arrays
    # the chosen cut, period and order once on a small grid against rank 0 alone: a run
    # whose rows arrive wrong prints no throughput (config.multi_rank_check).  When the
    # chosen configuration fails, the most conservative one - static cut, serial order, the
    # path rounds 1-5 rehearsed - is checked in its place, and only if THAT is bit-exact does
    # the run go on, with it, and say so: the scaling run is the first time real RCCL moves
    # these rows, and one order's or one cut's fault should cost its gain, not the curve.
    check_note = None
    while True:
      torch.cuda.synchronize()
      differing, check_grid, check_iterate = multi_rank_check(
          torch, dist, program, spec, make_input,
          lambda n: specmod.iteration_margins(spec, n), rank, world, static, r_lo, r_hi,
          plan.exchange, overlap, dims, args.iterate, backend, dev)
      if differing == 0:
        break
      failed = '%s cut, %s order, exchange every %d: %d cells differ' % (
          'static' if static else 're-cut', 'bands-first' if overlap else 'serial',
          plan.exchange, differing)
      if rank == 0:
        import sys
        sys.stderr.write('soda_hip bench: multi-rank self-check FAILED (%s)\n' % failed)
      # (no second chance when the configuration already is the conservative one, when it
      # has been tried, or when period and order were given on the command line)
      if (static and not overlap) or check_note is not None or given:
        if rank == 0:
          import json
          print(json.dumps(dict(metric='gcell_updates_per_s', value=None, n_gpus=world,
                                error='multi-rank self-check failed',
                                config=dict(multi_rank_check='%d cells differ' % differing,
                                            multi_rank_check_failed=failed,
                                            multi_rank_check_grid=check_grid,
                                            multi_rank_check_iterate=check_iterate,
                                            exchange_every=plan.exchange,
                                            exchange_overlapped=overlap,
                                            slab_cut='static' if static else 'recut'))),
                flush=True)
        raise SystemExit(3)
      check_note = failed
      static = True
      plan, arrays, order, step = setup(plan.exchange, False, static_cut=True)
      overlap = order.overlapped
      a, b, c = arrays
    for _ in range(args.warmup):
      step()
    if table is None:
      tune_split(plan, arrays)
      step()
    torch.cuda.synchronize()
    order.exchange_ms()
    seconds, (_, exchanges) = fenced(step, args.steps)
    wall = reduce_max([seconds])[0]
    exchange_ms = order.exchange_ms() / max(1, args.steps)
    # the same steps with the exchanges left out (stale ghost rows: launch times do not
    # depend on the data): separates what the slabs' redundant rows and pipeline fill
    # cost from what the exchange over xGMI costs
    order.skip_exchange = True
    step()
    seconds, _ = fenced(step, args.steps)
    compute_only = reduce_max([seconds])[0] / max(1, args.steps)
    order.skip_exchange = False
    result = None
    if rank == 0:
      valid = specmod.valid_cells(spec, dims, args.iterate)
      nominal = int(np.prod(dims)) * args.iterate
      per_step = wall / args.steps
      abytes = specmod.algorithmic_bytes_per_update(spec)
      # dominant kernel of THIS rank: its first super-step timed per launch on
      # this stream, priced on the valid updates of its OWN rows (the same
      # definition as at N = 1: ghost rows are redundant work and earn nothing)
      first = min(plan.exchange, args.iterate)
      first_row, rows, _ = super_step_shapes(plan, args.iterate)[0]
      row_bytes = a[0].numel() * a.element_size()
      first_dims = plan.local_dims[:-1] + [rows]
      timing = program.sweep_timed([a.data_ptr() + first_row * row_bytes],
                                   [b.data_ptr() + first_row * row_bytes], first_dims,
                                   first, warmup=1, repeats=3)
      sched = program.schedule(first_dims, first)
      out_rows = (plan.cuts[0][rank], plan.cuts[0][rank + 1]) \
          if isinstance(plan, RecutPlan) else (plan.start, plan.stop)
      updates = per_iteration_updates(spec, dims, first, rows=out_rows)
      result = dict(
          metric='gcell_updates_per_s', value=valid / per_step / 1e9,
          unit='Gcell-updates/s', n_gpus=world, steps=args.steps,
          warmup=args.warmup, ms_per_step=per_step * 1e3, higher_is_better=True,
          scaling='strong', vs_baseline=None,
          dtype='f32' if dt.kind == 'f' else 'u%d' % (8 * dt.itemsize),
          data='synthetic',
          config=dict(workload='%s.soda %s %s, iterate %d' % (
              args.app, dt.name, 'x'.join(map(str, dims)), args.iterate),
                      app=args.app, dims=dims, iterate=args.iterate,
                      parallelism='outer-dim slabs x%d' % world,
                      multi_rank_check='bit-exact' if check_note is None else
                      'bit-exact with the static cut and the serial order, taken because '
                      'the chosen configuration FAILED (%s)' % check_note,
                      multi_rank_check_grid='x'.join(map(str, check_grid)),
                      multi_rank_check_iterate=check_iterate,
                      slab_cut='static' if static else 'recut every super-step',
                      exchange_every=plan.exchange, exchanges_per_step=exchanges,
                      exchange_ms_per_step=exchange_ms,
                      exchange_overlapped=overlap,
                      exchange_choice='given' if table is None else
                      'measured (fastest of exchange_candidates_ms: %d steps each, the '
                      'default pair kept within %.0f %%; its last row = the chosen pair '
                      'under the static cut)' % (EXCHANGE_REPEATS, EXCHANGE_MARGIN * 100),
                      exchange_candidates_ms=table or [],
                      compute_only_ms_per_step=compute_only * 1e3,
                      ghost_rows=[plan.exchange * r_lo, plan.exchange * r_hi],
                      local_rows=plan.local_extent,
                      super_step_schedule=schedule_text(sched),
                      valid_cell_updates=valid, nominal_cell_updates=nominal,
                      nominal_gcell_updates_per_s=nominal / per_step / 1e9,
                      effective_GBps=valid * abytes / per_step / 1e9,
                      device=host.device_info(local_rank)['arch']),
          roofline=roofline_block(spec, program, sched, updates, timing,
                                  first_dims, first))
      if cpu_baseline is not None and getattr(args, 'cpu_seconds', 0) > 0 and (
          world == 1 or getattr(args, 'cpu_baseline_at_all_n', False)):
        # the CPU figure is a property of the host, reported on the N = 1 line; with
        # --cpu-baseline-at-all-n also beside the other points of the curve: the same
        # whole-grid workload on this host's cores, timed by rank 0 while the other
        # ranks wait at the barrier below - outside the timed region
        result['cpu_baseline'] = cpu_baseline(spec, dims, args.cpu_seconds)
      result['roofline']['note'] = (
          'rank 0, first super-step of %d iterations on %d rows (%d of them its own '
          'output rows, on which the updates are counted)' % (
              first, rows, out_rows[1] - out_rows[0]))
    # rank 0 has timed its dominant kernel meanwhile: leave together
    torch.cuda.synchronize()
    dist.barrier()
    program.close()
    return result
  finally:
    dist.destroy_process_group()
