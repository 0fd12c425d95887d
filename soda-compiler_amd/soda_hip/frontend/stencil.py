"""Stencil analysis: what a parsed program computes, stage by stage.

Counterpart of the reference's `soda.core.Stencil` / `Tensor` (reference
src/soda/core.py:20-405) and window helpers (core.py:782-835), minus everything
that exists only to lay out an FPGA (stream delays, reuse FIFOs, the dataflow
module graph: core.py:407-777, src/soda/dataflow.py).

Semantics kept:
  * a stage `S(st) = f(T(ld) ...)` means `S[x] = f(T[x + ld - st] ...)`; only the
    relative offsets matter.  Each stage is normalised so that its smallest load
    index is 0 in every dimension (core.py:372-379);
  * `iterate: N` applies the whole program N times, output j feeding input j;
    the reference clones every stage per iteration and names the clones
    `<name>_iter<k>` (core.py:342-360).  `instances()` reproduces those names so
    that the analysis can be compared with the reference's, but nothing here or
    in the back end scales with N except a small table of valid boxes;
  * a stage instance is defined on the box where all transitive reads of the
    program inputs are in bounds: per dimension `[lo, N - hi)` with `lo = -min`
    and `hi = max` of the composed window (core.py:794-835; host.py:1082-1091).
    Minkowski-summing bounding boxes is exact for boxes, so only boxes are kept.
"""
import collections

from . import expr as ex
from .errors import SemanticError
from .types import c_type, is_float


class Stage:
  """One compute statement after normalisation (one iteration's worth)."""

  def __init__(self, stmt, norm, tensor_types):
    self.kind = stmt.kind                  # 'local' | 'output'
    self.haoda_type = stmt.haoda_type
    self.name = stmt.name
    self.st_idx = tuple(a - b for a, b in zip(stmt.ref.idx, norm))
    self.expr = ex.shift_loads(stmt.expr, norm)
    var_types = {}
    self.lets = []
    for let in stmt.lets:
      shifted = ex.shift_loads(let.expr, norm)
      let_type = let.declared_type or ex.type_of(shifted, tensor_types, var_types)
      var_types[let.name] = let_type
      self.lets.append((let.name, let_type, shifted))
    self.var_types = var_types

  def loads(self):
    """All loads in source order: lets first, then the expression
    (reference soda/visitor.py get_load_tuple via Tensor.visit_loads)."""
    out = []
    for _, _, e in self.lets:
      out.extend(ex.loads_of(e))
    out.extend(ex.loads_of(self.expr))
    return out

  def rel_loads(self):
    """Unique (tensor, offset relative to the store) pairs, source order."""
    seen = collections.OrderedDict()
    for ld in self.loads():
      rel = tuple(a - b for a, b in zip(ld.idx, self.st_idx))
      seen[(ld.name, rel)] = None
    return list(seen)

  def parents(self):
    return list(collections.OrderedDict((ld.name, None) for ld in self.loads()))


class Box:
  """Per-dimension [lo, hi] hull of a set of offsets."""
  __slots__ = ('lo', 'hi')

  def __init__(self, lo, hi):
    self.lo, self.hi = tuple(lo), tuple(hi)

  @classmethod
  def origin(cls, dim):
    return cls((0,) * dim, (0,) * dim)

  def shifted(self, lo_off, hi_off):
    return Box([a + b for a, b in zip(self.lo, lo_off)],
               [a + b for a, b in zip(self.hi, hi_off)])

  def hull(self, other):
    if other is None:
      return self
    return Box([min(a, b) for a, b in zip(self.lo, other.lo)],
               [max(a, b) for a, b in zip(self.hi, other.hi)])

  def __repr__(self):
    return 'Box(%s, %s)' % (self.lo, self.hi)


class Stencil:
  """Analysis of one program.  Constructor keywords are the ones the
  reference driver passes to `core.Stencil` (reference src/sodac:109-123)."""

  def __init__(self, burst_width, iterate, app_name, input_stmts, local_stmts,
               output_stmts, dim, tile_size, unroll_factor, param_stmts=(),
               dram_in=None, dram_out=None):
    if iterate < 1:
      raise SemanticError('cannot iterate %d times' % iterate)
    self.iterate = iterate
    self.burst_width = burst_width
    self.app_name = app_name
    self.tile_size = tuple(tile_size)
    self.unroll_factor = unroll_factor
    self.dim = dim
    self.param_stmts = tuple(param_stmts)
    self.input_stmts = tuple(input_stmts)
    self.local_stmts = tuple(local_stmts)
    self.output_stmts = tuple(output_stmts)
    self._apply_dram(dram_in, self.input_stmts, '^', 'input')
    self._apply_dram(dram_out, self.output_stmts, ',', 'output')

    self.input_names = tuple(s.name for s in self.input_stmts)
    self.local_names = tuple(s.name for s in self.local_stmts)
    self.output_names = tuple(s.name for s in self.output_stmts)
    self.param_names = tuple(s.name for s in self.param_stmts)
    self.input_types = tuple(s.haoda_type for s in self.input_stmts)
    self.local_types = tuple(s.haoda_type for s in self.local_stmts)
    self.output_types = tuple(s.haoda_type for s in self.output_stmts)

    if self.iterate > 1:
      if len(self.input_stmts) != len(self.output_stmts):
        raise SemanticError(
            'number of input tensors must be the same as output if iterate > 1 '
            'times, currently there are %d input(s) but %d output(s)' %
            (len(self.input_stmts), len(self.output_stmts)))
      if self.input_types != self.output_types:
        raise SemanticError(
            'input must have the same type(s) as output if iterate > 1 '
            'times, current input has type [%s] but output has type [%s]' %
            (', '.join(self.input_types), ', '.join(self.output_types)))

    self.tensor_types = collections.OrderedDict()
    for stmt in self.input_stmts + self.local_stmts + self.output_stmts:
      if stmt.name in self.tensor_types:
        raise SemanticError('tensor `%s` is defined more than once' % stmt.name)
      self.tensor_types[stmt.name] = stmt.haoda_type
    if self.param_stmts:
      raise SemanticError(
          '`param` arrays are not supported by the HIP back end (they are '
          'unusable in the reference as well: its printers read attributes the '
          'parser never sets)')

    self.stages = self._build_stages()
    self.order = self._topological_order()

  # ------------------------------------------------------------------ setup
  @staticmethod
  def _apply_dram(spec, stmts, sep, what):
    """`--dram-in/--dram-out` syntax of the reference (core.py:198-226):
    either `B.B...` for every tensor or `name:B.B` pairs.  Banks mean nothing
    on a GPU; the values are validated and stored so that a command line
    written for the reference keeps working."""
    if spec is None:
      return
    if ':' in spec:
      by_name = {s.name: s for s in stmts}
      for item in spec.split(sep):
        name, banks = item.split(':')
        if name not in by_name:
          raise SemanticError('no %s named `%s`' % (what, name))
        by_name[name].dram = tuple(map(int, banks.split('.')))
    else:
      for stmt in stmts:
        stmt.dram = tuple(map(int, spec.split('.')))

  def _build_stages(self):
    stages = collections.OrderedDict()
    for stmt in self.local_stmts + self.output_stmts:
      if len(stmt.ref.idx) != self.dim:
        raise SemanticError('`%s` is indexed with %d indices in a %d-D program'
                            % (stmt.name, len(stmt.ref.idx), self.dim))
      loads = []
      defined = set()
      for let in list(stmt.lets) + [None]:
        expr = stmt.expr if let is None else let.expr
        # a variable is a `let` made earlier in the same statement (the reference
        # dies with a KeyError on any other, core.py:116-120)
        for node in ex.walk(expr):
          if isinstance(node, ex.Var) and node.name not in defined:
            raise SemanticError('`%s` uses undefined variable `%s`'
                                % (stmt.name, node.name))
        if let is not None:
          defined.add(let.name)
          loads.extend(ex.loads_of(let.expr))
      loads.extend(ex.loads_of(stmt.expr))
      if not loads:
        raise SemanticError('`%s` reads no tensor' % stmt.name)
      for ld in loads:
        if ld.name not in self.tensor_types:
          raise SemanticError('`%s` reads undefined tensor `%s`' %
                              (stmt.name, ld.name))
        if len(ld.idx) != self.dim:
          raise SemanticError('`%s` is indexed with %d indices in a %d-D '
                              'program' % (ld.name, len(ld.idx), self.dim))
      norm = tuple(min(ld.idx[d] for ld in loads) for d in range(self.dim))
      stages[stmt.name] = Stage(stmt, norm, self.tensor_types)
    return stages

  def _topological_order(self):
    """Breadth-first from the inputs, a stage entering once all its parents
    have (the order of core.py:407-554 `chronological_tensors`, without the
    stream delays)."""
    children = collections.OrderedDict(
        (n, []) for n in self.tensor_types)
    for stage in self.stages.values():
      for parent in stage.parents():
        children[parent].append(stage.name)
    done = list(self.input_names)
    seen = set(done)
    queue = collections.deque(done)
    while queue:
      for child in children[queue.popleft()]:
        if child in seen:
          continue
        if all(p in seen for p in self.stages[child].parents()):
          seen.add(child)
          done.append(child)
          queue.append(child)
    missing = [n for n in self.stages if n not in seen]
    if missing:
      raise SemanticError('stage(s) %s depend on themselves or on a later '
                          'iteration' % ', '.join(missing))
    return tuple(n for n in done if n in self.stages)

  # ------------------------------------------------------------- queries
  def name_in_iter(self, name, iteration):
    """Name of `name`'s clone in `iteration` (core.py:342-358)."""
    if name in self.input_names:
      return name if iteration == 0 else '%s_iter%d' % (name, iteration)
    if name in self.output_names:
      if iteration < self.iterate - 1:
        return '%s_iter%d' % (
            self.input_names[self.output_names.index(name)], iteration + 1)
      return name
    return name if iteration == 0 else '%s_iter%d' % (name, iteration)

  def stage_window(self, stage):
    """Hull of a stage's load offsets per parent: {parent: Box}."""
    out = collections.OrderedDict()
    for name, rel in stage.rel_loads():
      box = Box(rel, rel)
      out[name] = box.hull(out.get(name))
    return out

  def iteration_boxes(self, iterations=None):
    """Valid-region bookkeeping.  Returns a list, one dict per iteration,
    mapping stage name -> Box of composed offsets back to the ORIGINAL program
    inputs.  The stage is defined on `[-box.lo, N - box.hi)` per dimension."""
    iterations = self.iterate if iterations is None else iterations
    feed = {name: Box.origin(self.dim) for name in self.input_names}
    result = []
    for _ in range(iterations):
      boxes = dict(feed)
      for name in self.order:
        acc = None
        for parent, win in self.stage_window(self.stages[name]).items():
          acc = boxes[parent].shifted(win.lo, win.hi).hull(acc)
        # a cell also has to lie inside the array itself: the box always
        # contains the origin.  (The reference lets the loop start at a negative
        # index for a window that lies entirely on one side of the store point,
        # host.py:1082-1091 with core.py:832-835 -- undefined behaviour there.)
        boxes[name] = acc.hull(Box.origin(self.dim))
      result.append({n: boxes[n] for n in self.order})
      if len(self.input_names) == len(self.output_names):
        feed = {i: boxes[o]
                for i, o in zip(self.input_names, self.output_names)}
    return result

  def overall_windows(self, iterations=None, inputs=None):
    """{output: sorted tuple of offsets} - the POINT SET of program-input cells
    an output cell depends on after `iterations` iterations, relative to the
    store (reference core.py:794-830 get_overall_stencil_window over all
    inputs).  The bounding box of it is what iteration_boxes tracks; the points
    themselves decide STENCIL_DISTANCE, the delay of the reference's tiled
    output layout (core.py:782-785)."""
    iterations = self.iterate if iterations is None else iterations
    origin = (0,) * self.dim
    # `inputs`: only dependences on these program inputs (the reference sizes its
    # copy-back loops by the FIRST input alone, host.py:832-839)
    feed = {name: {origin} if inputs is None or name in inputs else set()
            for name in self.input_names}
    points = {}
    for _ in range(iterations):
      points = dict(feed)
      for name in self.order:
        acc = set()
        for parent, rel in self.stages[name].rel_loads():
          acc |= {tuple(a + b for a, b in zip(p, rel)) for p in points[parent]}
        points[name] = acc
      if len(self.input_names) == len(self.output_names):
        feed = {i: points[o] for i, o in zip(self.input_names, self.output_names)}
    return {o: tuple(sorted(points[o])) for o in self.output_names}

  def valid_margins(self, iterations=None):
    """[(lo, hi)] per iteration for the program OUTPUTS (hull over outputs):
    after k+1 iterations the outputs are defined on `[lo_d, N_d - hi_d)`."""
    out = []
    for boxes in self.iteration_boxes(iterations):
      acc = None
      for name in self.output_names:
        acc = boxes[name].hull(acc)
      out.append((tuple(-v for v in acc.lo), tuple(acc.hi)))
    return out

  def radius(self):
    """Growth of the composed window in ONE iteration, as (lo, hi) tuples of
    non-negative ints: how far an output reaches back into the inputs."""
    lo, hi = self.valid_margins(1)[0]
    return lo, hi

  def instances(self):
    """Every stage clone of every iteration in execution order, named like the
    reference names them: dicts with name, stage, iteration, parent renames and
    the loop bounds of the reference's CPU golden loops (host.py:1082-1091)."""
    all_boxes = self.iteration_boxes()
    out = []
    for k, name in self.unrolled_order():
      stage = self.stages[name]
      box = all_boxes[k][name]
      out.append(dict(
          name=self.name_in_iter(name, k), base=name, iteration=k,
          stage=stage,
          rename={p: self.name_in_iter(p, k) for p in stage.parents()},
          loop_lo=tuple(-v for v in box.lo), loop_hi_margin=tuple(box.hi),
          is_output=(name in self.output_names and k == self.iterate - 1)))
    return out

  def unrolled_order(self):
    """(iteration, stage) pairs in the order of the reference's
    `chronological_tensors` (core.py:407-554): ONE breadth-first walk over the graph
    of all iterations' clones, a clone entering when all its parents have.  For a
    single chain this is iteration after iteration in `self.order`; with several
    outputs the walk interleaves differently from the second iteration on (the
    queue carries over).  The run time executes iteration after iteration - any
    topological order computes the same values."""
    def parents_of(k, name):
      out = []
      for p in self.stages[name].parents():
        if p in self.input_names:
          if k == 0:
            out.append(('input', p))
          else:      # the matching output of the previous iteration
            out.append((k - 1, self.output_names[self.input_names.index(p)]))
        else:
          out.append((k, p))
      return out
    nodes = [(k, name) for k in range(self.iterate) for name in self.stages]
    children = collections.OrderedDict((('input', n), []) for n in self.input_names)
    for node in nodes:
      children[node] = []
    parents = {}
    for node in nodes:
      parents[node] = parents_of(*node)
      for p in parents[node]:
        if node not in children[p]:
          children[p].append(node)
    seen = set(('input', n) for n in self.input_names)
    queue = collections.deque(('input', n) for n in self.input_names)
    order = []
    while queue:
      for child in children[queue.popleft()]:
        if child not in seen and all(p in seen for p in parents[child]):
          seen.add(child)
          order.append(child)
          queue.append(child)
    assert len(order) == len(nodes), (order, nodes)
    return order

  def is_float(self):
    return is_float(self.input_types[0])

  def c_type_of(self, name):
    return c_type(self.tensor_types[name])


def stencil_from_program(program, burst_width=None, unroll_factor=None,
                         tile_size=None, iterate=None, dram_in=None,
                         dram_out=None):
  """Applies command-line overrides the way the reference driver does
  (reference src/sodac:94-123) and builds the analysis."""
  tiles = []
  for d in range(program.dim - 1):
    if tile_size is not None and d < len(tile_size) and tile_size[d] > 0:
      tiles.append(tile_size[d])
    else:
      tiles.append(program.tile_size[d])
  tiles.append(0)
  return Stencil(
      burst_width=program.burst_width if burst_width is None else burst_width,
      iterate=program.iterate if iterate is None else iterate,
      app_name=program.app_name,
      input_stmts=program.input_stmts, param_stmts=program.param_stmts,
      local_stmts=program.local_stmts, output_stmts=program.output_stmts,
      dim=program.dim, tile_size=tiles,
      unroll_factor=(program.unroll_factor if unroll_factor is None
                     else unroll_factor),
      dram_in=dram_in, dram_out=dram_out)
