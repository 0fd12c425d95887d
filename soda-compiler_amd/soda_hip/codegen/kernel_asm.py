"""Hand-ordered arithmetic for plain float programs: the cells of a stage instance
as ONE interleaved stream of VALU instructions instead of C++ statements.

Why: the stage value is `T(expr)` evaluated cell by cell (reference
src/soda/codegen/xilinx/hls_kernel.py:487-489); written as C++ the compiler's
scheduler, short of registers in the deep 3-D kernels, computes the cells one after
the other on a single accumulator, every instruction waiting for the one before it.
A wavefront then issues one VALU instruction per ~7.5 cycles instead of one per ~4.3
(tools/depbench.hip: a dependent chain of v_add_f32 against four independent ones),
and at two wavefronts per SIMD that is what the arithmetic of a plane costs
(tools/blk_stamps.py: ~780 cycles per 112-instruction level).

Here the expression becomes a short list of two-operand operations (the same IEEE
operations in the same association as the C++ text: left to right, parentheses kept,
no contraction; repeated sub-expressions computed once, as the compiler's CSE does),
and a GROUP of cells walks through that list in lock step: operation k of every cell
of the group, then operation k + 1, ... - one `asm volatile` per instruction, which
keeps exactly this order, with the compiler still allocating the registers.  Operands
that come from a neighbouring lane are fetched first, by the usual DPP helpers (the
compiler knows their hazards; inside an `asm` it would not).

Scope: `kernel_stream2d_wp.packable` programs without division (a float division is
not one instruction): all tensors float, literals float-suffixed or integer, + - * and
unary minus.  Everything else keeps the C++ form.
"""
import re
import struct

import numpy as np

from . import spec as specmod

_TOKEN = re.compile(r'\s*(?:(\{[^}]*\})|((?:\d+\.?\d*|\.\d+)(?:[eE][+-]?\d+)?)[fF]?|([-+*/()]))')


class Unsupported(Exception):
  pass


def tokenize(text):
  out, pos = [], 0
  text = text.strip()
  while pos < len(text):
    m = _TOKEN.match(text, pos)
    if not m:
      raise Unsupported('cannot read %r' % text[pos:pos + 20])
    if m.group(1):
      out.append(('load', m.group(1)))
    elif m.group(2):
      out.append(('num', m.group(2)))
    else:
      out.append(('op', m.group(3)))
    pos = m.end()
  return out


def parse(text):
  """Expression text of a packable stage -> tree: ('load', name, rel) | ('num', f32)
  | ('neg', x) | (op, a, b) with C precedence and left associativity."""
  toks = tokenize(text)
  pos = [0]

  def peek():
    return toks[pos[0]] if pos[0] < len(toks) else (None, None)

  def take():
    pos[0] += 1
    return toks[pos[0] - 1]

  def primary():
    kind, val = take()
    if kind == 'load':
      m = specmod.LOAD_RE.match(val)
      return ('load', m.group(1), tuple(int(v) for v in m.group(2).split(',')))
    if kind == 'num':
      return ('num', np.float32(val))
    if (kind, val) == ('op', '('):
      e = additive()
      if take() != ('op', ')'):
        raise Unsupported('unbalanced parentheses')
      return e
    if (kind, val) == ('op', '-'):
      return ('neg', primary())
    if (kind, val) == ('op', '+'):
      return primary()
    raise Unsupported('unexpected token %r' % (val,))

  def multiplicative():
    e = primary()
    while peek() in (('op', '*'), ('op', '/')):
      op = take()[1]
      if op == '/':
        raise Unsupported('division is not one instruction')
      e = (op, e, primary())
    return e

  def additive():
    e = multiplicative()
    while peek() in (('op', '+'), ('op', '-')):
      op = take()[1]
      e = (op, e, multiplicative())
    return e

  tree = additive()
  if pos[0] != len(toks):
    raise Unsupported('trailing input')
  return tree


def fold(tree):
  """Constant sub-expressions evaluated in float32, as the compiler folds them."""
  kind = tree[0]
  if kind in ('load', 'num'):
    return tree
  if kind == 'neg':
    x = fold(tree[1])
    return ('num', np.float32(-x[1])) if x[0] == 'num' else ('neg', x)
  a, b = fold(tree[1]), fold(tree[2])
  if a[0] == 'num' and b[0] == 'num':
    with np.errstate(all='ignore'):
      v = {'+': a[1] + b[1], '-': a[1] - b[1], '*': a[1] * b[1]}[kind]
    return ('num', np.float32(v))
  return (kind, a, b)


def lower(tree):
  """-> (ops, result): ops = [(opcode, a, b)] in evaluation order, operands
  ('t', index of an earlier op) | ('load', name, rel) | ('num', f32); identical
  sub-trees share one op."""
  ops, seen = [], {}

  def visit(node):
    if node[0] in ('load', 'num'):
      return node
    key = repr(node)
    if key in seen:
      return seen[key]
    if node[0] == 'neg':
      entry = ('neg', visit(node[1]), None)
    else:
      entry = (node[0], visit(node[1]), visit(node[2]))
    ops.append(entry)
    seen[key] = ('t', len(ops) - 1)
    return seen[key]
  result = visit(fold(tree))
  return ops, result


def literal(value):
  return '0x%08x' % struct.unpack('<I', struct.pack('<f', float(value)))[0]


def supported(stage):
  if stage['lets']:
    return False
  try:
    ops, result = lower(parse(stage['expr']))
  except Unsupported:
    return False
  return bool(ops) and result[0] == 't'


def emit_cells(stage, cells, emit, indent, group=4, prefix='soda_a'):
  """cells = [(target lvalue, load(tensor, rel) -> C expression)].  Emits the cells
  in groups of `group`, each group's operations interleaved."""
  ops, result = lower(parse(stage['expr']))
  last_use = {}
  for k, (_, a, b) in enumerate(ops):
    for operand in (a, b):
      if operand is not None and operand[0] == 't':
        last_use[operand[1]] = k
  for g0 in range(0, len(cells), group):
    batch = cells[g0:g0 + group]
    emit(indent + '{')
    leaf = [{} for _ in batch]

    def leaf_of(i, node):
      """C expression (a register) of a load operand of cell i."""
      key = (node[1], node[2])
      if key not in leaf[i]:
        text = batch[i][1](node[1], node[2])
        if 'from_lane' in text or 'pk_' in text:    # a lane-crossing operand
          name = '%s_x%d_%d' % (prefix, i, len(leaf[i]))
          emit('%s  const float %s = %s;' % (indent, name, text))
          text = name
        leaf[i][key] = text
      return leaf[i][key]
    # lane-crossing operands first (DPP moves, scheduled by the compiler)
    for i in range(len(batch)):
      for _, a, b in ops:
        for operand in (a, b):
          if operand is not None and operand[0] == 'load':
            leaf_of(i, operand)
    n_tmp = len(ops)
    emit('%s  float %s;' % (indent, ', '.join(
        '%s_t%d_%d' % (prefix, i, k) for i in range(len(batch)) for k in range(n_tmp - 1))
        or '%s_unused' % prefix))
    for k, (op, a, b) in enumerate(ops):
      final = k == len(ops) - 1
      for i, (target, _) in enumerate(batch):
        dst = target if final else '%s_t%d_%d' % (prefix, i, k)

        def reg(x, i=i):
          return leaf_of(i, x) if x[0] == 'load' else '%s_t%d_%d' % (prefix, i, x[1])
        if op == 'neg':
          if a[0] == 'num':
            raise Unsupported('constant result')
          emit('%s  asm volatile("v_xor_b32 %%0, 0x80000000, %%1" : "=v"(%s) : "v"(%s));'
               % (indent, dst, reg(a)))
          continue
        if a[0] == 'num' and b[0] == 'num':
          raise Unsupported('unfolded constants')
        if a[0] == 'num' or b[0] == 'num':
          const, var = (a, b) if a[0] == 'num' else (b, a)
          if op == '+':
            text = 'v_add_f32 %%0, %s, %%1' % literal(const[1])
          elif op == '*':
            text = 'v_mul_f32 %%0, %s, %%1' % literal(const[1])
          elif a[0] == 'num':       # constant - x
            text = 'v_sub_f32 %%0, %s, %%1' % literal(const[1])
          else:                     # x - constant
            text = 'v_subrev_f32 %%0, %s, %%1' % literal(const[1])
          emit('%s  asm volatile("%s" : "=v"(%s) : "v"(%s));' % (indent, text, dst,
                                                                   reg(var)))
          continue
        mnemonic = {'+': 'v_add_f32', '-': 'v_sub_f32', '*': 'v_mul_f32'}[op]
        emit('%s  asm volatile("%s %%0, %%1, %%2" : "=v"(%s) : "v"(%s), "v"(%s));'
             % (indent, mnemonic, dst, reg(a), reg(b)))
    emit(indent + '}')
