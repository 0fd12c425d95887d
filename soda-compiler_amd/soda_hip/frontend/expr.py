"""Expression tree of the SODA DSL and its two printers.

The reference keeps every expression as a ten-level singleton chain and then
collapses it (reference src/haoda/ir/__init__.py:29-59 grammar, :157-349 nodes,
src/haoda/ir/arithmetic/base.py:16-93 `flatten`).  Here the parser builds the
collapsed form directly; what has to be *identical* is the text the tree prints
as, because that text is the arithmetic contract:

  * operand order is never changed: the reference's merge of nested
    same-precedence chains (base.py:48-66) looks at children before they are
    collapsed, so their type never equals the parent's and nothing is ever
    re-associated;
  * a compound chain prints as `(` a op b op c `)`, where the wrapper first
    strips EVERY leading `(` / trailing `)` pair of the joined text and then adds
    one pair back (ir:215-222, :871-878).  For a chain whose first operand starts
    with `(` and whose last ends with `)` this drops the chain's own parentheses,
    e.g. `65535 - (a*a + b*b)` prints as `(65535 - (a * a) + (b * b))`.  The
    reference evaluates that text on both its FPGA and its CPU side, so it is
    what "the reference's result" means and it is reproduced here on purpose.
"""

# precedence levels, loosest first (reference ir:29-55)
LEVELS = (
    ('||',),
    ('&&',),
    ('|',),
    ('^',),
    ('&',),
    ('==', '!='),
    ('<=', '>=', '<', '>'),
    ('+', '-'),
    ('*', '/', '%'),
)
UNARY_OPS = ('+', '-', '~', '!')

# reference src/soda/grammar.py:25-32
FUNC_NAMES = frozenset('''
cos sin tan acos asin atan atan2 cosh sinh tanh acosh asinh atanh
exp frexp ldexp log log10 modf exp2 expm1 ilogb log1p log2 logb scalbn scalbln
pow sqrt cbrt hypot erf erfc tgamma lgamma
ceil floor fmod trunc round lround llround rint lrint llrint nearbyint
remainder remquo copysign nan nextafter nexttoward fdim fmax fmin fabs abs fma
min max select'''.split())


class Node:
  __slots__ = ()

  def children(self):
    return ()

  def rebuild(self, children):
    return self

  def __eq__(self, other):
    return type(self) is type(other) and self.key() == other.key()

  def __hash__(self):
    return hash((type(self).__name__, self.key()))

  def __repr__(self):
    return '%s(%s)' % (type(self).__name__, soda_text(self))


class Num(Node):
  """A literal, kept verbatim with its C suffix (`0.2f`, `3`, `0x10u`)."""
  __slots__ = ('text',)

  def __init__(self, text):
    self.text = text

  def key(self):
    return self.text


class Load(Node):
  """`name(i, j, ...)`: element of a tensor at a constant index."""
  __slots__ = ('name', 'idx')

  def __init__(self, name, idx):
    self.name, self.idx = name, tuple(idx)

  def key(self):
    return (self.name, self.idx)


class Var(Node):
  """A `let` variable (or `name[i]` element of a param array)."""
  __slots__ = ('name', 'idx')

  def __init__(self, name, idx=()):
    self.name, self.idx = name, tuple(idx)

  def key(self):
    return (self.name, self.idx)


class Unary(Node):
  __slots__ = ('ops', 'operand')

  def __init__(self, ops, operand):
    self.ops, self.operand = tuple(ops), operand

  def key(self):
    return (self.ops, self.operand)

  def children(self):
    return (self.operand,)

  def rebuild(self, children):
    return Unary(self.ops, children[0])


class Chain(Node):
  """`a op b op c` at one precedence level, two or more operands."""
  __slots__ = ('level', 'operands', 'operators')

  def __init__(self, level, operands, operators):
    assert len(operands) == len(operators) + 1 >= 2
    self.level = level
    self.operands, self.operators = tuple(operands), tuple(operators)

  def key(self):
    return (self.level, self.operands, self.operators)

  def children(self):
    return self.operands

  def rebuild(self, children):
    return Chain(self.level, children, self.operators)


class Cast(Node):
  __slots__ = ('haoda_type', 'expr')

  def __init__(self, haoda_type, expr):
    self.haoda_type, self.expr = haoda_type, expr

  def key(self):
    return (self.haoda_type, self.expr)

  def children(self):
    return (self.expr,)

  def rebuild(self, children):
    return Cast(self.haoda_type, children[0])


class Call(Node):
  __slots__ = ('name', 'args')

  def __init__(self, name, args):
    self.name, self.args = name, tuple(args)

  def key(self):
    return (self.name, self.args)

  def children(self):
    return self.args

  def rebuild(self, children):
    return Call(self.name, children)


def make_unary(ops, operand):
  """Applies the reference's identity-unary rule (base.py:79-88): a prefix made
  only of `+`/`-` with an even number of `-`, or only of an even number of `!`,
  disappears; anything else is kept verbatim."""
  ops = tuple(ops)
  minus, plus, nots = ops.count('-'), ops.count('+'), ops.count('!')
  if minus % 2 == 0 and minus + plus == len(ops):
    return operand
  if nots % 2 == 0 and nots == len(ops):
    return operand
  return Unary(ops, operand)


def transform(node, fn):
  """Bottom-up rewrite: `fn(node)` may return a replacement or None."""
  kids = node.children()
  if kids:
    node = node.rebuild(tuple(transform(k, fn) for k in kids))
  out = fn(node)
  return node if out is None else out


def walk(node):
  """Pre-order, left-to-right (the order loads appear in the source)."""
  yield node
  for k in node.children():
    for n in walk(k):
      yield n


def loads_of(node):
  return [n for n in walk(node) if isinstance(n, Load)]


def shift_loads(node, delta, skip=()):
  """Subtracts `delta` from every load index (reference mutator.shift)."""
  def fn(n):
    if isinstance(n, Load) and n.name not in skip:
      return Load(n.name, tuple(a - b for a, b in zip(n.idx, delta)))
  return transform(node, fn)


def rename_loads(node, rename):
  def fn(n):
    if isinstance(n, Load):
      return Load(rename(n.name), n.idx)
  return transform(node, fn)


# ---------------------------------------------------------------------------
# printers
# ---------------------------------------------------------------------------
def _strip_parens(text):
  while text.startswith('(') and text.endswith(')'):
    text = text[1:-1]
  return text


def _wrap(text):
  return '(%s)' % _strip_parens(text)


def soda_text(node):
  """DSL-syntax text, as the reference's `__str__` methods give it."""
  if isinstance(node, Num):
    return node.text
  if isinstance(node, Load):
    return '%s(%s)' % (node.name, ', '.join(map(str, node.idx)))
  if isinstance(node, Var):
    return node.name + ''.join('[%d]' % i for i in node.idx)
  if isinstance(node, Unary):
    return ''.join(node.ops) + soda_text(node.operand)
  if isinstance(node, Chain):
    text = soda_text(node.operands[0])
    for op, operand in zip(node.operators, node.operands[1:]):
      text += ' %s %s' % (op, soda_text(operand))
    return _wrap(text)
  if isinstance(node, Cast):
    return node.haoda_type + _wrap(soda_text(node.expr))
  if isinstance(node, Call):
    return '%s(%s)' % (node.name, ', '.join(map(soda_text, node.args)))
  raise TypeError(node)


def c_text(node, load_text, c_type_of):
  """C/C++ (and HIP) text.  `load_text(load)` prints a tensor element,
  `c_type_of(haoda_type)` a cast's target type."""
  def go(n):
    if isinstance(n, Num):
      return n.text
    if isinstance(n, Load):
      return load_text(n)
    if isinstance(n, Var):
      return n.name + ''.join('[%d]' % i for i in n.idx)
    if isinstance(n, Unary):
      return ''.join(n.ops) + go(n.operand)
    if isinstance(n, Chain):
      text = go(n.operands[0])
      for op, operand in zip(n.operators, n.operands[1:]):
        text += ' %s %s' % (op, go(operand))
      return _wrap(text)
    if isinstance(n, Cast):
      return 'static_cast<%s >%s' % (c_type_of(n.haoda_type), _wrap(go(n.expr)))
    if isinstance(n, Call):
      return '%s(%s)' % (n.name, ', '.join(go(a) for a in n.args))
    raise TypeError(n)
  return go(node)


# ---------------------------------------------------------------------------
# static type of an expression (reference ir:208-211, :243-245, :291-313,
# :335-339): the type of the FIRST operand wins, literals by their spelling.
# ---------------------------------------------------------------------------
def literal_type(text):
  low = text.lower()
  if 'u' in low:
    return 'uint64' if 'll' in low else 'uint32'
  if 'll' in low:
    return 'int64'
  if 'fl' in low:
    return 'double'
  if 'f' in low or 'e' in low:
    return 'float'
  if '.' in low:
    return 'double'
  return 'int32'


def type_of(node, tensor_types, var_types):
  if isinstance(node, Num):
    return literal_type(node.text)
  if isinstance(node, Load):
    return tensor_types.get(node.name)
  if isinstance(node, Var):
    return var_types.get(node.name)
  if isinstance(node, Unary):
    return type_of(node.operand, tensor_types, var_types)
  if isinstance(node, Chain):
    return type_of(node.operands[0], tensor_types, var_types)
  if isinstance(node, Cast):
    return node.haoda_type
  if isinstance(node, Call):
    pick = node.args[1] if node.name == 'select' else node.args[0]
    return type_of(pick, tensor_types, var_types)
  raise TypeError(node)
