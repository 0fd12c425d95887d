"""Hand-written recursive-descent parser for `.soda` text.

Accepts the language of the reference's textX grammar (reference
src/soda/grammar.py:8-43 statements, src/haoda/ir/__init__.py:13-64 expressions)
without depending on textX, which is not a dependency of this project.

Facts of that grammar reproduced here:
  * the four header keys (`kernel`, `burst width`, `unroll factor`, `iterate`)
    are all mandatory, each exactly once, and may appear anywhere between the
    statement groups (the rule is an unordered group, grammar.py:9-19);
  * `input`+, `param`*, `local`*, `output`+ statements: each kind forms ONE
    contiguous run, runs in any order;
  * `#` starts a comment that runs to the end of the line (grammar.py:23);
  * numbers keep their C spelling as text (`0.2f`, `3`, `0x1Fu`); tensor indices
    and tile sizes are decimal integers with an optional sign;
  * operand alternatives are tried in the order cast, call, tensor reference,
    number, variable, parenthesised expression (ir:57), so a name from the math
    function list followed by `(` is a call, never a tensor.
"""
import re

from . import expr as ex
from .errors import SemanticError, SodaSyntaxError
from .types import is_type_name

_TOKEN = re.compile(r'''
    (?P<ws>[ \t\r\n]+|\#[^\n]*)
  | (?P<num>
        (?:(?:\d*\.\d+|\d+\.)(?:[+-]?[Ee]\d+)?|\d+[+-]?[Ee]\d+)[FfLl]?
      | 0[Xx][0-9a-fA-F]+(?:[Uu][Ll][Ll]?|[Ll]?[Ll]?[Uu]?)
      | 0[Bb][01]+(?:[Uu][Ll][Ll]?|[Ll]?[Ll]?[Uu]?)
      | \d+(?:[Uu][Ll][Ll]?|[Ll]?[Ll]?[Uu]?)
    )
  | (?P<id>[^\W\d]\w*)
  | (?P<op>\|\||&&|==|!=|<=|>=|[-+*/%<>|^&~!(),\[\]=:.])
''', re.X)

_STATEMENT_HEADS = ('kernel', 'burst', 'unroll', 'iterate',
                    'input', 'param', 'local', 'output')


class Token:
  __slots__ = ('kind', 'text', 'line', 'col')

  def __init__(self, kind, text, line, col):
    self.kind, self.text, self.line, self.col = kind, text, line, col


def tokenize(text):
  tokens, pos, line, bol = [], 0, 1, 0
  while pos < len(text):
    m = _TOKEN.match(text, pos)
    if m is None:
      raise SodaSyntaxError('unexpected character %r' % text[pos], line,
                            pos - bol + 1)
    kind = m.lastgroup
    if kind != 'ws':
      tokens.append(Token(kind, m.group(kind), line, pos - bol + 1))
    newlines = text.count('\n', pos, m.end())
    if newlines:
      line += newlines
      bol = text.rfind('\n', pos, m.end()) + 1
    pos = m.end()
  tokens.append(Token('eof', '', line, pos - bol + 1))
  return tokens


# ---------------------------------------------------------------------------
# statements
# ---------------------------------------------------------------------------
class InputStmt:
  """`input [dram B(.B)*] T: name[(t0, t1, ..., *)]`; `tile_size` carries the
  trailing 0 for the streamed dimension like the reference's node does
  (grammar.py:55-61)."""

  def __init__(self, haoda_type, name, tile_size, dram=()):
    self.haoda_type, self.name = haoda_type, name
    self.tile_size = tuple(tile_size) + (0,)
    self.dram = tuple(dram) or (0,)

  def __str__(self):
    text = 'input %s: %s' % (self.haoda_type, self.name)
    if self.tile_size[:-1]:
      text += '(%s, *)' % ', '.join(map(str, self.tile_size[:-1]))
    return text


class Let:
  def __init__(self, haoda_type, name, expr):
    self.declared_type, self.name, self.expr = haoda_type, name, expr

  def __str__(self):
    text = '%s = %s' % (self.name, ex._strip_parens(ex.soda_text(self.expr)))
    if self.declared_type is not None:
      text = '%s %s' % (self.declared_type, text)
    return text


class ComputeStmt:
  """`local` / `output` statement: `T: let* name(idx) = expr`."""
  kind = None

  def __init__(self, haoda_type, lets, ref, expr, dram=()):
    self.haoda_type = haoda_type
    self.lets = tuple(lets)
    self.ref = ref          # ex.Load holding the store index
    self.expr = expr
    self.dram = tuple(dram) or (0,)

  @property
  def name(self):
    return self.ref.name

  def __str__(self):
    lets = ''
    if self.lets:
      lets = '\n  %s\n ' % '\n  '.join(map(str, self.lets))
    return '%s %s:%s %s = %s' % (
        self.kind, self.haoda_type, lets, ex.soda_text(self.ref),
        ex._strip_parens(ex.soda_text(self.expr)))


class LocalStmt(ComputeStmt):
  kind = 'local'


class OutputStmt(ComputeStmt):
  kind = 'output'


class ParamStmt:
  def __init__(self, haoda_type, name, size, attrs=(), dram=()):
    self.haoda_type, self.name = haoda_type, name
    self.size, self.attrs = tuple(size), tuple(attrs)
    self.dram = tuple(dram) or (0,)

  def __str__(self):
    return 'param %s%s: %s%s' % (
        self.haoda_type, ''.join(', %s' % a for a in self.attrs), self.name,
        ''.join('[%d]' % s for s in self.size))


class Program:
  """A parsed `.soda` file (the reference's `SodaProgram`, grammar.py:129-160)."""

  def __init__(self, burst_width, iterate, app_name, unroll_factor,
               input_stmts, param_stmts, local_stmts, output_stmts):
    self.burst_width, self.iterate = burst_width, iterate
    self.app_name, self.unroll_factor = app_name, unroll_factor
    self.input_stmts = tuple(input_stmts)
    self.param_stmts = tuple(param_stmts)
    self.local_stmts = tuple(local_stmts)
    self.output_stmts = tuple(output_stmts)
    # The one input that carries tile sizes fixes them for all; an input
    # WITHOUT tile sizes after one WITH them is a mismatch in the reference
    # too (grammar.py:134-149 compares whole tuples).
    tile = None
    for stmt in self.input_stmts:
      if tile is not None:
        if tile != stmt.tile_size:
          raise SemanticError("tile size %s doesn't match previous one %s" %
                              (stmt.tile_size, tile))
      elif stmt.tile_size[:-1]:
        tile = stmt.tile_size
    if tile is None:   # 1-D program
      tile = self.input_stmts[-1].tile_size
    self.tile_size = tile
    self.dim = len(tile)

  def __str__(self):
    parts = ['burst width: %d' % self.burst_width,
             'iterate: %d' % self.iterate,
             'kernel: %s' % self.app_name,
             'unroll factor: %d' % self.unroll_factor]
    for group in (self.input_stmts, self.param_stmts, self.local_stmts,
                  self.output_stmts):
      if group:
        parts.append('\n'.join(map(str, group)))
    return '\n'.join(parts)


# ---------------------------------------------------------------------------
# the parser
# ---------------------------------------------------------------------------
class _Parser:
  def __init__(self, text):
    self.toks = tokenize(text)
    self.pos = 0

  # -- token helpers --------------------------------------------------------
  def peek(self, ahead=0):
    return self.toks[min(self.pos + ahead, len(self.toks) - 1)]

  def fail(self, message, tok=None):
    tok = tok or self.peek()
    raise SodaSyntaxError(message, tok.line, tok.col)

  def next(self):
    tok = self.toks[self.pos]
    if tok.kind != 'eof':
      self.pos += 1
    return tok

  def accept(self, text):
    if self.peek().text == text and self.peek().kind != 'eof':
      return self.next()
    return None

  def expect(self, text):
    tok = self.accept(text)
    if tok is None:
      self.fail('expected %r, found %r' % (text, self.peek().text or 'end of file'))
    return tok

  def ident(self, what='identifier'):
    tok = self.peek()
    if tok.kind != 'id':
      self.fail('expected %s, found %r' % (what, tok.text or 'end of file'))
    return self.next().text

  def type_name(self):
    tok = self.peek()
    if tok.kind != 'id' or not is_type_name(tok.text):
      self.fail('expected a type, found %r' % (tok.text or 'end of file'))
    return self.next().text

  def signed_int(self, what='integer'):
    """textX INT: optional sign, decimal digits."""
    sign = 1
    if self.peek().text in ('+', '-') and self.peek(1).kind == 'num':
      sign = -1 if self.next().text == '-' else 1
    tok = self.peek()
    if tok.kind != 'num' or not tok.text.isdigit():
      self.fail('expected %s, found %r' % (what, tok.text or 'end of file'))
    return sign * int(self.next().text)

  def c_int(self):
    """haoda `Int`: optional sign, hex/bin/oct/dec with C suffix -> value."""
    sign = 1
    if self.peek().text in ('+', '-'):
      sign = -1 if self.next().text == '-' else 1
    tok = self.peek()
    if tok.kind != 'num' or not re.match(
        r'(?:0[Xx][0-9a-fA-F]+|0[Bb][01]+|\d+)[UuLl]*\Z', tok.text):
      self.fail('expected an integer, found %r' % (tok.text or 'end of file'))
    text = self.next().text.rstrip('UuLl')
    low = text.lower()
    if low.startswith('0x'):
      value = int(text, 16)
    elif low.startswith('0b'):
      value = int(text[2:], 2)
    elif len(text) > 1 and text[0] == '0':
      value = int(text, 8)
    else:
      value = int(text)
    return sign * value

  # -- expressions ----------------------------------------------------------
  def expression(self, level=0):
    if level == len(ex.LEVELS):
      return self.unary()
    ops = ex.LEVELS[level]
    operands, operators = [self.expression(level + 1)], []
    while self.peek().kind == 'op' and self.peek().text in ops:
      operators.append(self.next().text)
      operands.append(self.expression(level + 1))
    if not operators:
      return operands[0]
    return ex.Chain(level, operands, operators)

  def unary(self):
    ops = []
    while self.peek().kind == 'op' and self.peek().text in ex.UNARY_OPS:
      ops.append(self.next().text)
    return ex.make_unary(ops, self.operand())

  def operand(self):
    tok = self.peek()
    if tok.kind == 'num':
      return ex.Num(self.next().text)
    if tok.kind == 'op' and tok.text == '(':
      self.next()
      inner = self.expression()
      self.expect(')')
      return inner
    if tok.kind == 'id':
      call_like = self.peek(1).text == '(' and self.peek(1).kind == 'op'
      if call_like and is_type_name(tok.text):
        self.next(); self.next()
        inner = self.expression()
        self.expect(')')
        return ex.Cast(tok.text, inner)
      if call_like and tok.text in ex.FUNC_NAMES:
        self.next(); self.next()
        args = [self.expression()]
        while self.accept(','):
          args.append(self.expression())
        self.expect(')')
        return ex.Call(tok.text, args)
      if call_like:
        return self.reference()
      self.next()
      idx = []
      while self.accept('['):
        idx.append(self.c_int())
        self.expect(']')
      return ex.Var(tok.text, idx)
    self.fail('expected an operand, found %r' % (tok.text or 'end of file'))

  def reference(self):
    name = self.ident('tensor name')
    self.expect('(')
    idx = [self.signed_int('tensor index')]
    while self.accept(','):
      idx.append(self.signed_int('tensor index'))
    self.expect(')')
    if self.peek().text == '~' and self.peek(1).kind == 'num':
      # `~ latency` annotation: FPGA pipeline hint, no meaning on a GPU
      self.next()
      self.c_int()
    return ex.Load(name, idx)

  # -- statements -----------------------------------------------------------
  def dram(self):
    banks = []
    if self.peek().kind == 'id' and self.peek().text == 'dram' and \
        self.peek(1).kind == 'num':
      self.next()
      while True:
        tok = self.next()
        if tok.kind != 'num' or not re.match(r'\d+(\.\d+)*\.?\Z', tok.text):
          self.fail('expected a DRAM bank number', tok)
        banks.extend(int(b) for b in tok.text.split('.') if b)
        if tok.text.endswith('.') or self.accept('.'):
          continue
        break
    return banks

  def input_stmt(self):
    dram = self.dram()
    haoda_type = self.type_name()
    self.expect(':')
    name = self.ident('input name')
    tiles = []
    if self.accept('('):
      while not self.accept('*'):
        tiles.append(self.signed_int('tile size'))
        self.expect(',')
      self.expect(')')
    return InputStmt(haoda_type, name, tiles, dram)

  def compute_stmt(self, cls, with_dram):
    dram = self.dram() if with_dram else ()
    haoda_type = self.type_name()
    self.expect(':')
    lets = []
    while True:
      t0, t1, t2 = self.peek(), self.peek(1), self.peek(2)
      typed = (t0.kind == 'id' and is_type_name(t0.text) and t1.kind == 'id'
               and t2.text == '=' and t2.kind == 'op')
      untyped = t0.kind == 'id' and t1.text == '=' and t1.kind == 'op'
      if not (typed or untyped):
        break
      let_type = self.next().text if typed else None
      let_name = self.next().text
      self.expect('=')
      lets.append(Let(let_type, let_name, self.expression()))
    ref = self.reference()
    self.expect('=')
    return cls(haoda_type, lets, ref, self.expression(), dram)

  def param_stmt(self):
    dram = self.dram()
    haoda_type = self.type_name()
    attrs = []
    while self.accept(','):
      word = self.ident('param attribute')
      if word == 'dup':
        attrs.append('dup %d' % self.c_int())
      elif word == 'partition':
        strategy = self.ident('partition strategy')
        text = 'partition %s' % strategy
        if strategy == 'cyclic':
          if self.ident() != 'factor':
            self.fail("expected 'factor'")
          self.expect('=')
          text += ' factor=%d' % self.c_int()
        elif strategy != 'complete':
          self.fail("expected 'complete' or 'cyclic'")
        if self.peek().kind == 'id' and self.peek().text == 'dim' and \
            self.peek(1).text == '=':
          self.next(); self.next()
          text += ' dim=%d' % self.c_int()
        attrs.append(text)
      else:
        self.fail("expected 'dup' or 'partition'")
    self.expect(':')
    name = self.ident('param name')
    size = []
    while self.accept('['):
      size.append(self.signed_int('param size'))
      self.expect(']')
    return ParamStmt(haoda_type, name, size, attrs, dram)

  def program(self):
    header = {}
    groups = {'input': [], 'param': [], 'local': [], 'output': []}
    closed = set()
    last = None

    def set_header(key, value, tok):
      if key in header:
        self.fail('duplicate %r' % key, tok)
      header[key] = value

    while self.peek().kind != 'eof':
      tok = self.peek()
      if tok.kind != 'id' or tok.text not in _STATEMENT_HEADS:
        self.fail('expected a statement (one of %s), found %r' % (
            ', '.join(_STATEMENT_HEADS), tok.text))
      self.next()
      kind = tok.text
      if last in groups and last != kind:
        closed.add(last)
      if kind in closed:
        self.fail('%s statements must be contiguous' % kind, tok)
      last = kind
      if kind == 'kernel':
        self.expect(':')
        set_header('kernel', self.ident('kernel name'), tok)
      elif kind == 'burst':
        if self.ident() != 'width':
          self.fail("expected 'width'")
        self.expect(':')
        set_header('burst width', self.signed_int(), tok)
      elif kind == 'unroll':
        if self.ident() != 'factor':
          self.fail("expected 'factor'")
        self.expect(':')
        set_header('unroll factor', self.signed_int(), tok)
      elif kind == 'iterate':
        self.expect(':')
        set_header('iterate', self.signed_int(), tok)
      elif kind == 'input':
        groups['input'].append(self.input_stmt())
      elif kind == 'param':
        groups['param'].append(self.param_stmt())
      elif kind == 'local':
        groups['local'].append(self.compute_stmt(LocalStmt, False))
      else:
        groups['output'].append(self.compute_stmt(OutputStmt, True))
    for key in ('burst width', 'iterate', 'kernel', 'unroll factor'):
      if key not in header:
        self.fail('missing %r' % key)
    if not groups['input']:
      self.fail('a program needs at least one input statement')
    if not groups['output']:
      self.fail('a program needs at least one output statement')
    return Program(header['burst width'], header['iterate'], header['kernel'],
                   header['unroll factor'], groups['input'], groups['param'],
                   groups['local'], groups['output'])


def parse(text):
  """Parses `.soda` source text into a `Program`."""
  return _Parser(text).program()


def parse_expression(text):
  """Parses one expression (used by tests)."""
  p = _Parser(text)
  node = p.expression()
  if p.peek().kind != 'eof':
    p.fail('unexpected %r after expression' % p.peek().text)
  return node
