"""Generator of random SODA programs for the parity tests: asymmetric and
one-sided windows, negative-only offsets, fan-in and fan-out DAGs, `let`s, casts,
integer division, several inputs, 2-D and 3-D.  Plain numpy, so that
tests/golden/make_golden.py (python3.9, next to the reference) and the tests
share it.  The texts the committed fixtures were made from are stored with them
(tests/golden/random_programs.json): numpy does not promise the same random
stream in every version."""
import numpy as np


def random_program(rng, seed):
  dim = 3 if rng.random() < 0.25 else 2
  floaty = rng.random() < 0.6
  dtype = rng.choice(['float', 'float', 'float', 'double']) if floaty else \
      rng.choice(['uint16', 'int32', 'int32', 'uint8', 'int64', 'int16'])
  n_inputs = 1 if rng.random() < 0.75 else 2
  n_locals = int(rng.integers(0, 4))
  iterate = int(rng.integers(1, 6)) if n_inputs == 1 else 1
  reach = 2 if dim == 2 else 1
  names = ['in%d' % i for i in range(n_inputs)]
  lines = ['kernel: rnd%d' % seed, 'burst width: 512', 'unroll factor: 2',
           'iterate: %d' % iterate]
  tile = ', '.join(['32'] * (dim - 1))
  for i, n in enumerate(names):
    lines.append('input %s: %s(%s, *)' % (dtype, n, tile) if i == n_inputs - 1
                 else 'input %s: %s' % (dtype, n))
  if n_inputs == 2:   # only the LAST input may carry the tile (reference quirk)
    lines[-2], lines[-1] = 'input %s: %s' % (dtype, names[0]), \
        'input %s: %s(%s, *)' % (dtype, names[1], tile)

  def offset():
    # sometimes one-sided windows: only negative or only positive offsets
    mode = rng.integers(0, 4)
    lo, hi = (-reach, reach) if mode < 2 else ((-reach, 0) if mode == 2 else (0, reach))
    return tuple(int(rng.integers(lo, hi + 1)) for _ in range(dim))

  def load(name):
    return '%s(%s)' % (name, ', '.join(map(str, offset())))

  def literal():
    if floaty:
      return rng.choice(['0.25f', '0.5f', '1.5f', '0.125f', '3.0f', '0.2f', '0.3'])
    return str(int(rng.integers(1, 5)))

  def expression(available, must_use):
    terms = []
    pool = list(must_use) + [rng.choice(available)
                             for _ in range(int(rng.integers(1, 4)))]
    for name in pool:
      t = load(name)
      r = rng.random()
      if r < 0.3:
        t = '%s * %s' % (t, literal())
      elif r < 0.4:
        t = '(%s - %s)' % (t, load(rng.choice(available)))
      elif r < 0.5 and not floaty:
        t = '(%s + %s) / 3' % (t, load(name))
      elif r < 0.5 and floaty:
        t = '%s / %s' % (t, literal())
      terms.append(t)
    text = terms[0]
    for t in terms[1:]:
      text += rng.choice([' + ', ' - ', ' + ']) + t
    if rng.random() < 0.3:
      text = '(%s) * %s' % (text, literal())
    return text

  available = list(names)
  unused = list(names)
  for k in range(n_locals):
    name = 'loc%d' % k
    use = [unused.pop(0)] if unused and rng.random() < 0.7 else []
    if rng.random() < 0.25:
      lines.append('local %s: t = %s %s(%s) = t + %s' % (
          dtype, expression(available, use), name, ', '.join(['0'] * dim),
          load(rng.choice(available))))
    else:
      lines.append('local %s: %s(%s) = %s' % (
          dtype, name, ', '.join(map(str, offset())), expression(available, use)))
    available.append(name)
    unused.append(name)
  # the output reads everything still unused, so that no stage is dead
  lines.append('output %s: out(%s) = %s' % (
      dtype, ', '.join(['0'] * dim), expression(available, unused)))
  return '\n'.join(lines) + '\n', dim, dtype, iterate




def operator_program(rng, seed):
  """Integer (and a few float-compare) programs over the operators the arithmetic
  generator above never emits: % & | ^ comparisons && || unary - ~ !, hexadecimal,
  octal and suffixed literals, casts between widths.  Every window contains the
  store point, so the reference's own CPU loops have a defined answer."""
  dim = 3 if rng.random() < 0.2 else 2
  dtype = str(rng.choice(['uint16', 'int32', 'uint8', 'int16', 'uint32', 'int64',
                          'int8', 'uint64']))
  other = str(rng.choice(['int32', 'uint16', 'int16', 'uint8']))
  iterate = int(rng.integers(1, 4))
  reach = 2 if dim == 2 else 1
  lines = ['kernel: ops%d' % seed, 'burst width: 512', 'unroll factor: 2',
           'iterate: %d' % iterate,
           'input %s: a(%s, *)' % (dtype, ', '.join(['32'] * (dim - 1)))]

  def ref(name, centre=False):
    idx = [0] * dim if centre else [int(rng.integers(-reach, reach + 1))
                                    for _ in range(dim)]
    return '%s(%s)' % (name, ', '.join(map(str, idx)))

  def literal():
    return str(rng.choice(['1', '2', '3', '5', '7', '0x0f', '0x3', '017', '3u', '6l',
                           '0xffu', '9']))

  def term(names):
    n = str(rng.choice(names))
    r = rng.random()
    if r < 0.15:
      return '(%s %% %s)' % (ref(n), str(rng.choice(['3', '5', '7', '0x10'])))
    if r < 0.30:
      return '(%s & %s)' % (ref(n), literal())
    if r < 0.40:
      return '(%s | %s)' % (ref(n), ref(str(rng.choice(names))))
    if r < 0.50:
      return '(%s ^ %s)' % (ref(n), ref(str(rng.choice(names))))
    if r < 0.60:
      return '(%s %s %s)' % (ref(n), str(rng.choice(['<', '<=', '>', '>=', '==', '!='])),
                             ref(str(rng.choice(names))))
    if r < 0.68:
      return '(%s > %s && %s != %s)' % (ref(n), literal(), ref(n), literal())
    if r < 0.74:
      return '(%s < %s || %s == %s)' % (ref(n), literal(), ref(str(rng.choice(names))),
                                        literal())
    if r < 0.80:
      return '(-%s)' % ref(n)
    if r < 0.86:
      return '(~%s & 0xff)' % ref(n)
    if r < 0.90:
      return '(!%s)' % ref(n)
    if r < 0.95:
      return '%s(%s) * %s' % (other, ref(n), literal())
    return '%s / %s' % (ref(n), str(rng.choice(['2', '3', '4'])))

  def expression(names):
    parts = [ref(str(rng.choice(names)), centre=True)]     # the store point is read
    parts += [term(names) for _ in range(int(rng.integers(2, 5)))]
    text = parts[0]
    for t in parts[1:]:
      text += str(rng.choice([' + ', ' - ', ' + '])) + t
    return text

  names = ['a']
  if rng.random() < 0.5:
    lines.append('local %s: m(%s) = %s' % (other, ', '.join(['0'] * dim),
                                           expression(names)))
    names.append('m')
  use = expression(names)
  if 'm' in names and 'm(' not in use:
    use += ' + ' + ref('m', centre=True)
  lines.append('output %s: out(%s) = %s' % (dtype, ', '.join(['0'] * dim), use))
  return '\n'.join(lines) + '\n', dim, dtype, iterate


def structure_program(rng, seed):
  """Programs that stress the STRUCTURE the other generators keep small: three
  inputs, windows reaching up to 4 cells, up to seven stages, several outputs
  (with `iterate` > 1 when inputs and outputs pair up), 3-D with two inputs, a local
  that only another local reads.  Every window contains the store point."""
  dim = 3 if rng.random() < 0.3 else 2
  dtype = str(rng.choice(['float', 'float', 'int32', 'double', 'uint16']))
  floaty = dtype in ('float', 'double')
  n_inputs = int(rng.integers(1, 4))
  n_outputs = int(rng.integers(1, 3))
  n_locals = int(rng.integers(1, 6))
  iterate = int(rng.integers(2, 6)) if n_inputs == n_outputs else 1
  reach = int(rng.integers(1, 5)) if dim == 2 else int(rng.integers(1, 3))
  ins = ['in%d' % i for i in range(n_inputs)]
  lines = ['kernel: st%d' % seed, 'burst width: 512', 'unroll factor: 4',
           'iterate: %d' % iterate]
  tile = ', '.join(['64'] * (dim - 1))
  for i, n in enumerate(ins):      # only the LAST input may carry the tile
    lines.append('input %s: %s(%s, *)' % (dtype, n, tile) if i == n_inputs - 1
                 else 'input %s: %s' % (dtype, n))

  def ref(name, centre=False):
    idx = [0] * dim if centre else [int(rng.integers(-reach, reach + 1))
                                    for _ in range(dim)]
    return '%s(%s)' % (name, ', '.join(map(str, idx)))

  def literal():
    return str(rng.choice(['0.25f', '0.5f', '0.125f', '2.0f'] if floaty
                          else ['1', '2', '3']))

  def expression(names, must):
    parts = [ref(str(rng.choice(names)), centre=True)]
    pool = list(must) + [str(rng.choice(names)) for _ in range(int(rng.integers(1, 4)))]
    for n in pool:
      t = ref(n)
      if rng.random() < 0.4:
        t = '%s * %s' % (t, literal())
      parts.append(t)
    text = parts[0]
    for t in parts[1:]:
      text += str(rng.choice([' + ', ' - ', ' + '])) + t
    if floaty and rng.random() < 0.5:
      text = '(%s) * %s' % (text, literal())
    return text

  names = list(ins)
  unused = list(ins)
  for k in range(n_locals):
    name = 'loc%d' % k
    must = [unused.pop(0)] if unused else []
    lines.append('local %s: %s(%s) = %s' % (dtype, name, ', '.join(['0'] * dim),
                                            expression(names, must)))
    names.append(name)
    unused.append(name)
  for k in range(n_outputs):
    share = [unused.pop(0) for _ in range((len(unused) + n_outputs - k - 1)
                                          // (n_outputs - k))] if unused else []
    lines.append('output %s: out%s(%s) = %s' % (
        dtype, '' if k == 0 else str(k), ', '.join(['0'] * dim),
        expression(names, share)))
  return '\n'.join(lines) + '\n', dim, dtype, iterate


def cube_program(rng, seed):
  """3-D iteration chains for the deep 3-D kernels (single-wave depth 2, wave-
  pipelined and block-form depth 4): one input, one output, optionally a local in
  between, 4..13 iterations, windows inside {-1,0,1}^3 INCLUDING diagonals (a read
  that leaves the band in y and the plane in z at once)."""
  dtype = str(rng.choice(['float', 'float', 'float', 'int32']))
  floaty = dtype == 'float'
  iterate = int(rng.integers(4, 14))
  lines = ['kernel: cube%d' % seed, 'burst width: 512', 'unroll factor: 2',
           'iterate: %d' % iterate, 'input %s: a(32, 32, *)' % dtype]

  def offsets(n):
    seen = {(0, 0, 0)}
    while len(seen) < n:
      seen.add(tuple(int(v) for v in rng.integers(-1, 2, size=3)))
    rest = sorted(seen - {(0, 0, 0)})
    rng.shuffle(rest)
    return [(0, 0, 0)] + [tuple(o) for o in rest]

  def expression(name):
    terms = []
    for o in offsets(int(rng.integers(4, 10))):
      t = '%s(%d, %d, %d)' % ((name,) + o)
      if rng.random() < 0.3:
        t += ' * %s' % (str(rng.choice(['0.5f', '0.25f', '2.0f'])) if floaty
                        else str(rng.choice(['2', '3'])))
      terms.append(t)
    text = terms[0]
    for t in terms[1:]:
      text += str(rng.choice([' + ', ' + ', ' - '])) + t
    scale = str(rng.choice(['0.125f', '0.0625f', '0.2f'])) if floaty else '1'
    return '(%s) * %s' % (text, scale) if floaty else text

  src = 'a'
  if rng.random() < 0.35:
    lines.append('local %s: m(0, 0, 0) = %s' % (dtype, expression('a')))
    src = 'm'
  lines.append('output %s: out(0, 0, 0) = %s' % (dtype, expression(src)))
  return '\n'.join(lines) + '\n', 3, dtype, iterate


def program_set(n_plain=40, n_deep=16, n_ops=16, n_struct=16, n_cube=12):
  """[(key, text, dim, iterate, shape)] - the programs of the GPU random tests."""
  out = []
  for seed in range(n_plain):
    rng = np.random.default_rng(1000 + seed)
    text, dim, dtype, iterate = random_program(rng, seed)
    out.append(('plain%d' % seed, text, dim, iterate))
  for seed in range(n_deep):
    rng = np.random.default_rng(5000 + seed)
    while True:
      text, dim, dtype, iterate = random_program(rng, seed)
      if dim == 2 and 'input %s: in1' % dtype not in text:
        break
    deep = int(rng.integers(8, 21))
    text = text.replace('iterate: %d\n' % iterate, 'iterate: %d\n' % deep)
    out.append(('deep%d' % seed, text, dim, deep))
  for seed in range(n_ops):
    rng = np.random.default_rng(9000 + seed)
    text, dim, dtype, iterate = operator_program(rng, seed)
    out.append(('ops%d' % seed, text, dim, iterate))
  for seed in range(n_struct):
    rng = np.random.default_rng(12000 + seed)
    text, dim, dtype, iterate = structure_program(rng, seed)
    out.append(('struct%d' % seed, text, dim, iterate))
  for seed in range(n_cube):
    rng = np.random.default_rng(15000 + seed)
    text, dim, dtype, iterate = cube_program(rng, seed)
    out.append(('cube%d' % seed, text, dim, iterate))
  return out


if __name__ == '__main__':      # writes the program texts the fixtures are made from
  import json
  import os
  here = os.path.dirname(os.path.abspath(__file__))
  with open(os.path.join(here, 'golden', 'random_programs.json'), 'w') as f:
    json.dump({k: dict(text=t, dim=d, iterate=i) for k, t, d, i in program_set()},
              f, indent=1, sort_keys=True)
