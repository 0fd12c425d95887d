"""kernel_common.split_early_prefix: the 3-D block form computes the part of a level's
expression that reads only planes of earlier steps BEFORE the per-plane barrier
(kernel_stream3d_blk.emit, early=1).  The split must not change a single operation or
their order: early first, then the rest on its value, is the expression itself."""
import os
import re
import sys

import numpy as np

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, 'soda-compiler_amd'))
from soda_hip.codegen import kernel_common as kc        # noqa: E402
from soda_hip.codegen import spec as specmod            # noqa: E402

LATE = lambda tensor, rel: rel[2] == 1                   # noqa: E731


def evaluate(text, values, pre=None):
  """`text` in float32 arithmetic, left to right as C evaluates it (numpy scalars)."""
  env = {'np': np}
  names = {}

  def load(tensor, rel):
    key = 'v%d' % len(names)
    names[key] = values[(tensor, rel)]
    return key
  text = specmod.substitute_loads(text, load)
  text = re.sub(r'((?:\d+\.?\d*|\.\d+)(?:[eE][+-]?\d+)?)f', r'np.float32(\1)', text)
  if pre is not None:
    text = text.replace(kc.PRE_MARK, 'pre')
    env['pre'] = pre
  env.update(names)
  return eval(text, env)       # noqa: S307 - test-local arithmetic


def test_the_samples_split_where_the_new_plane_enters():
  j = ('(({t1:0,0,0} + {t1:1,0,0} + {t1:-1,0,0} + {t1:0,1,0} + {t1:0,-1,0} + {t1:0,0,1} + '
       '{t1:0,0,-1}) * 0.142857142f)')
  early, rest = kc.split_early_prefix(j, LATE)
  assert early == '({t1:0,0,0} + {t1:1,0,0} + {t1:-1,0,0} + {t1:0,1,0} + {t1:0,-1,0})'
  assert rest == '((@PRE@ + {t1:0,0,1} + {t1:0,0,-1}) * 0.142857142f)'
  h = ('((.125f * ({in:1,0,0} - (2.f * {in:0,0,0}) + {in:-1,0,0})) + (.125f * ({in:0,1,0} - '
       '(2.f * {in:0,0,0}) + {in:0,-1,0})) + (.125f * ({in:0,0,1} - (2.f * {in:0,0,0}) + '
       '{in:0,0,-1})) + {in:0,0,0})')
  early, rest = kc.split_early_prefix(h, LATE)
  assert early.count('{in:') == 6 and '{in:0,0,1}' not in early
  assert rest.startswith('(@PRE@ + (.125f * ({in:0,0,1}') and rest.endswith('+ {in:0,0,0})')


def test_nothing_is_taken_where_the_order_would_change():
  for text in ('({a:0,0,1} + {a:0,0,0})',                # the new plane comes first
               '((2 * 3) + {a:0,0,1})',                   # integer literals alone
               '({a:0,0,0} + {a:1,0,0} * {a:0,0,1})',     # mixed precedence at one level
               '({a:0,0,0} + {a:0,0,1} + {a:1,0,0})',     # one operand before the new plane
               '(sqrt({a:0,0,0}) + {a:0,0,1})',           # not a plain expression
               '({a:0,0,0} + {a:1,0,0})'):                # nothing late at all
    assert kc.split_early_prefix(text, LATE) == (None, text)
  early, rest = kc.split_early_prefix('(-{a:0,0,0} + {a:1,0,0} + {a:0,0,1})', LATE)
  assert early == '((-{a:0,0,0}) + {a:1,0,0})' and rest == '(@PRE@ + {a:0,0,1})'


def random_expression(rng, depth=0):
  """A fully parenthesised chain of one operator class per level, as the front end
  prints them."""
  loads = ['{a:%d,%d,%d}' % (x, y, z) for x in (-1, 0, 1) for y in (-1, 0, 1)
           for z in (-1, 0, 1)]
  ops = ['+', '-'] if rng.random() < 0.6 else ['*', '/']
  parts = []
  for _ in range(int(rng.integers(2, 6))):
    kind = rng.random()
    if kind < 0.55 or depth >= 2:
      parts.append(str(rng.choice(loads)))
    elif kind < 0.7:
      parts.append('%.3ff' % (0.25 + rng.random()))
    else:
      parts.append(random_expression(rng, depth + 1))
  text = parts[0]
  for p in parts[1:]:
    text += ' %s %s' % (rng.choice(ops), p)
  return '(%s)' % text


def test_early_then_rest_is_the_expression_bit_for_bit():
  rng = np.random.default_rng(5)
  taken = 0
  for _ in range(400):
    text = random_expression(rng)
    values = {('a', (x, y, z)): np.float32(0.5 + rng.random())
              for x in (-1, 0, 1) for y in (-1, 0, 1) for z in (-1, 0, 1)}
    early, rest = kc.split_early_prefix(text, LATE)
    whole = evaluate(text, values)
    if early is None:
      assert rest == text
      continue
    taken += 1
    assert not any(LATE(t, tuple(int(v) for v in rel.split(',')))
                   for t, rel in specmod.LOAD_RE.findall(early))
    pre = evaluate(early, values)
    assert isinstance(pre, np.float32)
    got = evaluate(rest, values, pre=pre)
    assert np.float32(got).tobytes() == np.float32(whole).tobytes(), (text, early, rest)
  assert taken > 80
