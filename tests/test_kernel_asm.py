"""kernel_asm: the expression of a plain float stage as a list of two-operand
operations must be the C++ expression's own operations in the same association -
checked by evaluating both in IEEE float32 on random operands (numpy), bit for bit."""
import re

import numpy as np
import pytest

from soda_hip.codegen import kernel, kernel_asm
from soda_hip.codegen import spec as specmod

from test_codegen import spec_of

EXPRESSIONS = [
    '(({t1:0,0,0} + {t1:1,0,0} + {t1:-1,0,0} + {t1:0,1,0} + {t1:0,-1,0} + {t1:0,0,1} + '
    '{t1:0,0,-1}) * 0.142857142f)',
    '((.125f * ({in:1,0,0} - (2.f * {in:0,0,0}) + {in:-1,0,0})) + (.125f * ({in:0,1,0} - '
    '(2.f * {in:0,0,0}) + {in:0,-1,0})) + (.125f * ({in:0,0,1} - (2.f * {in:0,0,0}) + '
    '{in:0,0,-1})) + {in:0,0,0})',
    '({a:0,0} - {a:1,0} - {a:0,1} * 3 + 2.5f - {a:1,1} * (1.5f - 4) + -{a:0,0} * -2.f)',
    '(7.7f * ({x:0,0} + {y:0,1}) - ({x:0,0} + {y:0,1}) * ({x:0,0} + {y:0,1}))',
    '(1e-3f * {a:0,0} + .5e2f - (3 - {a:1,0}))',
]


def evaluate_ops(ops, result, leaves):
  values = []
  for op, a, b in ops:
    def val(x):
      return values[x[1]] if x[0] == 't' else x[1] if x[0] == 'num' else leaves[(x[1], x[2])]
    with np.errstate(all='ignore'):
      if op == 'neg':
        values.append(np.float32(-val(a)))
      else:
        values.append(np.float32({'+': np.add, '-': np.subtract, '*': np.multiply}[op](
            val(a), val(b))))
  return values[result[1]]


def evaluate_text(text, leaves):
  names = {}

  def leaf(m):
    key = (m.group(1), tuple(int(v) for v in m.group(2).split(',')))
    names.setdefault(key, 'v%d' % len(names))
    return names[key]
  code = specmod.LOAD_RE.sub(leaf, text)
  code = re.sub(r'(?<![\w.])((?:\d+\.?\d*|\.\d+)(?:[eE][+-]?\d+)?)[fF]?', r'np.float32(\1)',
                code)
  env = {'np': np}
  env.update({name: leaves[key] for key, name in names.items()})
  with np.errstate(all='ignore'):
    return np.float32(eval(code, env))   # noqa: S307 - test expression from this file


@pytest.mark.parametrize('text', EXPRESSIONS)
def test_operation_list_equals_the_expression(text):
  ops, result = kernel_asm.lower(kernel_asm.parse(text))
  assert result[0] == 't' and ops
  rng = np.random.default_rng(5)
  keys = {(m.group(1), tuple(int(v) for v in m.group(2).split(',')))
          for m in specmod.LOAD_RE.finditer(text)}
  for trial in range(200):
    scale = np.float32(10.0 ** rng.integers(-30, 30))
    leaves = {k: np.float32(rng.standard_normal()) * scale for k in keys}
    got, want = evaluate_ops(ops, result, leaves), evaluate_text(text, leaves)
    assert got.tobytes() == want.tobytes() or (np.isnan(got) and np.isnan(want))


def test_common_subexpressions_are_computed_once_and_constants_folded():
  ops, _ = kernel_asm.lower(kernel_asm.parse(EXPRESSIONS[1]))
  assert sum(1 for op, a, b in ops if op == '*' and a == ('num', np.float32(2.0))) == 1
  ops, _ = kernel_asm.lower(kernel_asm.parse('({a:0,0} * (1.5f - 4))'))
  assert ops == [('*', ('load', 'a', (0, 0)), ('num', np.float32(-2.5)))]
  assert kernel_asm.literal(np.float32(0.142857142)) == '0x3e124925'


def test_scope():
  stage = dict(lets=[], expr='({a:0,0} / 3.f)')
  assert not kernel_asm.supported(stage)                  # division
  assert not kernel_asm.supported(dict(lets=[dict(name='l')], expr='({a:0,0} + 1.f)'))
  assert not kernel_asm.supported(dict(lets=[], expr='sqrt({a:0,0})'))
  assert not kernel_asm.supported(dict(lets=[], expr='{a:0,0}'))    # nothing to compute
  assert kernel_asm.supported(dict(lets=[], expr=EXPRESSIONS[0]))


def test_block_form_with_hand_ordered_arithmetic_compiles(tmp_path):
  """asm_sched=1: the 3-D block form with every level's cells as interleaved inline
  instructions builds for gfx950; an integer program keeps the C++ form."""
  spec = spec_of('jacobi3d', iterate=8)
  text, table = kernel.generate(spec, deep3d='blk', blk_asm_sched=1)
  blk = [k for k in table if k.get('stack')][0]
  assert blk['asm_sched'] == 1 and 'asm volatile("v_add_f32 %0, %1, %2"' in text
  assert 'asm volatile("v_mul_f32 %0, 0x3e124925, %1"' in text
  kernel.compile_to_code_object(text, str(tmp_path / 'j3d.hsaco'))
