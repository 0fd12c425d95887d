"""Front end vs. the reference's own analysis (tests/golden/analysis.json, made
by tests/golden/make_golden.py from the real reference under python3.9)."""
import glob
import json
import os

import pytest

from soda_hip import frontend
from soda_hip.codegen import spec as specmod
from soda_hip.frontend import expr as ex
from soda_hip.frontend.errors import SemanticError, SodaSyntaxError
from soda_hip.frontend.types import c_type

from conftest import GOLDEN, SAMPLES

with open(os.path.join(GOLDEN, 'analysis.json')) as f:
  ANALYSIS = json.load(f)
# programs of tests/samples/extra (let, casts, ~lat, C calls): the reference's
# analysis of them, made by `make_golden.py --extra`
with open(os.path.join(GOLDEN, 'extra_analysis.json')) as f:
  EXTRA_ANALYSIS = json.load(f)


def describe(stencil):
  """Same shape as make_golden.analysis_of(), from OUR analysis."""
  stages = []
  for inst in stencil.instances():
    stage = inst['stage']
    rename = inst['rename']

    def load_text(ld, stage=stage, rename=rename):
      rel = [a - b for a, b in zip(ld.idx, stage.st_idx)]
      return '%s[%s]' % (rename[ld.name], ','.join(map(str, rel)))

    loads = {}
    for ld in stage.loads():
      loads.setdefault(rename[ld.name], []).append(list(ld.idx))
    stages.append(dict(
        name=inst['name'], haoda_type=stage.haoda_type,
        c_type=c_type(stage.haoda_type), st_idx=list(stage.st_idx),
        expr_str=ex.soda_text(ex.rename_loads(stage.expr, rename.get)),
        c_expr=ex.c_text(stage.expr, load_text, c_type),
        lets=[dict(name=name, haoda_type=let_type, c_type=c_type(let_type),
                   c_expr=ex.c_text(e, load_text, c_type))
              for name, let_type, e in stage.lets],
        loads=loads, loop_lo=list(inst['loop_lo']),
        loop_hi_margin=list(inst['loop_hi_margin']),
        is_output=inst['is_output']))
  return stages


# the random programs of the GPU tests (tests/random_programs.py), analysed by the
# reference: `make_golden.py --random`
with open(os.path.join(GOLDEN, 'random_analysis.json')) as f:
  RANDOM_ANALYSIS = json.load(f)
with open(os.path.join(GOLDEN, 'random_programs.json')) as f:
  RANDOM_PROGRAMS = json.load(f)


@pytest.mark.parametrize('key', sorted(ANALYSIS) + sorted(EXTRA_ANALYSIS) +
                         sorted(RANDOM_ANALYSIS))
def test_analysis_matches_reference(key):
  ref = ANALYSIS.get(key) or EXTRA_ANALYSIS.get(key) or RANDOM_ANALYSIS[key]
  if key in RANDOM_ANALYSIS:
    st = frontend.loads(RANDOM_PROGRAMS[key]['text'])
  else:
    app, it = key.split('.iter')
    folder = SAMPLES if key in ANALYSIS else os.path.join(SAMPLES, 'extra')
    st = frontend.load(os.path.join(folder, app + '.soda'), iterate=int(it))
  assert st.app_name == ref['app_name']
  assert st.dim == ref['dim']
  assert list(st.tile_size) == ref['tile_size']
  assert st.burst_width == ref['burst_width']
  assert st.unroll_factor == ref['unroll_factor']
  assert list(st.input_names) == ref['input_names']
  assert list(st.local_names) == ref['local_names']
  assert list(st.output_names) == ref['output_names']
  ours = describe(st)
  # execution order: inputs first, then instances, like chronological_tensors
  assert list(st.input_names) + [s['name'] for s in ours] == ref['chronological']
  assert len(ours) == len(ref['stages'])
  # A window that excludes the store point makes the reference print a negative
  # margin (`p<dims[0]--1`, which no compiler accepts); here every box contains the
  # cell itself (deliberate deviation), which also moves the boxes
  # of the stages downstream: loop bounds are compared for the other programs.
  one_sided = any(min(t['loop_lo'] + t['loop_hi_margin']) < 0 for t in ref['stages'])
  for mine, theirs in zip(ours, ref['stages']):
    for field in ('name', 'haoda_type', 'c_type', 'st_idx', 'expr_str', 'c_expr',
                  'lets', 'loop_lo', 'loop_hi_margin', 'is_output'):
      if one_sided and field in ('loop_lo', 'loop_hi_margin'):
        assert min(mine[field]) >= 0
        continue
      if field == 'is_output' and mine['name'] in st.output_names:
        # the reference calls a tensor an output when nothing reads it
        # (core.py:146); a declared output that feeds another stage is still an
        # output here (it is handed back to the caller all the same)
        assert mine[field]
        continue
      assert mine[field] == theirs[field], (field, mine['name'])
    # the reference sorts each parent's loads by linearised offset
    # (core.py:389-395); compare as multisets per parent
    assert {k: sorted(v) for k, v in mine['loads'].items()} == \
        {k: sorted(v) for k, v in theirs['loads'].items()}
  # STENCIL_DIM_n macros of the generated host (host.py:1183-1186)
  # ... which the reference takes from the FIRST output's window (host.py:1183-1186)
  first = [t for t in ours if t['name'] == st.output_names[0]][-1]
  lo, hi = first['loop_lo'], first['loop_hi_margin']
  for d in range(st.dim):
    assert one_sided or ref['macros']['STENCIL_DIM_%d' % d] == lo[d] + hi[d] + 1


def test_random_program_texts_are_the_committed_ones():
  """The fixtures under tests/golden/random.* were made from the texts in
  random_programs.json; the generator must still produce them (numpy does not
  promise one random stream for ever: if this fails, regenerate both with
  `python3 tests/random_programs.py` and `make_golden.py --random`)."""
  import random_programs
  now = {k: t for k, t, _, _ in random_programs.program_set()}
  assert sorted(now) == sorted(RANDOM_PROGRAMS)
  for key in now:
    assert now[key] == RANDOM_PROGRAMS[key]['text'], key


def test_specs_equal_those_built_from_the_reference_stencil():
  """tests/golden/plugin_specs.json: sha256 of the spec that the back end's plug-in
  entry (`backend.to_spec`) builds from the REFERENCE's own `core.Stencil`, for the
  hand-written extras and all random programs (written by
  `tools/check_reference_plugin.py --write` under python3.9, next to the reference).
  The own front end must arrive at the same spec, byte for byte: what the kernels
  are generated from does not depend on which of the two analysed the program."""
  import hashlib
  with open(os.path.join(GOLDEN, 'plugin_specs.json')) as f:
    digests = json.load(f)
  assert len(digests) >= 100
  for key, want in sorted(digests.items()):
    family, name = key.split('.', 1)
    if family == 'extra':
      st = frontend.load(os.path.join(SAMPLES, 'extra', name + '.soda'))
    else:
      st = frontend.loads(RANDOM_PROGRAMS[name]['text'])
    text = specmod.dumps(specmod.spec_from_stencil(st))
    assert hashlib.sha256(text.encode()).hexdigest() == want, key


def test_samples_all_parse():
  files = sorted(glob.glob(os.path.join(SAMPLES, '*.soda')))
  assert len(files) == 8
  for path in files:
    st = frontend.load(path)
    assert st.stages


def test_readme_example_any_key_order():
  # header keys may come in any order and between statement groups
  text = '''
    input float: a(16, *)   # comment
    iterate: 3
    output float: b(0, 0) = a(0, 0) + a(1, 1)
    unroll factor: 4
    kernel: demo
    burst width: 256
  '''
  st = frontend.loads(text)
  assert (st.app_name, st.iterate, st.unroll_factor, st.burst_width) == \
      ('demo', 3, 4, 256)
  assert st.radius() == ((0, 0), (1, 1))
  assert st.valid_margins()[-1] == ((0, 0), (3, 3))


@pytest.mark.parametrize('text,rendered', [
    ('a(0) + b(1) * 2', '(a(0) + (b(1) * 2))'),
    ('- - a(0)', 'a(0)'),
    ('-a(0)', '-a(0)'),
    ('!!a(0)', 'a(0)'),
    ('(a(0))', 'a(0)'),
    ('65535 - (a(0) * a(0) + b(0) * b(0))', '(65535 - (a(0) * a(0)) + (b(0) * b(0)))'),
    ('float(a(0)) / 3', '(float(a(0)) / 3)'),
    ('sqrt(a(0) + 1.0f)', 'sqrt((a(0) + 1.0f))'),
    # the wrapper's strip-all-then-add-one rule drops the && chain's own parens
    ('a(0) < b(0) && a(0) != 3 || x', '((a(0) < b(0)) && (a(0) != 3) || x)'),
    ('a(0) & 0xFFu | 1 ^ 2', '(a(0) & 0xFFu) | (1 ^ 2)'),
])
def test_expression_text(text, rendered):
  assert ex.soda_text(frontend.parse_expression(text)) == rendered


def test_cast_c_text():
  node = frontend.parse_expression('uint16(a(0, 0) + 1) * 2')
  text = ex.c_text(node, lambda ld: 'A', c_type)
  assert text == '(static_cast<uint16_t >(A + 1) * 2)'


def test_let_types():
  st = frontend.loads('''
    kernel: k
    burst width: 64
    unroll factor: 1
    iterate: 1
    input uint16: a(8, *)
    output uint16: float s = a(0, 0) + a(1, 0) t = s * 0.5f b(0, 0) = t + a(0, 1)
  ''')
  stage = st.stages['b']
  assert [(n, t) for n, t, _ in stage.lets] == [('s', 'float'), ('t', 'float')]
  assert st.radius() == ((0, 0), (1, 1))


@pytest.mark.parametrize('text', [
    'kernel: k\nburst width: 1\nunroll factor: 1\ninput float: a(4,*)\noutput float: b(0,0) = a(0,0)',  # no iterate
    'kernel: k\nkernel: j\nburst width: 1\nunroll factor: 1\niterate: 1\ninput float: a(4,*)\noutput float: b(0,0) = a(0,0)',
    'kernel: k\nburst width: 1\nunroll factor: 1\niterate: 1\ninput float: a(4,*)',  # no output
    'kernel: k\nburst width: 1\nunroll factor: 1\niterate: 1\ninput float: a(4,*)\noutput float: b(0,0) = a(0,0) +',
    'kernel: k\nburst width: 1\nunroll factor: 1\niterate: 1\ninput float: a(4,*)\noutput float: b(0,0) = a(0,x)',
    'kernel: k\nburst width: 1\nunroll factor: 1\niterate: 1\ninput float: a(4,*)\noutput float: b(0,0) = a(0,0)\ninput float: c',
    'kernel: k\nburst width: 1\nunroll factor: 1\niterate: 1\ninput flat: a(4,*)\noutput float: b(0,0) = a(0,0)',
    'kernel: k\nburst width: 1\nunroll factor: 1\niterate: 1\ninput float: a(4,*)\noutput float: b(0,0) = a(0,0) $ 2',
])
def test_syntax_errors(text):
  with pytest.raises(SodaSyntaxError):
    frontend.loads(text)


def test_semantic_errors():
  base = 'kernel: k\nburst width: 1\nunroll factor: 1\n'
  with pytest.raises(SemanticError, match='cannot iterate 0 times'):
    frontend.loads(base + 'iterate: 0\ninput float: a(4,*)\noutput float: b(0,0) = a(0,0)')
  with pytest.raises(SemanticError, match='number of input tensors'):
    frontend.loads(base + 'iterate: 2\ninput float: a(4,*)\ninput float: c(4,*)\n'
                   'output float: b(0,0) = a(0,0) + c(0,0)')
  with pytest.raises(SemanticError, match='same type'):
    frontend.loads(base + 'iterate: 2\ninput float: a(4,*)\noutput uint16: b(0,0) = a(0,0)')
  with pytest.raises(SemanticError, match="doesn't match previous"):
    frontend.loads(base + 'iterate: 1\ninput float: a(4,*)\ninput float: c\n'
                   'output float: b(0,0) = a(0,0) + c(0,0)')
  with pytest.raises(SemanticError, match='undefined tensor'):
    frontend.loads(base + 'iterate: 1\ninput float: a(4,*)\noutput float: b(0,0) = z(0,0)')
  # a variable must be a `let` made earlier in the statement (the reference dies
  # with a KeyError in propagate_type, core.py:116-120)
  with pytest.raises(SemanticError, match='undefined variable `lat`'):
    frontend.loads(base + 'iterate: 1\ninput float: a(4,*)\noutput float: b(0,0) = a(0,0) + lat')
  with pytest.raises(SemanticError, match='undefined variable `t`'):
    frontend.loads(base + 'iterate: 1\ninput float: a(4,*)\n'
                   'output float: s = t + a(0,0) t = a(1,0) b(0,0) = s + t')


def test_overrides_like_sodac():
  st = frontend.load(os.path.join(SAMPLES, 'jacobi2d.soda'), iterate=7,
                     tile_size=[2000], unroll_factor=8, burst_width=256)
  assert (st.iterate, st.tile_size, st.unroll_factor, st.burst_width) == \
      (7, (2000, 0), 8, 256)
  assert st.valid_margins()[-1] == ((7, 7), (7, 7))
  # tile size 0 keeps the DSL's value on that dimension (sodac:94-102)
  st = frontend.load(os.path.join(SAMPLES, 'jacobi3d.soda'), tile_size=[0, 64])
  assert st.tile_size == (32, 64, 0)
