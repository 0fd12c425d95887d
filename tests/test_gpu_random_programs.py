"""Randomly generated SODA programs, compiled by the HIP back end at run time
(hiprtc) and compared bit for bit with the CPU oracle: asymmetric and one-sided
windows, negative-only offsets (stages that run AHEAD of their parents in the
streaming pipeline), fan-in and fan-out DAGs, `let`s, casts, integer division,
several inputs, 2-D and 3-D, fused and per-stage kernels, every depth split; and
integer programs over the remaining operators of the grammar (% & | ^ comparisons
&& || unary - ~ !, non-decimal literals)."""
import json
import os

import numpy as np
import pytest

from soda_hip import frontend
from soda_hip.codegen import kernel
from soda_hip.codegen import spec as specmod
from soda_hip.runtime import host
from oracle import soda_oracle  # noqa: F401

import gpu_util


pytestmark = pytest.mark.gpu


GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
with open(os.path.join(GOLDEN, 'random_programs.json')) as f:
  PROGRAMS = json.load(f)          # the texts the reference fixtures were made from
with open(os.path.join(GOLDEN, 'random_manifest.json')) as f:
  REFERENCE = {v['key']: (k, v) for k, v in json.load(f).items() if k.endswith('.npz')}


def run_case(key, shape, rng, **gen):
  """One random program through the HIP back end (hiprtc), every depth split:
  (a) on the inputs of the REFERENCE's own run of this program, against the
  reference's result (tests/golden/random.<key>.npz, made by `make_golden.py
  --random`; 28 of the 56 programs - the others have a window that excludes the
  store point, for which the reference's loops leave the arrays or do not compile);
  (b) on a larger grid against the CPU oracle."""
  entry = PROGRAMS[key]
  text, iterate = entry['text'], entry['iterate']
  stencil = frontend.loads(text)
  spec = specmod.spec_from_stencil(stencil)
  src, table = kernel.generate(spec, **gen)
  prog = host.open_program(source=src, spec=spec)
  fused = [k['depth'] for k in table if k['kind'] == 'fused']
  cases = []
  if key in REFERENCE:
    fixture, meta = REFERENCE[key]
    data = np.load(os.path.join(GOLDEN, fixture))
    ins = [np.ascontiguousarray(data['in_' + t['name']]) for t in spec['inputs']]
    cases.append((ins, {n: data['out_' + n] for n in spec['outputs']},
                  'reference fixture'))
  orc = gpu_util.make_oracle(spec)
  inputs = []
  for t in spec['inputs']:
    dt = np.dtype(specmod.NUMPY_NAME[t['c_type']])
    if dt.kind == 'f':
      inputs.append((rng.random(shape, dtype=np.float32) + np.float32(0.5)).astype(dt))
    else:
      inputs.append(rng.integers(0, 200, size=shape).astype(dt))
  cases.append((inputs, orc.run(inputs, iterate=iterate), 'oracle'))
  for ins, wants, what in cases:
    sl = orc.valid_slices(tuple(reversed(ins[0].shape)), iterate)
    for max_depth in sorted({0, 1, -1} if fused else {-1}):
      prog.set_max_depth(max_depth)
      gots = prog.run_numpy(ins, iterate=iterate)
      for name, got in zip(spec['outputs'], gots):
        want = wants[name]
        if len(spec['outputs']) > 1:
          # every output has its own box (host.py:1082-1091).  The reference's arrays
          # are zero outside it and so is what the host-buffer protocol hands back;
          # the oracle ping-pongs and may hold earlier iterations there
          blo, bhi = specmod.iteration_boxes(spec, iterate)[-1][name]
          own = tuple(slice(-blo[d], max(-blo[d], got.shape[::-1][d] - bhi[d]))
                      for d in reversed(range(spec['dim'])))
          same = np.array_equal(got[own], want[own], equal_nan=True)
          assert what != 'oracle' or got[own].size > 0, (key, name)
          if what == 'reference fixture':
            same = same and np.array_equal(got, want, equal_nan=True)
        elif want[sl].size == 0:
          continue
        else:
          same = np.array_equal(got[sl], want[sl], equal_nan=True)
        assert same, '%s `%s` vs %s, max_depth %d (fused depths %s)\n%s' % (
            key, name, what, max_depth, fused, text)
  prog.close()
  prog.blob.unload()
  return table


@pytest.mark.parametrize('seed', range(int(os.environ.get('SODA_RANDOM_SEEDS', '40'))))
def test_random_program(seed):
  rng = np.random.default_rng(1000 + seed)
  key = 'plain%d' % seed
  run_case(key, (41, 333) if PROGRAMS[key]['dim'] == 2 else (19, 23, 150), rng)


@pytest.mark.parametrize('seed', range(16))
def test_random_operator_program(seed):
  """Integer programs over % & | ^, comparisons, && ||, unary - ~ !, hexadecimal /
  octal / suffixed literals and casts between widths (tests/random_programs.py:
  operator_program), all 16 with a fixture from the reference's own CPU loops."""
  rng = np.random.default_rng(9000 + seed)
  key = 'ops%d' % seed
  assert key in REFERENCE
  run_case(key, (41, 333) if PROGRAMS[key]['dim'] == 2 else (19, 23, 150), rng)


@pytest.mark.parametrize('seed', range(16))
def test_random_structure_program(seed):
  """Up to three inputs, windows reaching 4 cells, up to seven stages, one or two
  outputs (iterated when inputs and outputs pair up), 2-D and 3-D
  (tests/random_programs.py: structure_program), all 16 with a reference fixture."""
  rng = np.random.default_rng(12000 + seed)
  key = 'struct%d' % seed
  assert key in REFERENCE
  entry = PROGRAMS[key]
  shape = [61, 333] if entry['dim'] == 2 else [24, 27, 150]
  # deep iteration over wide windows: grow the grid until every output keeps at
  # least 20 cells per dimension (two of the programs leave nothing on the
  # reference's small fixture grid: there the comparison is of all-zero arrays)
  spec = specmod.spec_from_stencil(frontend.loads(entry['text']))
  boxes = specmod.iteration_boxes(spec, entry['iterate'])[-1]
  for name in spec['outputs']:
    lo, hi = boxes[name]
    for d in range(spec['dim']):
      axis = spec['dim'] - 1 - d
      shape[axis] = max(shape[axis], hi[d] - lo[d] + 20)
  run_case(key, tuple(shape), rng)


@pytest.mark.parametrize('seed', range(12))
def test_random_3d_chain_program(seed):
  """3-D iteration chains with diagonal reads, 4..13 iterations: the single-wave,
  wave-pipelined and block-form deep 3-D kernels as the generator picks them
  (tests/random_programs.py: cube_program), all 12 with a reference fixture."""
  rng = np.random.default_rng(15000 + seed)
  key = 'cube%d' % seed
  assert key in REFERENCE
  table = run_case(key, (70, 90, 200), rng)
  assert any(k['kind'] == 'fused' for k in table)
  if any(k.get('stack') and k['depth'] == 4 for k in table) and (
      seed % 4 == 0 or os.environ.get('SODA_TEST_ALL_FORMS')):
    # the block form ALONE (the scheduler may have preferred the wave-pipelined
    # kernel above), plain and with packed pair-rows where the program allows
    # (every fourth program: each variant is another hiprtc compile; all of them with
    # SODA_TEST_ALL_FORMS=1)
    table = run_case(key, (70, 90, 200), np.random.default_rng(15100 + seed),
                     deep3d='blk')
    assert [bool(k.get('stack')) for k in table if k['depth'] == 4] == [True]
    if PROGRAMS[key]['text'].count('float') and 'local' not in PROGRAMS[key]['text']:
      run_case(key, (70, 90, 200), np.random.default_rng(15200 + seed),
               deep3d='blk', blk_pairs=1)


@pytest.mark.parametrize('seed', range(int(os.environ.get('SODA_RANDOM_DEEP_SEEDS', '16'))))
def test_random_program_many_iterations(seed):
  """The same generator with `iterate` 8..20: deep fused kernels, among them
  the wave-pipelined and packed forms the generator picks for chains that do not
  fit one wavefront's registers."""
  run_case('deep%d' % seed, (400, 700), np.random.default_rng(5000 + seed))


@pytest.mark.parametrize('name', ['ops7222', 'st7327'])
def test_programs_the_round4_fuzzing_found(name):
  """The two 3-D programs whose block-form kernels did not compile before round 4
  (tests/test_codegen.py: ROUND4_FUZZ_FINDS - an int32 stage's edge rows through a
  uint32-typed LDS array; two stages reading one tensor's edge rows ahead), every depth
  split against the oracle (no reference fixture: they come from fresh fuzzing seeds)."""
  from test_codegen import ROUND4_FUZZ_FINDS
  PROGRAMS[name] = dict(text=ROUND4_FUZZ_FINDS[name], iterate=2, dim=3)
  table = run_case(name, (40, 80, 150), np.random.default_rng(16000))
  assert any(k.get('stack') for k in table)
