"""Randomly generated SODA programs, compiled by the HIP back end at run time
(hiprtc) and compared bit for bit with the CPU oracle: asymmetric and one-sided
windows, negative-only offsets (stages that run AHEAD of their parents in the
streaming pipeline), fan-in and fan-out DAGs, `let`s, casts, integer division,
several inputs, 2-D and 3-D, fused and per-stage kernels, every depth split; and
integer programs over the remaining operators of the grammar (% & | ^ comparisons
&& || unary - ~ !, non-decimal literals)."""
import json
import os
import tempfile

import numpy as np
import pytest

from soda_hip import frontend
from soda_hip.codegen import kernel
from soda_hip.codegen import spec as specmod
from soda_hip.runtime import host
from oracle import soda_oracle


pytestmark = pytest.mark.gpu


GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
with open(os.path.join(GOLDEN, 'random_programs.json')) as f:
  PROGRAMS = json.load(f)          # the texts the reference fixtures were made from
with open(os.path.join(GOLDEN, 'random_manifest.json')) as f:
  REFERENCE = {v['key']: (k, v) for k, v in json.load(f).items() if k.endswith('.npz')}
SCRATCH = tempfile.mkdtemp(prefix='soda_oracle_')      # not oracle/_build


def run_case(key, shape, rng):
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
  src, table = kernel.generate(spec)
  prog = host.open_program(source=src, spec=spec)
  fused = [k['depth'] for k in table if k['kind'] == 'fused']
  cases = []
  if key in REFERENCE:
    fixture, meta = REFERENCE[key]
    data = np.load(os.path.join(GOLDEN, fixture))
    ins = [np.ascontiguousarray(data['in_' + t['name']]) for t in spec['inputs']]
    cases.append((ins, data['out_out'], 'reference fixture'))
  orc = soda_oracle.Oracle(spec, build_dir=SCRATCH)
  inputs = []
  for t in spec['inputs']:
    dt = np.dtype(specmod.NUMPY_NAME[t['c_type']])
    if dt.kind == 'f':
      inputs.append((rng.random(shape, dtype=np.float32) + np.float32(0.5)).astype(dt))
    else:
      inputs.append(rng.integers(0, 200, size=shape).astype(dt))
  cases.append((inputs, orc.run(inputs, iterate=iterate)['out'], 'oracle'))
  for ins, want, what in cases:
    sl = orc.valid_slices(tuple(reversed(ins[0].shape)), iterate)
    for max_depth in sorted({0, 1, -1} if fused else {-1}):
      prog.set_max_depth(max_depth)
      got = prog.run_numpy(ins, iterate=iterate)[0]
      if want[sl].size == 0:
        continue
      same = np.array_equal(got[sl], want[sl], equal_nan=True)
      assert same, '%s vs %s, max_depth %d (fused depths %s)\n%s' % (
          key, what, max_depth, fused, text)
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


@pytest.mark.parametrize('seed', range(int(os.environ.get('SODA_RANDOM_DEEP_SEEDS', '16'))))
def test_random_program_many_iterations(seed):
  """The same generator with `iterate` 8..20: deep fused kernels, among them
  the wave-pipelined and packed forms the generator picks for chains that do not
  fit one wavefront's registers."""
  run_case('deep%d' % seed, (400, 700), np.random.default_rng(5000 + seed))
