"""Pins the CPU oracle (oracle/soda_oracle.py) to the reference: array-for-array
equality with the fixtures that tests/golden/make_golden.py produced by running
the reference's own emitted CPU loop nest."""
import hashlib
import json
import os

import numpy as np
import pytest

from soda_hip import frontend
from soda_hip.codegen import spec as specmod
from oracle import soda_oracle

from conftest import GOLDEN, SAMPLES

with open(os.path.join(GOLDEN, 'manifest.json')) as f:
  MANIFEST = json.load(f)
# hand-written programs (let, casts, ~lat, C calls) run by the reference:
# `make_golden.py --extra`
with open(os.path.join(GOLDEN, 'extra_manifest.json')) as f:
  EXTRA = json.load(f)
MANIFEST.update({k: v for k, v in EXTRA.items() if k.endswith('.npz')})

# the random programs of the GPU tests, run by the reference where its emitted
# loops compile: `make_golden.py --random`
with open(os.path.join(GOLDEN, 'random_manifest.json')) as f:
  RANDOM = {k: v for k, v in json.load(f).items() if k.endswith('.npz')}
with open(os.path.join(GOLDEN, 'random_programs.json')) as f:
  RANDOM_PROGRAMS = json.load(f)

_ORACLES = {}


def oracle_for(app):
  if app not in _ORACLES:
    path = os.path.join(SAMPLES, app + '.soda')
    if not os.path.exists(path):
      path = os.path.join(SAMPLES, 'extra', app + '.soda')
    st = frontend.load(path)
    _ORACLES[app] = soda_oracle.Oracle(specmod.spec_from_stencil(st))
  return _ORACLES[app]


@pytest.mark.parametrize('fixture', sorted(MANIFEST))
def test_oracle_matches_reference_fixture(fixture):
  meta = MANIFEST[fixture]
  app, it = meta['key'].split('.iter')
  it = int(it)
  data = np.load(os.path.join(GOLDEN, fixture))
  orc = oracle_for(app)
  spec = orc.spec
  dims = meta['dims']
  names = [t['name'] for t in spec['inputs']]
  if all('in_' + n in data for n in names):
    inputs = [np.ascontiguousarray(data['in_' + n]) for n in names]
  else:   # cfg1 fixture stores outputs only; inputs are the reference ramp
    inputs = soda_oracle.reference_init(spec, dims)
  got = orc.run(inputs, iterate=it, keep_all=True)
  expected = {k[4:]: data[k] for k in data.files if k.startswith('out_')}
  assert sorted(got) == sorted(expected)
  for name in expected:
    assert got[name].dtype == expected[name].dtype
    assert np.array_equal(got[name], expected[name], equal_nan=True), name
    assert hashlib.sha256(got[name].tobytes()).hexdigest() == \
        meta['sha256'][name]
  # ping-pong mode gives the same outputs on the region the reference defines
  pp = orc.run(inputs, iterate=it)
  sl = orc.valid_slices(dims, it)
  for name in spec['outputs']:
    assert np.array_equal(pp[name][sl], expected[name][sl], equal_nan=True)


@pytest.mark.parametrize('fixture', sorted(RANDOM))
def test_oracle_matches_reference_on_random_programs(fixture):
  meta = RANDOM[fixture]
  st = frontend.loads(RANDOM_PROGRAMS[meta['key']]['text'])
  spec = specmod.spec_from_stencil(st)
  orc = soda_oracle.Oracle(spec, build_dir=os.environ.get('TMPDIR', '/tmp'))
  data = np.load(os.path.join(GOLDEN, fixture))
  inputs = [np.ascontiguousarray(data['in_' + t['name']]) for t in spec['inputs']]
  got = orc.run(inputs, iterate=meta['iterate'], keep_all=True)
  for name in spec['outputs']:
    want = data['out_' + name]
    assert got[name].dtype == want.dtype
    assert np.array_equal(got[name], want, equal_nan=True), (fixture, name)
    assert hashlib.sha256(got[name].tobytes()).hexdigest() == meta['sha256'][name]


def test_reference_ramp_is_what_fixtures_used():
  data = np.load(os.path.join(GOLDEN, 'jacobi2d.iter2.37x29.ramp.npz'))
  orc = oracle_for('jacobi2d')
  (ramp,) = soda_oracle.reference_init(orc.spec, (37, 29))
  assert np.array_equal(ramp, data['in_t1'])
  data = np.load(os.path.join(GOLDEN, 'blur.iter1.37x29.ramp.npz'))
  (ramp,) = soda_oracle.reference_init(oracle_for('blur').spec, (37, 29))
  assert np.array_equal(ramp, data['in_input'])


def test_oracle_boxes_agree_with_frontend():
  for app in ('blur', 'jacobi2d', 'jacobi3d', 'denoise2d', 'sobel2d', 'heat3d'):
    st = frontend.load(os.path.join(SAMPLES, app + '.soda'))
    spec = specmod.spec_from_stencil(st)
    n = 5 if len(spec['inputs']) == len(spec['outputs']) else 1
    a = soda_oracle.iteration_boxes(spec, n)
    b = specmod.iteration_boxes(spec, n)
    c = st.iteration_boxes(n)
    for k in range(n):
      for name in a[k]:
        assert tuple(a[k][name][0]) == tuple(b[k][name][0]) == c[k][name].lo
        assert tuple(a[k][name][1]) == tuple(b[k][name][1]) == c[k][name].hi


def test_independent_numpy_jacobi2d():
  """A hand-written numpy formula, independent of the generated C."""
  rng = np.random.default_rng(7)
  a = rng.random((40, 50), dtype=np.float32)
  orc = oracle_for('jacobi2d')
  got = orc.run([a], iterate=3)['t0']
  cur = a
  for _ in range(3):
    nxt = np.zeros_like(cur)
    s = cur[2:, 1:-1] + cur[1:-1, 2:]
    s = s + cur[1:-1, 1:-1]
    s = s + cur[:-2, 1:-1]
    s = s + cur[1:-1, :-2]
    nxt[1:-1, 1:-1] = s * np.float32(0.2)
    cur = nxt
  assert np.array_equal(got[3:-3, 3:-3], cur[3:-3, 3:-3])


def test_independent_numpy_blur():
  rng = np.random.default_rng(8)
  a = rng.integers(0, 65536, size=(33, 47), dtype=np.uint16)
  got = oracle_for('blur').run([a])['blur_y']
  w = a.astype(np.int32)
  bx = ((w[:-2, :] + w[1:-1, :] + w[2:, :]) // 3).astype(np.uint16).astype(np.int32)
  by = ((bx[:, :-2] + bx[:, 1:-1] + bx[:, 2:]) // 3).astype(np.uint16)
  assert np.array_equal(got[:-2, :-2], by)


def test_valid_slices_of_an_empty_region_are_empty():
  """A grid smaller than the composed window has no defined cell; the slices
  must not wrap around (numpy reads a negative stop from the end)."""
  spec = specmod.spec_from_stencil(frontend.load(os.path.join(SAMPLES, 'jacobi2d.soda')))
  orc = soda_oracle.Oracle(spec)
  sl = orc.valid_slices((30, 12), iterate=8)     # 12 rows, 16 needed
  assert np.zeros((12, 30))[sl].size == 0
  sl = orc.valid_slices((30, 18), iterate=8)
  assert np.zeros((18, 30))[sl].shape == (2, 14)
