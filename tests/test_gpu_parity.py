"""Parity of the HIP path against the CPU oracle and the reference fixtures.
Everything here runs on a real MI355X and goes through libsoda_hip.so's C ABI.

Bar: bit-exact for the integer programs AND for the float programs (the kernels
evaluate the reference's expression text in the reference's order with FP
contraction off, so there is nothing to round differently); the reference's own
comparator (relative 1e-5, host.py:1124-1132) is checked through `<app>_test`.
"""
import json
import os

import numpy as np
import pytest

from soda_hip.runtime import host
from oracle import soda_oracle

import gpu_util
from conftest import GOLDEN

pytestmark = pytest.mark.gpu

with open(os.path.join(GOLDEN, 'manifest.json')) as f:
  MANIFEST = json.load(f)
# + the hand-written let / cast / ~lat / C-call programs, run by the reference
with open(os.path.join(GOLDEN, 'extra_manifest.json')) as f:
  MANIFEST.update({k: v for k, v in json.load(f).items() if k.endswith('.npz')})

APPS = ('blur', 'jacobi2d', 'jacobi3d', 'seidel2d', 'heat3d', 'sobel2d',
        'denoise2d', 'denoise3d')
# every fixture is held to bit-exactness except programs that call libm
# approximations (exp, log, sin, pow ...: tests/samples/extra/transcend.soda)
LIBM_TOLERANCE = ('transcend',)
_PROGRAMS = {}
_ORACLES = {}


def program(app):
  if app not in _PROGRAMS:
    _PROGRAMS[app] = gpu_util.open_prebuilt(app)
  return _PROGRAMS[app]


def oracle(app):
  if app not in _ORACLES:
    _ORACLES[app] = gpu_util.make_oracle(gpu_util.load_spec(app))
  return _ORACLES[app]


def check(app, inputs, iterate, max_depth=0):
  prog = program(app)
  prog.set_max_depth(max_depth)
  got = prog.run_numpy(inputs, iterate=iterate)
  orc = oracle(app)
  want = orc.run(inputs, iterate=iterate)
  dims = tuple(reversed(inputs[0].shape))
  sl = orc.valid_slices(dims, iterate)
  for name, g in zip(prog.spec['outputs'], got):
    w = want[name]
    assert g[sl].size > 0
    bad = np.argwhere(g[sl] != w[sl])
    assert bad.size == 0, '%s %s it=%d depth=%d: %d cells differ, first at %s' % (
        app, name, iterate, max_depth, len(bad), bad[0])
  prog.set_max_depth(0)


def test_device_is_gfx950():
  assert host.device_count() >= 1
  info = host.device_info(0)
  assert info['arch'].startswith('gfx950'), info


@pytest.mark.parametrize('fixture', sorted(
    k for k in MANIFEST if k.endswith('.random.npz') or k.endswith('.ramp.npz')))
def test_fixture(fixture):
  """HIP result == what the reference's own CPU loop nest produced."""
  meta = MANIFEST[fixture]
  app, it = meta['key'].split('.iter')
  it = int(it)
  data = np.load(os.path.join(GOLDEN, fixture))
  prog = program(app)
  names = [t['name'] for t in prog.spec['inputs']]
  if all('in_' + n in data for n in names):
    inputs = [np.ascontiguousarray(data['in_' + n]) for n in names]
  else:
    inputs = host.reference_init(prog.spec, meta['dims'])
  got = prog.run_numpy(inputs, iterate=it)
  sl = oracle(app).valid_slices(tuple(meta['dims']), it)
  for name, g in zip(prog.spec['outputs'], got):
    want = data['out_' + name]
    if app in LIBM_TOLERANCE:
      # transcendental C calls: the device's libm against the host's; the bar is the
      # reference comparator's own (host.py:1124-1132): (gpu-cpu)^2/cpu^2 <= (1e-5)^2
      g64, w64 = g.astype(np.float64), want.astype(np.float64)
      assert np.all((g64[sl] - w64[sl]) ** 2 <= (1e-5 * w64[sl]) ** 2), (fixture, name)
      assert np.array_equal(g == 0, want == 0)      # and the same cells left alone
      continue
    assert np.array_equal(g[sl], want[sl], equal_nan=True), (fixture, name)
    # and the cells the reference never defines stay zero in the host protocol
    assert np.array_equal(g, want, equal_nan=True), (fixture, name)


@pytest.mark.parametrize('app', APPS)
def test_random_vs_oracle_default_iterate(app):
  spec = gpu_util.load_spec(app)
  shape = (203, 517) if spec['dim'] == 2 else (37, 45, 70)
  inputs = gpu_util.random_inputs(spec, shape, small_ints=(app == 'sobel2d'))
  check(app, inputs, spec['iterate'])


@pytest.mark.parametrize('iterate,max_depth', [
    (1, 0), (2, 0), (3, 0), (5, 0), (8, 0), (16, 0), (21, 0), (37, 0), (50, 0),
    (20, 0), (24, 0), (44, 0), (100, 0), (45, 20), (40, 16),
    (7, 1), (7, 2), (9, 4), (20, 8), (33, 12), (6, -1)])
def test_jacobi2d_iterations_and_depths(iterate, max_depth):
  """Temporal blocking: any split of `iterate` into fused depths, and the
  per-stage kernels (max_depth -1), give the same bits."""
  spec = gpu_util.load_spec('jacobi2d')
  inputs = gpu_util.random_inputs(spec, (300, 1100))
  check('jacobi2d', inputs, iterate, max_depth)


# The generator options behind the SHIPPED wave-pipelined kernels (what
# kernel.generate picks by itself, spelled out) ...
SHIPPED_FORMS = [
    # depth 12/16: wide strips, 12-slot ring, four workgroups per CU
    ('jacobi2d', dict(wave_groups=4, pairs=2, vgpr_budget=250, ring=12, max_period=12,
                      waves_per_eu=4)),
    # depth 20/24: wide strips, 6-slot ring, three workgroups per CU
    ('jacobi2d', dict(wave_groups=4, pairs=2, vgpr_budget=250, ring=6, waves_per_eu=3)),
    ('seidel2d', dict(wave_groups=4, pairs=2, vgpr_budget=250, ring=6)),
    ('blur', dict(wave_groups=4, vgpr_budget=200, ring=6)),
]
# ... and the forms that were measured and not shipped (docs/DESIGN_HISTORY.md 4.1a).  They
# stay correct: ALWAYS_TESTED_EXPERIMENTAL of them run in every test session (the
# same ones on any day: fixed test ids), all of them with SODA_TEST_ALL_FORMS=1.
EXPERIMENTAL_FORMS = [
    ('jacobi2d', dict(wave_groups=4)),
    ('jacobi2d', dict(wave_groups=4, pairs=1, vgpr_budget=250)),
    ('jacobi2d', dict(wave_groups=2, pairs=1, vgpr_budget=250)),
    ('jacobi2d', dict(wave_groups=4, pairs=1, vgpr_budget=250, ring=6)),
    ('jacobi2d', dict(wave_groups=4, vgpr_budget=250, ring=4)),
    ('seidel2d', dict(wave_groups=4, pairs=1, vgpr_budget=250, ring=6, waves_per_eu=4)),
    ('seidel2d', dict(wave_groups=4, pairs=1, vgpr_budget=250)),
    ('blur', dict(wave_groups=4, vgpr_budget=200)),
    ('seidel2d', dict(wave_groups=4, pairs=2, vgpr_budget=250, ring=12, max_period=12)),
    ('seidel2d', dict(wave_groups=2, pairs=2, vgpr_budget=250, ring=12, max_period=12))]


# (scalar pipelined; two packed strips with scalar DPP adds: the two code paths of the
# generator that no shipped form goes through)
ALWAYS_TESTED_EXPERIMENTAL = (0, 3)


def _tested_forms():
  if os.environ.get('SODA_TEST_ALL_FORMS'):
    return SHIPPED_FORMS + EXPERIMENTAL_FORMS
  return SHIPPED_FORMS + [EXPERIMENTAL_FORMS[i] for i in ALWAYS_TESTED_EXPERIMENTAL]


@pytest.mark.parametrize('app,options', _tested_forms())
def test_wave_pipelined_generator_forms(app, options):
  """The wave-pipelined forms of the fused 2-D kernel (wavefront pipeline through
  LDS; strips packed into v_pk_*_f32 operands; LDS input ring) produce the
  oracle's bits.  One code object per form (built for up to 31 iterations: depths
  1..12), run at several iteration counts and ragged shapes."""
  from soda_hip.codegen import kernel
  from soda_hip.runtime import host
  spec = gpu_util.load_spec(app, iterate=31)
  text, table = kernel.generate(spec, **options)
  assert any(k.get('groups') for k in table), [k['name'] for k in table]
  prog = host.open_program(source=text, spec=spec)
  orc = gpu_util.make_oracle(spec)
  try:
    for iterate, shape in ((8, (130, 1300)), (19, (300, 2100)), (9, (64, 64)),
                           (13, (90, 256)), (12, (70, 257)), (21, (50, 511)),
                           (31, (80, 512)), (10, (40, 513)), (24, (75, 995))):
      inputs = gpu_util.random_inputs(spec, shape)
      got = prog.run_numpy(inputs, iterate=iterate)[0]
      want = orc.run(inputs, iterate=iterate)[spec['outputs'][0]]
      sl = orc.valid_slices(tuple(reversed(shape)), iterate)
      assert np.array_equal(got[sl], want[sl], equal_nan=True), (app, options, iterate)
  finally:
    prog.close()
    prog.blob.unload()


@pytest.mark.parametrize('app,options,shape,iterate', [
    ('blur', dict(nontemporal=2), (90, 1100), 1),
    ('jacobi2d', dict(nontemporal=3), (77, 2100), 5),
    ('sobel2d', dict(nontemporal=2), (130, 515), 1),
    ('denoise3d', dict(nt=2), (40, 70, 300), 1),
    ('jacobi3d', dict(nt=2, wp_nt=2, blk_nt=3), (30, 80, 260), 7),
    ('heat3d', dict(nt=2, wp_nt=4, blk_nt=2), (33, 70, 140), 6)])
def test_nontemporal_store_paths(app, options, shape, iterate):
  """Stores (and loads) around the caches.  The shipped kernels take these paths
  only for launches whose box is larger than the Infinity Cache (the full-size tests
  below); forced on here, on small ragged arrays, they give the same bits."""
  from soda_hip.codegen import kernel
  spec = gpu_util.load_spec(app, iterate=iterate)
  text, table = kernel.generate(spec, **options)
  assert 'nontemporal' in text or ', 2); }' in text
  prog = host.open_program(source=text, spec=spec)
  try:
    inputs = gpu_util.random_inputs(spec, shape)
    got = prog.run_numpy(inputs, iterate=iterate)
    orc = gpu_util.make_oracle(spec)
    want = orc.run(inputs, iterate=iterate)
    sl = orc.valid_slices(tuple(reversed(shape)), iterate)
    for j, name in enumerate(spec['outputs']):
      assert np.array_equal(got[j][sl], want[name][sl], equal_nan=True), (app, options)
  finally:
    prog.close()
    prog.blob.unload()


@pytest.mark.parametrize('shape', [
    (64, 64), (65, 257), (100, 241), (257, 1021), (41, 2049), (600, 97)])
def test_jacobi2d_ragged_shapes(shape):
  """Widths that are not multiples of the vector width / strip width, heights
  that are not multiples of the chunk."""
  spec = gpu_util.load_spec('jacobi2d')
  check('jacobi2d', gpu_util.random_inputs(spec, shape), 5)


@pytest.mark.parametrize('app,iterate', [('blur', 1), ('blur', 3), ('seidel2d', 6),
                                         ('sobel2d', 2)])
@pytest.mark.parametrize('max_depth', [0, -1])
def test_multistage_fused_vs_stage_kernels(app, iterate, max_depth):
  spec = gpu_util.load_spec(app)
  inputs = gpu_util.random_inputs(spec, (130, 1300), small_ints=(app == 'sobel2d'))
  check(app, inputs, iterate, max_depth)


@pytest.mark.parametrize('app', ['jacobi3d', 'heat3d'])
@pytest.mark.parametrize('shape,iterate,max_depth', [
    ((70, 45, 37), 1, 0), ((70, 45, 37), 2, 0), ((40, 61, 130), 5, 0),
    ((33, 29, 300), 4, 1), ((64, 64, 64), 3, -1), ((50, 33, 257), 6, 0)])
def test_3d_fused_plane_streaming(app, shape, iterate, max_depth):
  """2.5-D fused kernels (depth 1 and 2) against the oracle: tiles that overhang
  the grid in x and y, z-chunks with remainders, odd iteration counts (2+2+1)."""
  spec = gpu_util.load_spec(app)
  check(app, gpu_util.random_inputs(spec, shape), iterate, max_depth)


@pytest.mark.parametrize('app', ['jacobi3d', 'heat3d'])
@pytest.mark.parametrize('shape,iterate', [
    ((20, 32, 64), 4), ((30, 33, 65), 4), ((24, 35, 67), 8), ((40, 100, 200), 9),
    ((44, 61, 130), 13), ((12, 90, 64), 5), ((300, 40, 70), 4)])
def test_3d_wave_pipelined_depth_4(app, shape, iterate):
  """The depth-4 wave-pipelined 3-D kernel (one level per wavefront, plane tiles
  through LDS, 64x32 tiles made of two row blocks joined by v_permlane32_swap,
  overhanging tiles moved inside the array): exactly one tile, one cell more than
  a tile, ragged edges in x and y, z-chunks with remainders."""
  prog = program(app)
  names = [k['name'] for k in prog.kernels]
  assert app + '_fused_k4' in names, names
  spec = gpu_util.load_spec(app)
  inputs = gpu_util.random_inputs(spec, shape)
  check(app, inputs, iterate)
  timing = prog.run_numpy(inputs, iterate=iterate, timed=True)[1]
  assert timing['max_depth'] == 4, timing


@pytest.mark.parametrize('options', [dict(), dict(wp_pairs=0)] + (
    [dict(wp_prefetch=1), dict(wp_pairs=1, wp_waves_per_eu=3),
     dict(wp_loader=1, wp_waves_per_eu=3), dict(wp_split=1), dict(wp_pairs=1, wp_rows=12)]
    if os.environ.get('SODA_TEST_ALL_FORMS') else []))
def test_3d_wave_pipelined_forms_on_heat3d(options):
  """The depth-4 generator on heat3d (FMA-sensitive expression) in its optional
  forms: packed pair-rows (the default for float programs) and scalar,
  LDS-direct loader wavefront, one row block per wavefront, register prefetch."""
  from soda_hip.codegen import kernel
  # (one code object per form: a blob serves any run-time iteration count)
  spec = gpu_util.load_spec('heat3d', iterate=9)
  text, table = kernel.generate(spec, depths=[2, 4], **options)
  assert any(k['depth'] == 4 for k in table)
  prog = host.open_program(source=text, spec=spec)
  orc = gpu_util.make_oracle(spec)
  try:
    for shape, iterate in (((30, 45, 70), 4), ((24, 64, 131), 9)):
      inputs = gpu_util.random_inputs(spec, shape)
      got, timing = prog.run_numpy(inputs, iterate=iterate, timed=True)
      assert timing['max_depth'] == 4
      want = orc.run(inputs, iterate=iterate)[spec['outputs'][0]]
      sl = orc.valid_slices(tuple(reversed(shape)), iterate)
      assert np.array_equal(got[0][sl], want[sl], equal_nan=True), (options, shape)
  finally:
    prog.close()
    prog.blob.unload()


@pytest.mark.parametrize('app,options', [
    ('jacobi3d', dict(deep3d='blk')),                      # 8 bands, LDS ring of 2 planes
    ('jacobi3d', dict(deep3d='blk', blk_prefetch=1)),      # ... one plane ahead in registers
    ('heat3d', dict(deep3d='blk')),                        # ring + packed pair-rows
    ('heat3d', dict(deep3d='blk', blk_pairs=0)),           # ring, scalar arithmetic
    ('jacobi3d', dict(deep3d='blk', blk_mask_loads=0)),    # unmasked global_load_lds ring
    ('jacobi3d', dict(deep3d='blk', blk_lean_fill=0)),     # every level at every step
    # row segments stored in whole 64-byte pieces (shipped: only in launches beyond the
    # Infinity Cache - the full-size tests; forced on here), and never
    ('heat3d', dict(deep3d='blk', blk_wide_stores=1, blk_nt=2)),
    ('jacobi3d', dict(deep3d='blk', blk_wide_stores=0)),
    # packed pair-rows for a light program; four bands per workgroup
    ('jacobi3d', dict(deep3d='blk', blk_pairs=1)),
    ('heat3d', dict(deep3d='blk', blk_stack=4, blk_prefetch=0, blk_pairs=1, blk_ring=2))])
def test_3d_block_form(app, options):
  """The block form of the depth-4 3-D kernel (kernel_stream3d_blk: all levels in
  every wavefront, bands of rows per wavefront, edge rows through LDS, tiles
  moved inside the array, raw-buffer loads and stores) ALONE - no wave-pipelined
  kernel next to it - on ragged shapes at and above its smallest array."""
  from soda_hip.codegen import kernel
  # (one code object per form: a blob serves any run-time iteration count)
  spec = gpu_util.load_spec(app, iterate=13)
  text, table = kernel.generate(spec, depths=[2, 4], **options)
  deep = [k for k in table if k['depth'] == 4]
  assert [bool(k.get('stack')) for k in deep] == [True], [k['name'] for k in deep]
  prog = host.open_program(source=text, spec=spec)
  orc = gpu_util.make_oracle(spec)
  try:
    for shape, iterate in (((30, 64, 128), 4), ((45, 131, 140), 9), ((150, 70, 257), 13)):
      if shape[1] < deep[0]['min_extent'][1] or shape[2] < deep[0]['min_extent'][0]:
        continue
      inputs = gpu_util.random_inputs(spec, shape)
      got, timing = prog.run_numpy(inputs, iterate=iterate, timed=True)
      assert timing['max_depth'] == 4
      want = orc.run(inputs, iterate=iterate)[spec['outputs'][0]]
      sl = orc.valid_slices(tuple(reversed(shape)), iterate)
      assert want[sl].size > 0
      assert np.array_equal(got[0][sl], want[sl], equal_nan=True), (options, shape)
  finally:
    prog.close()
    prog.blob.unload()


def test_small_arrays_skip_kernels_without_a_guarded_path():
  """Arrays smaller than one 64x32 tile are served by the shallower fused
  kernels (soda_hip_kernel.min_extent), with the same results."""
  spec = gpu_util.load_spec('jacobi3d')
  prog = program('jacobi3d')
  for shape in ((40, 31, 100), (40, 50, 63), (16, 20, 20)):
    inputs = gpu_util.random_inputs(spec, shape)
    check('jacobi3d', inputs, 4)
    assert prog.run_numpy(inputs, iterate=4, timed=True)[1]['max_depth'] == 2


def test_empty_valid_region_is_not_an_error():
  """iterate so large that nothing is left: nothing launched, zeros back."""
  spec = gpu_util.load_spec('jacobi2d')
  inputs = gpu_util.random_inputs(spec, (20, 30))
  got = program('jacobi2d').run_numpy(inputs, iterate=10)
  assert not got[0].any()


def test_cfg1_blur_2000x100_sha256():
  """BASELINE config 1 through the GPU path: hash of the whole result array."""
  import hashlib
  meta = MANIFEST['blur.iter1.2000x100.ramp.npz']
  prog = program('blur')
  inputs = host.reference_init(prog.spec, [2000, 100])
  got = prog.run_numpy(inputs, iterate=1)[0]
  assert hashlib.sha256(got.tobytes()).hexdigest() == meta['sha256']['blur_y']


def test_app_test_entry_point(capfd):
  """The generated `<app>_test(blob, dims)`: PASS line, zero mismatches, the two
  timing lines of the reference (host.py:796-800)."""
  spec = gpu_util.load_spec('jacobi2d')
  blob = os.path.join(gpu_util.BLOBS, 'jacobi2d.hsaco')
  errors = host.app_test(spec, blob, [500, 300, 0, 0])
  out, err = capfd.readouterr()
  assert errors == 0
  assert 'INFO: PASS!' in err
  assert 'Kernel execution time:' in out and 'Kernel throughput:' in out
  spec = gpu_util.load_spec('blur')
  assert host.app_test(spec, os.path.join(gpu_util.BLOBS, 'blur.hsaco'),
                       [2000, 100, 0, 0]) == 0


def test_jit_path_matches_prebuilt():
  """Kernel text compiled at run time by hiprtc == the hipcc-built blob."""
  prog = gpu_util.open_jit('jacobi2d', iterate=4)
  spec = prog.spec
  inputs = gpu_util.random_inputs(spec, (150, 700))
  got = prog.run_numpy(inputs, iterate=4)[0]
  ref = program('jacobi2d').run_numpy(inputs, iterate=4)[0]
  assert np.array_equal(got, ref)
  prog.close()
  # the C-call wrappers (device libm entry points declared in the kernel text) link
  # under hiprtc as under hipcc, with the same bits
  for app in ('rounding', 'transcend'):
    prog = gpu_util.open_jit(app)
    inputs = gpu_util.random_inputs(prog.spec, (90, 300))
    got = prog.run_numpy(inputs, iterate=1)[0]
    ref = program(app).run_numpy(inputs, iterate=1)[0]
    assert got.std() > 0 and np.array_equal(got, ref), app
    prog.close()


def test_wrong_blob_is_refused():
  from soda_hip.runtime import capi
  spec = gpu_util.load_spec('blur')
  with pytest.raises(capi.SodaHipError) as e:
    host.open_program(blob=os.path.join(gpu_util.BLOBS, 'jacobi2d.hsaco'), spec=spec)
  assert e.value.code == -103


def test_full_size_properties_jacobi2d_8192():
  """BASELINE config 2 size (8192^2, 100 iterations): size-independent checks.
  (a) a linear ramp is a fixed point of the averaging stencil up to rounding:
      the reference's own test input, checked with the reference's comparator;
  (b) depth-16 blocking and depth-1 launches agree bit for bit;
  (c) the oracle on the whole grid: every cell of the valid box."""
  prog = program('jacobi2d')
  n, it = 8192, 100
  rng = np.random.default_rng(5)
  a = rng.random((n, n), dtype=np.float32)
  prog.set_max_depth(0)
  deep = prog.run_numpy([a], iterate=it)[0]
  prog.set_max_depth(1)
  flat = prog.run_numpy([a], iterate=it)[0]
  prog.set_max_depth(0)
  assert np.array_equal(deep, flat)
  assert deep[it:-it, it:-it].std() > 0
  # (c) every cell of the valid box against the oracle
  want = oracle('jacobi2d').run([a], iterate=it)['t0']
  sl = oracle('jacobi2d').valid_slices((n, n), it)
  assert want[sl].shape == (n - 2 * it, n - 2 * it)
  assert np.array_equal(deep[sl], want[sl])
  del want
  # (a)
  ramp = host.reference_init(prog.spec, [n, n])
  out = prog.run_numpy(ramp, iterate=it)[0]
  inner = (slice(it, n - it), slice(it, n - it))
  rel = np.abs(out[inner] - ramp[0][inner]) / np.abs(ramp[0][inner])
  assert rel.max() < 1e-4


@pytest.mark.parametrize('app,n,iterate,max_depth', [
    ('jacobi2d', 8192, 7, 4),       # 4 + 2 + 1: all three shallow kernels, 512 MiB in + out
    ('seidel2d', 8192, 3, 2),
    ('sobel2d', 12288, 1, 0),       # uint16: 576 MiB in + out
    ('blur', 12288, 3, 2),
    ('jacobi3d', 448, 3, 2),        # 3-D block kernels of depth 2 and 1: 686 MiB
])
def test_streaming_launches_of_the_shallow_kernels_against_the_oracle(app, n, iterate,
                                                                     max_depth):
  """Boxes beyond the Infinity Cache take the launch rule of soda_hip_kernel.stream_chunk
  (short chunks, the measured cap) and, in the depth <= 2 kernels, the row loop that does
  not load its pipeline-flush rows: every cell of the valid box against the oracle, for
  every shallow kernel that has such a record."""
  prog = program(app)
  spec = prog.spec
  dims = [n] * spec['dim']
  rng = np.random.default_rng(n + iterate)
  dt = prog.in_dtypes[0]
  shape = tuple(reversed(dims))
  a = rng.random(shape, dtype=np.float32) if dt.kind == 'f' else \
      rng.integers(0, 2048, size=shape).astype(dt)
  prog.set_max_depth(max_depth)
  try:
    got, timing = prog.run_numpy([a], iterate=iterate, timed=True)
  finally:
    prog.set_max_depth(0)
  chunked = [k['name'] for k in prog.kernels if k.get('stream_chunk', 0) > 0]
  assert chunked, 'no kernel of %s carries a measured stream chunk' % app
  orc = oracle(app)
  want = orc.run([a], iterate=iterate)[spec['outputs'][0]]
  sl = orc.valid_slices(tuple(dims), iterate)
  assert want[sl].size > 0 and np.array_equal(got[0][sl], want[sl])


def test_full_size_cfg4_jacobi2d_16384_x1000():
  """BASELINE config 4, the headline workload, at full size against the oracle, EVERY
  cell of the valid box (the OpenMP oracle at the box's CPU quota takes ~20 s for the
  2.7e11 updates; conftest sets the team size - at 256 threads on 16 CPUs it took
  ten times that, which is why rounds 1-3 compared windows only); the default schedule
  (the packed wave-pipelined depth-24 / depth-20 kernels fed through the LDS ring,
  whichever split of 1000 the scheduler prices cheapest) agrees with the depth-16 and
  the depth-8 ones everywhere; and the result does not depend on where the grid lies
  (the same input shifted by an odd number of rows and columns: every cell equal)."""
  prog = program('jacobi2d')
  n, it = 16384, 1000
  a = np.random.default_rng(6).random((n, n), dtype=np.float32)
  prog.set_max_depth(0)
  deep, timing = prog.run_numpy([a], iterate=it, timed=True)
  assert timing['max_depth'] >= 20
  want = oracle('jacobi2d').run([a], iterate=it)['t0']
  sl = oracle('jacobi2d').valid_slices((n, n), it)
  assert want[sl].shape == (n - 2 * it, n - 2 * it) and want[sl].std() > 0
  assert np.array_equal(deep[0][sl], want[sl])
  del want
  for limit in (16, 8):
    prog.set_max_depth(limit)
    other = prog.run_numpy([a], iterate=it)[0]
    assert np.array_equal(deep[0], other), limit
  prog.set_max_depth(0)
  # translation: the same input cut 91 rows and 37 columns in (no strip, chunk or
  # 64-byte piece lines up with the first run's) gives the same cells, every one of them
  del other
  dy, dx = 91, 37
  moved = prog.run_numpy([np.ascontiguousarray(a[dy:, dx:])], iterate=it)[0]
  assert moved.shape == (n - dy, n - dx)
  assert np.array_equal(moved[it:n - dy - it, it:n - dx - it],
                        deep[0][dy + it:n - it, dx + it:n - it])


def test_full_size_cfg5_jacobi3d_512_x200_and_cfg3_blur_16384():
  """BASELINE configs 5 and 3 at full size against the oracle, every cell."""
  prog = program('jacobi3d')
  a = np.random.default_rng(8).random((512, 512, 512), dtype=np.float32)
  got, timing = prog.run_numpy([a], iterate=200, timed=True)
  assert timing['max_depth'] == 4
  want = oracle('jacobi3d').run([a], iterate=200)[prog.spec['outputs'][0]]
  sl = oracle('jacobi3d').valid_slices((512, 512, 512), 200)
  assert want[sl].size == 112 ** 3 and np.array_equal(got[0][sl], want[sl])
  del a, got, want
  prog = program('blur')
  b = np.random.default_rng(9).integers(0, 65536, size=(16384, 16384)).astype(np.uint16)
  got = prog.run_numpy([b], iterate=1)[0]
  want = oracle('blur').run([b], iterate=1)[prog.spec['outputs'][0]]
  sl = oracle('blur').valid_slices((16384, 16384), 1)
  assert np.array_equal(got[sl], want[sl])


@pytest.mark.parametrize('app,dims,iterate', [('jacobi2d', [500, 300], 6),
                                              ('blur', [700, 300], 1)])
def test_generated_python_host_module_runs_as_the_readme_says(tmp_path, app, dims, iterate):
  """What `sodac --hip DIR` writes, RUN: `python DIR/<app>.py DIR/<app>.hsaco W H` as a
  child process (the README's quick start; the reference's test-bench, README.md:76-91,
  host.py:984-1167) says PASS with the reference's two timing lines, and the module's
  `<app>(inputs..., outputs..., blob)` (host.py:931-945) on caller-owned arrays equals
  the oracle on the valid box and leaves every other cell of the output alone."""
  import importlib.util
  import subprocess
  import sys
  from conftest import ROOT
  pkg = os.path.join(ROOT, 'soda-compiler_amd')
  out = tmp_path / 'out'
  subprocess.check_call([sys.executable, os.path.join(pkg, 'sodac'),
                         gpu_util.sample_path(app), '--iterate', str(iterate),
                         '--hip', str(out)])
  module, blob = out / (app + '.py'), out / (app + '.hsaco')
  for name in (app + '.py', app + '.hsaco', app + '_kernel.hip', app + '.h',
               app + '_host.cpp'):
    assert (out / name).exists(), name
  env = dict(os.environ, PYTHONPATH=os.pathsep.join(
      [pkg] + [p for p in os.environ.get('PYTHONPATH', '').split(os.pathsep) if p]))
  r = subprocess.run([sys.executable, str(module), str(blob)] + [str(d) for d in dims],
                     capture_output=True, text=True, env=env, timeout=600)
  assert r.returncode == 0, r.stderr[-2000:]
  assert 'INFO: PASS!' in r.stderr
  assert 'Kernel execution time:' in r.stdout and 'Kernel throughput:' in r.stdout
  # wrong argument count: the usage line and exit code 1 (README.md:79-83)
  r = subprocess.run([sys.executable, str(module), str(blob)], capture_output=True,
                     text=True, env=env, timeout=600)
  assert r.returncode == 1 and 'Usage:' in r.stderr
  # the module imported, its entry point on the caller's arrays
  spec_ = importlib.util.spec_from_file_location('generated_' + app, str(module))
  generated = importlib.util.module_from_spec(spec_)
  spec_.loader.exec_module(generated)
  assert generated.ITERATE == iterate
  spec = gpu_util.load_spec(app, iterate=iterate)
  assert generated.SPEC == json.loads(json.dumps(spec))
  shape = tuple(reversed(dims))
  inputs = gpu_util.random_inputs(spec, shape)
  marker = 7
  outputs = [np.full(shape, marker, dtype=inputs[0].dtype)]
  assert getattr(generated, app)(*inputs, *outputs, str(blob)) == 0
  orc = gpu_util.make_oracle(spec)
  want = orc.run(inputs, iterate=iterate)[spec['outputs'][0]]
  sl = orc.valid_slices(tuple(dims), iterate)
  assert want[sl].size > 0 and np.array_equal(outputs[0][sl], want[sl])
  outside = np.ones(shape, dtype=bool)
  outside[sl] = False
  assert (outputs[0][outside] == marker).all()
  # and the module's own <app>_test
  assert getattr(generated, app + '_test')(str(blob), dims + [0] * (4 - len(dims))) == 0


@pytest.mark.parametrize('app,dims,iterate', [
    ('jacobi2d', [500, 300], '12'), ('blur', [2000, 100], '1'),
    ('denoise2d', [300, 200], '1'), ('heat3d', [60, 50, 40], '3')])
def test_generated_cpp_host_program(tmp_path, app, dims, iterate):
  """`sodac --hip-host-cpp` output, built with g++ against include/soda_hip.h and
  libsoda_hip.so and run as the reference README's test-bench: the C-caller form
  of `<app>_test` (reference host.py:984-1167)."""
  import subprocess
  import sys
  from conftest import ROOT
  sodac = os.path.join(ROOT, 'soda-compiler_amd', 'sodac')
  csrc = os.path.join(ROOT, 'soda-compiler_amd', 'csrc')
  src = tmp_path / (app + '_host.cpp')
  subprocess.check_call([sys.executable, sodac,
                         os.path.join(ROOT, 'tests', 'samples', app + '.soda'),
                         '--hip-host-cpp', str(src)])
  exe = tmp_path / (app + '_host')
  subprocess.check_call(['g++', '-std=c++11', '-O1', '-fopenmp', '-ffp-contract=off',
                         '-DSODA_HIP_MAIN', '-I', os.path.join(ROOT, 'include'),
                         str(src), '-L', csrc, '-lsoda_hip', '-Wl,-rpath,' + csrc,
                         '-o', str(exe)])
  blob = os.path.join(gpu_util.BLOBS, app + '.hsaco')
  r = subprocess.run([str(exe), blob] + [str(d) for d in dims], capture_output=True,
                     text=True, env=dict(os.environ, SODA_ITERATE=iterate))
  assert r.returncode == 0, r.stderr
  assert 'INFO: PASS!' in r.stderr
  assert 'Kernel execution time:' in r.stdout
  # a blob of another program is refused with the mismatch code, not run
  other = os.path.join(gpu_util.BLOBS, 'seidel2d.hsaco')
  r = subprocess.run([str(exe), other] + [str(d) for d in dims], capture_output=True,
                     text=True)
  assert r.returncode != 0 and 'not generated for this program' in r.stderr


@pytest.mark.parametrize('app,dims,iterate', [
    ('jacobi2d', [700, 400], '30'), ('heat3d', [60, 50, 40], '5')])
def test_generated_cpp_host_multi_gpu_entry(tmp_path, app, dims, iterate):
  """`<app>_multi_gpu` of the generated C++ host (one thread per GPU over
  soda_hip_run_slab; -DSODA_HIP_MULTI_GPU): with the one GPU of this box the slab
  is the whole grid and no communicator is made - the reference's self-check must
  still say PASS.  (More than one GPU: ncclCommInitAll + ghost exchanges; that
  path has not run on hardware.)"""
  import subprocess
  import sys
  from conftest import ROOT
  sodac = os.path.join(ROOT, 'soda-compiler_amd', 'sodac')
  csrc = os.path.join(ROOT, 'soda-compiler_amd', 'csrc')
  src = tmp_path / (app + '_host.cpp')
  subprocess.check_call([sys.executable, sodac,
                         os.path.join(ROOT, 'tests', 'samples', app + '.soda'),
                         '--hip-host-cpp', str(src)])
  exe = tmp_path / (app + '_host')
  subprocess.check_call(['g++', '-std=c++17', '-O1', '-fopenmp', '-ffp-contract=off',
                         '-DSODA_HIP_MAIN', '-DSODA_HIP_MULTI_GPU',
                         '-D__HIP_PLATFORM_AMD__', '-I', '/opt/rocm/include',
                         '-I', os.path.join(ROOT, 'include'), str(src), '-L', csrc,
                         '-lsoda_hip', '-L/opt/rocm/lib', '-lrccl', '-lpthread',
                         '-Wl,-rpath,' + csrc, '-Wl,-rpath,/opt/rocm/lib',
                         '-o', str(exe)])
  blob = os.path.join(gpu_util.BLOBS, app + '.hsaco')
  r = subprocess.run([str(exe), blob] + [str(d) for d in dims], capture_output=True,
                     text=True, env=dict(os.environ, SODA_ITERATE=iterate, SODA_GPUS='1'))
  assert r.returncode == 0, r.stderr
  assert 'INFO: PASS!' in r.stderr


@pytest.mark.parametrize('app,dims,iterate,ranks', [
    ('jacobi2d', [700, 900], '70', '2'), ('jacobi2d', [500, 400], '40', '3'),
    ('heat3d', [60, 50, 90], '7', '3')])
def test_generated_cpp_host_multi_gpu_entry_with_several_ranks(tmp_path, app, dims,
                                                               iterate, ranks):
  """`<app>_multi_gpu` with 2 and 3 ranks: one thread per rank, ncclCommInitAll, the
  rendezvous before the first message, soda_hip_run_slab, the valid interior written
  back slab by slab - on this box's ONE GPU (SODA_HIP_REHEARSE_RANKS_ON_ONE_GPU) over
  the test-only librccl stand-in.  The reference's own self-check (CPU golden loops of
  the same generated file) must say PASS."""
  import subprocess
  import sys
  from conftest import ROOT
  standin = build_rccl_standin(tmp_path)
  sodac = os.path.join(ROOT, 'soda-compiler_amd', 'sodac')
  csrc = os.path.join(ROOT, 'soda-compiler_amd', 'csrc')
  src = tmp_path / (app + '_host.cpp')
  subprocess.check_call([sys.executable, sodac,
                         os.path.join(ROOT, 'tests', 'samples', app + '.soda'),
                         '--hip-host-cpp', str(src)])
  exe = tmp_path / (app + '_host')
  # librccl.so of the link line AND of libsoda_hip's dlopen is the stand-in: first on
  # the library path of this one child process
  subprocess.check_call(['g++', '-std=c++17', '-O1', '-fopenmp', '-ffp-contract=off',
                         '-DSODA_HIP_MAIN', '-DSODA_HIP_MULTI_GPU',
                         '-D__HIP_PLATFORM_AMD__', '-I', '/opt/rocm/include',
                         '-I', os.path.join(ROOT, 'include'), str(src), '-L', csrc,
                         '-lsoda_hip', '-L', os.path.dirname(standin), '-lrccl', '-lpthread',
                         '-Wl,-rpath,' + csrc, '-Wl,-rpath,' + os.path.dirname(standin),
                         '-Wl,-rpath,/opt/rocm/lib', '-o', str(exe)])
  blob = os.path.join(gpu_util.BLOBS, app + '.hsaco')
  env = dict(os.environ, SODA_ITERATE=iterate, SODA_GPUS=ranks,
             SODA_HIP_REHEARSE_RANKS_ON_ONE_GPU='1',
             LD_LIBRARY_PATH=os.path.dirname(standin) + ':' +
             os.environ.get('LD_LIBRARY_PATH', ''))
  r = subprocess.run([str(exe), blob] + [str(d) for d in dims], capture_output=True,
                     text=True, env=env, timeout=600)
  assert r.returncode == 0, r.stderr[-2000:]
  assert 'INFO: PASS!' in r.stderr
  assert os.path.basename(standin) == 'librccl.so'


def test_generated_multi_gpu_host_with_a_failing_rank(tmp_path):
  """The failure path of the generated `<app>_multi_gpu`: rank 1 of three is made
  to fail at its second super-step (SODA_HIP_FAIL_RANK, a hook behind SODA_HIP_TUNING).  Its
  thread aborts every communicator once - the library leaves them alone
  (soda_hip_slab.abort_on_error = 0): one owner, nothing aborted twice - the peers come out
  of their exchanges with an error, the program ends with a non-zero status instead of
  hanging, and the stand-in has seen no second abort of any communicator."""
  import subprocess
  import sys
  from conftest import ROOT
  standin = build_rccl_standin(tmp_path)
  sodac = os.path.join(ROOT, 'soda-compiler_amd', 'sodac')
  csrc = os.path.join(ROOT, 'soda-compiler_amd', 'csrc')
  src = tmp_path / 'jacobi2d_host.cpp'
  subprocess.check_call([sys.executable, sodac,
                         os.path.join(ROOT, 'tests', 'samples', 'jacobi2d.soda'),
                         '--hip-host-cpp', str(src)])
  exe = tmp_path / 'jacobi2d_host'
  subprocess.check_call(['g++', '-std=c++17', '-O1', '-fopenmp', '-ffp-contract=off',
                         '-DSODA_HIP_MAIN', '-DSODA_HIP_MULTI_GPU',
                         '-D__HIP_PLATFORM_AMD__', '-I', '/opt/rocm/include',
                         '-I', os.path.join(ROOT, 'include'), str(src), '-L', csrc,
                         '-lsoda_hip', '-L', os.path.dirname(standin), '-lrccl', '-lpthread',
                         '-Wl,-rpath,' + csrc, '-Wl,-rpath,' + os.path.dirname(standin),
                         '-Wl,-rpath,/opt/rocm/lib', '-o', str(exe)])
  blob = os.path.join(gpu_util.BLOBS, 'jacobi2d.hsaco')
  base = dict(os.environ, SODA_ITERATE='200', SODA_GPUS='3',
              SODA_HIP_REHEARSE_RANKS_ON_ONE_GPU='1',
              LD_LIBRARY_PATH=os.path.dirname(standin) + ':' +
              os.environ.get('LD_LIBRARY_PATH', ''))
  for extra in ({}, {'SODA_HIP_SLAB_STATIC_CUT': '1'}, {'SODA_HIP_SLAB_BANDS_FIRST': '1'}):
    r = subprocess.run([str(exe), blob, '700', '1500'], capture_output=True, text=True,
                       timeout=300, env=dict(base, SODA_HIP_TUNING='1', SODA_HIP_FAIL_RANK='1',
                                             SODA_HIP_FAIL_SUPERSTEP='1', **extra))
    assert r.returncode != 0, r.stdout[-500:]
    assert 'injected failure of rank 1 at super-step 1' in r.stderr, r.stderr[-2000:]
    assert 'aborted twice' not in r.stderr and 'INFO: PASS!' not in r.stderr
    # every rank reported: the failing one and the two whose exchanges were cut short
    assert r.stderr.count('ERROR: GPU') == 3, r.stderr[-2000:]
  # without the hook the same binary passes (the static cut as well as the default)
  for extra in ({}, {'SODA_HIP_SLAB_STATIC_CUT': '1'}):
    r = subprocess.run([str(exe), blob, '700', '1500'], capture_output=True, text=True,
                       timeout=300, env=dict(base, **extra))
    assert r.returncode == 0 and 'INFO: PASS!' in r.stderr, r.stderr[-2000:]


def test_run_slab_c_entry_with_one_rank():
  """soda_hip_run_slab / soda_hip_slab_extent (the slab driver below Python) with
  world = 1: same result as a plain sweep; a slab thinner than its ghost regions
  is refused."""
  import ctypes
  from soda_hip.runtime import capi
  prog = program('jacobi2d')
  spec = prog.spec
  (a,) = gpu_util.random_inputs(spec, (300, 900))
  want = prog.run_numpy([a], iterate=50)[0]
  slab = capi.Slab(rank=0, world=1, reach_lo=1, reach_hi=1, exchange=20,
                   own_first=0, own_last=300)
  slab.dims[0], slab.dims[1] = 900, 300
  local = (ctypes.c_int64 * 4)()
  glo, ghi = ctypes.c_int64(), ctypes.c_int64()
  capi.check(capi.lib().soda_hip_slab_extent(prog.handle, ctypes.byref(slab), local,
                                             ctypes.byref(glo), ctypes.byref(ghi)))
  assert list(local)[:2] == [900, 300] and (glo.value, ghi.value) == (0, 0)
  bufs = [host.DeviceArray(a.nbytes) for _ in range(3)]
  bufs[0].upload(a)
  for bf in bufs[1:]:
    bf.zero()
  result, n_ex = ctypes.c_void_p(), ctypes.c_int()
  capi.check(capi.lib().soda_hip_run_slab(
      prog.handle, ctypes.byref(slab), None, bufs[0].ptr, bufs[1].ptr, bufs[2].ptr, 50,
      None, ctypes.byref(result), ctypes.byref(n_ex)))
  capi.check(capi.lib().soda_hip_stream_synchronize(None))
  assert n_ex.value == 0 and result.value in (bufs[1].ptr, bufs[2].ptr)
  got = (bufs[1] if result.value == bufs[1].ptr else bufs[2]).download(a.shape, a.dtype)
  assert np.array_equal(got[50:-50, 50:-50], want[50:-50, 50:-50])
  # a middle rank whose ghost regions are deeper than its own rows
  bad = capi.Slab(rank=1, world=3, reach_lo=1, reach_hi=1, exchange=200,
                  own_first=100, own_last=200)
  bad.dims[0], bad.dims[1] = 900, 300
  assert capi.lib().soda_hip_slab_extent(prog.handle, ctypes.byref(bad), local, None,
                                         None) == -8
  mid = capi.Slab(rank=1, world=3, reach_lo=1, reach_hi=1, exchange=24,
                  own_first=100, own_last=200)
  mid.dims[0], mid.dims[1] = 900, 300
  capi.check(capi.lib().soda_hip_slab_extent(prog.handle, ctypes.byref(mid), local,
                                             ctypes.byref(glo), ctypes.byref(ghi)))
  assert list(local)[:2] == [900, 148] and (glo.value, ghi.value) == (24, 24)
  # world > 1 without a communicator is an error, not a crash
  assert capi.lib().soda_hip_run_slab(prog.handle, ctypes.byref(mid), None, bufs[0].ptr,
                                      bufs[1].ptr, bufs[2].ptr, 10, None,
                                      ctypes.byref(result), None) == -12
  for bf in bufs:
    bf.free()


def _buffer(extent=(), min_=(), host=None, elem_size=0):
  import ctypes
  from soda_hip.runtime import capi
  b = capi.BufferT()
  for d, n in enumerate(extent):
    b.extent[d] = n
  for d, n in enumerate(min_):
    b.min[d] = n
  if host is not None:
    b.host = host.ctypes.data
    stride = 1
    for d, n in enumerate(extent):
      b.stride[d] = stride
      stride *= n
  b.elem_size = elem_size
  return b


def _run_buffers(prog, ins, outs, iterate):
  import ctypes
  from soda_hip.runtime import capi
  pin = (ctypes.POINTER(capi.BufferT) * len(ins))(*[ctypes.pointer(b) for b in ins])
  pout = (ctypes.POINTER(capi.BufferT) * len(outs))(*[ctypes.pointer(b) for b in outs])
  return capi.lib().soda_hip_run_buffers(prog.handle, pin, pout, iterate, None)


def test_bounds_query_mode(capfd):
  """`<app>(buffer_t...)` called with buffers that have neither host nor device
  memory only reports shapes (reference host.py:204-252, halide_rewrite_buffer
  host.py:100-113): a null output keeps min and extents and gets dense strides; a
  null input gets the first output's min and its extents + stencil window - 1;
  dimensions beyond the program's are zeroed; elem_size is left alone; nothing
  runs and nothing is printed."""
  prog = program('jacobi2d')
  out = _buffer(extent=(100, 80), min_=(5, 7), elem_size=4)
  inp = _buffer(elem_size=123)
  inp.extent[3] = inp.stride[2] = inp.min[3] = 99     # stale values get cleared
  assert _run_buffers(prog, [inp], [out], 3) == 0
  # jacobi2d x3: window 7 x 7 (STENCIL_DIM of reference host.py:1183-1186)
  assert list(inp.extent) == [106, 86, 0, 0] and list(inp.min) == [5, 7, 0, 0]
  assert list(inp.stride) == [1, 106, 0, 0] and inp.elem_size == 123
  assert list(out.extent) == [100, 80, 0, 0] and list(out.min) == [5, 7, 0, 0]
  assert list(out.stride) == [1, 100, 0, 0]
  # only the input is null: the output's (caller-made) strides are left alone
  a = np.zeros((80, 100), dtype=np.float32)
  out2 = _buffer(extent=(100, 80), host=a, elem_size=4)
  inp2 = _buffer()
  assert _run_buffers(prog, [inp2], [out2], 1) == 0
  assert list(inp2.extent) == [102, 82, 0, 0] and list(out2.stride) == [1, 100, 0, 0]
  # a 3-D, two-stage-fused and a two-input program
  p3 = program('jacobi3d')
  o3, i3 = _buffer(extent=(20, 30, 40), min_=(1, 2, 3)), _buffer()
  assert _run_buffers(p3, [i3], [o3], 2) == 0
  assert list(i3.extent) == [24, 34, 44, 0] and list(i3.min) == [1, 2, 3, 0]
  assert list(i3.stride) == [1, 24, 24 * 34, 0] and list(o3.stride) == [1, 20, 600, 0]
  pb = program('blur')           # window 3 x 3, all towards higher indices
  ob, ib = _buffer(extent=(64, 48)), _buffer()
  assert _run_buffers(pb, [ib], [ob], 1) == 0
  assert list(ib.extent) == [66, 50, 0, 0]
  # two inputs: EVERY null input gets the window between the FIRST input and the first
  # output (reference host.py:226-233) - denoise2d reads its first input `f` at the
  # cell itself only, so both answers are the output's own extents, too small for `u`
  # (whose window is 5 x 5): the reference's answer, reproduced
  pd = program('denoise2d')
  od, i_f, i_u = _buffer(extent=(64, 48), min_=(2, 3)), _buffer(), _buffer()
  assert _run_buffers(pd, [i_f, i_u], [od], 1) == 0
  assert list(i_f.extent) == [64, 48, 0, 0] and list(i_u.extent) == [64, 48, 0, 0]
  assert list(i_u.min) == [2, 3, 0, 0] and list(i_u.stride) == [1, 64, 0, 0]
  captured = capfd.readouterr()
  assert 'Kernel execution time' not in captured.out


def test_bad_elem_size_is_halide_error_3(capfd):
  """Element-size checks come before anything touches the device (reference
  host.py:254-255, :969-982): -3 = halide_error_code_bad_elem_size, outputs are
  checked before inputs, a line on stderr names the buffer."""
  prog = program('jacobi2d')
  a = np.zeros((40, 50), dtype=np.float32)
  b = np.full((40, 50), 7, dtype=np.float32)
  good_in, good_out = _buffer((50, 40), host=a, elem_size=4), \
      _buffer((50, 40), host=b, elem_size=4)
  bad_in = _buffer((50, 40), host=a, elem_size=2)
  bad_out = _buffer((50, 40), host=b, elem_size=8)
  assert _run_buffers(prog, [bad_in], [good_out], 1) == -3
  assert host.capi.lib().soda_hip_error_name(-3) == b'bad_elem_size'
  assert 'input 0' in capfd.readouterr().err
  assert _run_buffers(prog, [bad_in], [bad_out], 1) == -3
  assert 'output 0' in capfd.readouterr().err          # outputs first
  assert (b == 7).all()                                 # nothing ran
  assert _run_buffers(prog, [good_in], [good_out], 1) == 0
  assert (b[1:-1, 1:-1] == 0).all() and b[0, 0] == 7    # valid interior only
  blur = program('blur')
  c = np.zeros((40, 50), dtype=np.float32)              # blur wants uint16
  assert _run_buffers(blur, [_buffer((50, 40), host=c, elem_size=4)],
                      [_buffer((50, 40), host=c, elem_size=2)], 1) == -3


def test_select_min_max_program():
  """select / min / max are in the reference's grammar (grammar.py:25-32) but its
  generated CPU host does not compile them (`min` is not declared), so there is no
  reference answer (tests/golden/extra_manifest.json records that).  The device
  semantics are the obvious C++ ones, checked against numpy here."""
  spec = gpu_util.load_spec('selectmm')
  prog = program('selectmm')
  (a,) = gpu_util.random_inputs(spec, (60, 300))
  got = prog.run_numpy([a])[0]
  # out(y, x) on rows 0..H-2, columns 1..W-2
  c, r, l, d = a[:-1, 1:-1], a[:-1, 2:], a[:-1, :-2], a[1:, 1:-1]
  want = np.where(c > r, np.minimum(c, d), np.maximum(r, l))
  assert np.array_equal(got[:-1, 1:-1], want)


def test_per_stage_kernels_beyond_the_grid_limit():
  """More rows than grid.y holds (65535): the per-stage kernels fold the rows into
  grid.y x grid.z instead of failing with buffer_extents_too_large; same cells as
  the fused kernels, which stream the rows in chunks."""
  prog = program('jacobi2d')
  a = np.random.default_rng(12).random((70001, 72), dtype=np.float32)
  prog.set_max_depth(-1)
  staged = prog.run_numpy([a], iterate=2)[0]
  prog.set_max_depth(0)
  fused = prog.run_numpy([a], iterate=2)[0]
  assert np.array_equal(staged, fused) and staged[2:-2, 2:-2].std() > 0
  p3 = program('jacobi3d')
  b = np.random.default_rng(13).random((3, 66001, 40), dtype=np.float32)
  p3.set_max_depth(-1)
  staged = p3.run_numpy([b], iterate=1)[0]
  want = oracle('jacobi3d').run([b], iterate=1)['t0']
  p3.set_max_depth(0)
  assert np.array_equal(staged[1:-1, 1:-1, 1:-1], want[1:-1, 1:-1, 1:-1])


def test_clock_probe_reports_the_shader_clock_under_load():
  """soda_hip_clock_probe_start / _finish: one wavefront sleeps beside the sweeps and
  counts shader cycles against the constant 100 MHz clock - what bench.py prints as
  roofline.shader_clock_ghz.  MI355X: at most 2.4 GHz; a loaded chip holds 1.9-2.4."""
  from soda_hip.runtime import host
  prog = program('jacobi2d')
  dims = [8192, 4096]
  a = np.random.default_rng(3).random((4096, 8192), dtype=np.float32)
  din = host.DeviceArray(a.nbytes)
  din.upload(a)
  dout = host.DeviceArray(a.nbytes)
  dout.zero()
  prog.set_max_depth(0)
  prog.sweep([din.ptr], [dout.ptr], dims, 48)          # scratch, clocks
  got = prog.shader_clock_during(
      lambda: [prog.sweep([din.ptr], [dout.ptr], dims, 48) for _ in range(20)], 0.02)
  host.capi.check(host.capi.lib().soda_hip_stream_synchronize(None))
  assert got is not None and 0.005 < got['seconds'] < 0.2
  assert 1.0 < got['ghz'] <= 2.45, got
  # one probe at a time, and _finish needs a _start
  import ctypes
  lib = host.capi.lib()
  assert lib.soda_hip_clock_probe_finish(prog.handle, ctypes.byref(ctypes.c_double()),
                                         None) != 0
  assert lib.soda_hip_clock_probe_start(prog.handle, 10) == 0
  assert lib.soda_hip_clock_probe_start(prog.handle, 10) != 0
  ghz = ctypes.c_double()
  assert lib.soda_hip_clock_probe_finish(prog.handle, ctypes.byref(ghz), None) == 0
  din.free()
  dout.free()


def test_tuning_the_streaming_launches_changes_speed_only():
  """soda_hip_plan_tune also times the chunk length x workgroup cap of memory-bound
  launches beyond the Infinity Cache against its neighbours (the calibrated pair was
  measured on one box): whatever it keeps, the sweep's results stay the oracle's, for a
  one-launch sweep (blur) and for one whose last launch streams (jacobi2d x 25)."""
  from soda_hip.runtime import host
  for app, dims, iterate in (('blur', [16384, 10240], 1), ('jacobi2d', [12288, 8192], 25)):
    prog = gpu_util.open_prebuilt(app)
    try:
      spec = gpu_util.load_spec(app, iterate=iterate)
      dt = prog.in_dtypes[0]
      rng = np.random.default_rng(31)
      shape = tuple(reversed(dims))
      a = rng.random(shape, dtype=np.float32) if dt.kind == 'f' else \
          rng.integers(0, 65536, size=shape).astype(dt)
      din = host.DeviceArray(a.nbytes)
      din.upload(a)
      dout = host.DeviceArray(a.nbytes)
      dout.zero()
      before = [e['name'] for e, _ in prog.schedule(dims, iterate)]
      prog.tune([din.ptr], [dout.ptr], dims, iterate)
      # (the split of iterate may change; every launch is still a kernel of the blob)
      assert sum(max(1, e['depth']) for e, _ in prog.schedule(dims, iterate)) == iterate
      assert before
      dout.zero()
      prog.sweep([din.ptr], [dout.ptr], dims, iterate)
      got = dout.download(a.shape, a.dtype)
      orc = gpu_util.make_oracle(gpu_util.load_spec(app))
      want = orc.run([a], iterate=iterate)[spec['outputs'][0]]
      sl = orc.valid_slices(tuple(dims), iterate)
      assert want[sl].size > 0 and np.array_equal(got[sl], want[sl])
      din.free()
      dout.free()
    finally:
      prog.close()


def test_tall_narrow_streaming_box_beyond_the_grid_limit():
  """A streaming launch beyond the Infinity Cache walks the kernel's MEASURED chunk
  (soda_hip_kernel.stream_chunk: 8-32 rows), and 600 000 rows in chunks of 8 would be
  75 000 workgroups along grid.y (limit 65535): the launcher lengthens the chunk instead
  of failing with buffer_extents_too_large (round-4 advice); results as the oracle's."""
  prog = program('jacobi2d')
  rows, cols = 600000, 256                  # 2 x 586 MiB: beyond the cache
  prog.set_max_depth(1)
  sched = prog.schedule([cols, rows], 1)
  assert [e['name'] for e, _ in sched] == ['jacobi2d_fused_k1']
  assert sched[0][0].get('stream_chunk', 0) * 65535 < rows       # the case the advice names
  a = np.random.default_rng(14).random((rows, cols), dtype=np.float32)
  got = prog.run_numpy([a], iterate=1)[0]
  want = oracle('jacobi2d').run([a], iterate=1)['t0']
  prog.set_max_depth(0)
  assert np.array_equal(got[1:-1, 1:-1], want[1:-1, 1:-1])


def test_half_program_against_ieee_binary16_arithmetic():
  """`half` programs have no reference answer here (the reference's golden loop
  needs Xilinx's hls_half.h, g++ 11 has no _Float16 in C++): they are checked against
  IEEE binary16 arithmetic emulated with numpy - every `half op half` rounded to
  half (float32 holds the exact sum / product of two halves, so rounding it once
  is the correctly rounded half result), `half * float-literal` promoted to float
  as in C++ and rounded on the store - on the fused and the per-stage kernels, two
  iterations.  Unpinned against the reference, pinned against IEEE."""
  from soda_hip import frontend
  from soda_hip.codegen import kernel, spec as specmod
  text = ('kernel: halfj\nburst width: 512\nunroll factor: 2\niterate: 2\n'
          'input half: a(32, *)\n'
          'output half: b(0, 0) = (a(0, 1) + a(1, 0) + a(0, 0) + a(0, -1) + a(-1, 0))'
          ' * 0.2f\n')
  spec = specmod.spec_from_stencil(frontend.loads(text))
  src, table = kernel.generate(spec)
  prog = host.open_program(source=src, spec=spec)
  a = np.random.default_rng(21).random((90, 333), dtype=np.float32).astype(np.float16)

  def step(x):
    s = x[2:, 1:-1] + x[1:-1, 2:]          # float16 + float16 -> rounded to float16
    s = s + x[1:-1, 1:-1]
    s = s + x[:-2, 1:-1]
    s = s + x[1:-1, :-2]
    out = np.zeros_like(x)
    out[1:-1, 1:-1] = (s.astype(np.float32) * np.float32(0.2)).astype(np.float16)
    return out
  want = step(step(a))
  for max_depth in (0, 1, -1):
    prog.set_max_depth(max_depth)
    got = prog.run_numpy([a], iterate=2)[0]
    assert got.dtype == np.float16
    assert np.array_equal(got[2:-2, 2:-2], want[2:-2, 2:-2]), max_depth
  prog.close()
  prog.blob.unload()


def test_one_dimensional_program_jit():
  """A 1-D program (the grammar's `name(*)` form): per-stage kernels, JIT path,
  iterate 3, against the oracle."""
  from soda_hip import frontend
  from soda_hip.codegen import kernel
  from soda_hip.codegen import spec as specmod
  text = '''
    kernel: smooth1d
    burst width: 512
    unroll factor: 1
    iterate: 3
    input float: a(*)
    output float: b(0) = (a(-1) + a(0) * 2.0f + a(1)) * 0.25f
  '''
  spec = specmod.spec_from_stencil(frontend.loads(text))
  assert spec['dim'] == 1
  src, table = kernel.generate(spec)
  prog = host.open_program(source=src, spec=spec)
  a = np.random.default_rng(3).random((100003,), dtype=np.float32)
  got = prog.run_numpy([a], iterate=3)[0]
  orc = gpu_util.make_oracle(spec)
  want = orc.run([a], iterate=3)['b']
  assert np.array_equal(got[3:-3], want[3:-3])
  assert got[3:-3].std() > 0
  prog.close()


def test_four_dimensional_program(tmp_path, capfd):
  """A 4-D program (`buffer_t` and `<app>_test(blob, dims[4])` stop at four
  dimensions, reference header.py:36-48, host.py:992): per-stage kernels with
  dimensions 1..3 folded into grid.y x grid.z.  Small boxes are pinned to the
  reference's own CPU loops by the extra.hyper4d fixtures (test_fixture); here a
  larger box against the oracle, the Python `<app>_test` and the generated C++
  host program."""
  import subprocess
  import sys
  from conftest import ROOT
  spec = gpu_util.load_spec('hyper4d')
  assert spec['dim'] == 4
  inputs = gpu_util.random_inputs(spec, (11, 23, 31, 203))
  check('hyper4d', inputs, 1)
  check('hyper4d', inputs, 4)
  blob = os.path.join(gpu_util.BLOBS, 'hyper4d.hsaco')
  assert host.app_test(spec, blob, [37, 12, 11, 10]) == 0
  out, err = capfd.readouterr()
  assert 'INFO: PASS!' in err
  sodac = os.path.join(ROOT, 'soda-compiler_amd', 'sodac')
  csrc = os.path.join(ROOT, 'soda-compiler_amd', 'csrc')
  src = tmp_path / 'hyper4d_host.cpp'
  subprocess.check_call([sys.executable, sodac, gpu_util.sample_path('hyper4d'),
                         '--hip-host-cpp', str(src)])
  exe = tmp_path / 'hyper4d_host'
  subprocess.check_call(['g++', '-std=c++11', '-O1', '-fopenmp', '-ffp-contract=off',
                         '-DSODA_HIP_MAIN', '-I', os.path.join(ROOT, 'include'),
                         str(src), '-L', csrc, '-lsoda_hip', '-Wl,-rpath,' + csrc,
                         '-o', str(exe)])
  r = subprocess.run([str(exe), blob, '37', '12', '11', '10'], capture_output=True,
                     text=True, env=dict(os.environ, SODA_ITERATE='3'))
  assert r.returncode == 0, r.stderr
  assert 'INFO: PASS!' in r.stderr


def test_multi_output_program_uses_stage_kernels():
  """Two outputs with different windows: each output is defined on its own box
  (reference host.py:1082-1091); only the per-stage kernels can honour that."""
  from soda_hip import frontend
  from soda_hip.codegen import kernel
  from soda_hip.codegen import spec as specmod
  text = '''
    kernel: two_out
    burst width: 512
    unroll factor: 1
    iterate: 1
    input float: a(64, *)
    output float: sx(0, 0) = a(0, 0) + a(1, 0)
    output float: sy(0, 0) = a(0, 0) - a(0, 3)
  '''
  spec = specmod.spec_from_stencil(frontend.loads(text))
  src, table = kernel.generate(spec)
  assert all(k['kind'] == 'stage' for k in table)
  prog = host.open_program(source=src, spec=spec)
  a = np.random.default_rng(4).random((70, 130), dtype=np.float32)
  sx, sy = prog.run_numpy([a], iterate=1)
  assert np.array_equal(sx[:, :-1], a[:, :-1] + a[:, 1:])
  assert np.array_equal(sy[:-3, :], a[:-3, :] - a[3:, :])
  prog.close()


def test_multi_output_host_buffer_protocol():
  """soda_hip_run_buffers writes back each output's own valid box."""
  from soda_hip import frontend
  from soda_hip.codegen import kernel
  from soda_hip.codegen import spec as specmod
  text = '''
    kernel: two_out
    burst width: 512
    unroll factor: 1
    iterate: 1
    input float: a(64, *)
    output float: sx(0, 0) = a(0, 0) + a(1, 0)
    output float: sy(0, 0) = a(0, 0) - a(0, 3)
  '''
  spec = specmod.spec_from_stencil(frontend.loads(text))
  src, _ = kernel.generate(spec)
  prog = host.open_program(source=src, spec=spec)
  a = np.random.default_rng(4).random((70, 130), dtype=np.float32)
  sx = np.full_like(a, -1.0)
  sy = np.full_like(a, -1.0)
  prog.run_buffers([a], [sx, sy], 1)
  assert np.array_equal(sx[:, :-1], a[:, :-1] + a[:, 1:]) and (sx[:, -1] == -1).all()
  assert np.array_equal(sy[:-3, :], a[:-3, :] - a[3:, :]) and (sy[-3:, :] == -1).all()
  prog.close()


def test_output_that_feeds_another_output():
  """tests/samples/extra/outchain.soda: `second` reads `first`, both are outputs.
  The reference's own CPU loops have no self-contained answer for such a program
  (they keep `first` in a scalar, read the DEVICE's `first` array for `second` and
  never compare `first`: host.py:1104-1118, core.py:146; extra_manifest.json records
  it), so the device result is checked against the oracle and against numpy: each
  output on its own box, the host-buffer entry leaving the rest untouched."""
  prog = program('outchain')
  assert all(k['kind'] == 'stage' for k in prog.kernels)
  a = np.random.default_rng(5).random((90, 150), dtype=np.float32)
  f32 = np.float32
  m = (a[:-1, :-1] + a[:-1, 1:] + a[1:, :-1]) * f32(0.25)     # rows 0..H-2, cols 0..W-2
  first = m[:, 1:] - m[:, :-1] * f32(0.5)                      # rows 0..H-2, cols 1..W-2
  second = (first[:-2, :-1] + first[2:, 1:]) + a[1:-2, 1:-2] * f32(2.0)
  got1, got2 = prog.run_numpy([a], iterate=1)
  assert np.array_equal(got1[:-1, 1:-1], first)
  assert np.array_equal(got2[1:-2, 1:-2], second)
  want = oracle('outchain').run([a], iterate=1)
  assert np.array_equal(want['first'][:-1, 1:-1], first)
  assert np.array_equal(want['second'][1:-2, 1:-2], second)
  o1 = np.full_like(a, -1.0)
  o2 = np.full_like(a, -1.0)
  prog.run_buffers([a], [o1, o2], 1)
  assert np.array_equal(o1[:-1, 1:-1], first) and np.array_equal(o2[1:-2, 1:-2], second)
  for o, inner in ((o1, (slice(0, -1), slice(1, -1))), (o2, (slice(1, -2), slice(1, -2)))):
    rest = np.ones(a.shape, bool)
    rest[inner] = False
    assert (o[rest] == -1).all()


@pytest.mark.parametrize('app,dims,world,exchange,iterate', [
    ('jacobi2d', (1500, 611), 2, 12, 30), ('jacobi2d', (1500, 611), 3, 5, 17),
    ('jacobi2d', (1500, 611), 4, 24, 48), ('jacobi2d', (900, 1400), 2, 144, 300),
    ('jacobi3d', (130, 70, 96), 2, 8, 20), ('jacobi3d', (67, 45, 120), 3, 4, 13),
    ('heat3d', (100, 64, 90), 2, 12, 24), ('hyper4d', (40, 14, 13, 33), 3, 2, 5),
    # ghost zones of different depth below and above (2 rows up, 1 down per iteration)
    ('skew2d', (1500, 611), 3, 12, 40), ('skew2d', (1100, 900), 4, 7, 21)])
def test_slab_decomposition_with_the_hip_engine(app, dims, world, exchange, iterate):
  """The multi-GPU driver's slab logic (soda_hip.runtime.dist) run with the REAL
  kernels: all ranks emulated on this one GPU, ghost rows copied by hand where
  RCCL would move them.  Covers what the gloo tests cannot: soda_hip_sweep's
  valid_lo/valid_hi contract on slabs with ghost rows (2-D) and ghost planes
  (3-D, where the depth-4 kernel moves its edge tiles inside the slab)."""
  import torch
  from soda_hip.codegen import spec as specmod
  from soda_hip.runtime import dist as sdist
  prog = program(app)
  spec = prog.spec
  dim = spec['dim']
  shape = tuple(reversed(dims))
  full = np.random.default_rng(11).random(shape, dtype=np.float32)
  table = specmod.iteration_margins(spec, iterate)
  zero = (tuple([0] * dim), tuple([0] * dim))
  margins_of = lambda k: zero if k == 0 else table[k - 1]
  engine = sdist.HipEngine(prog, torch)
  r_lo, r_hi = spec['radius']['lo'][-1], spec['radius']['hi'][-1]
  plans = [sdist.SlabPlan(list(dims), r, world, r_lo, r_hi, exchange)
           for r in range(world)]
  assert (r_lo, r_hi) == ((2, 1) if app == 'skew2d' else (1, 1))
  dev = torch.device('cuda', 0)
  cur, nxt = [], []
  for p in plans:
    a = torch.zeros(tuple(reversed(p.local_dims)), dtype=torch.float32, device=dev)
    a[p.ghost_lo:p.ghost_lo + p.own] = torch.from_numpy(full[p.start:p.stop]).to(dev)
    cur.append(a)
    nxt.append(torch.zeros_like(a))
  done = 0
  while done < iterate:
    for r, p in enumerate(plans):          # what exchange_ghosts() does over RCCL
      if p.has_lo:
        q = plans[r - 1]
        cur[r][0:p.ghost_lo] = cur[r - 1][q.ghost_lo + q.own - p.ghost_lo:
                                          q.ghost_lo + q.own]
      if p.has_hi:
        q = plans[r + 1]
        cur[r][p.ghost_lo + p.own:] = cur[r + 1][q.ghost_lo:q.ghost_lo + p.ghost_hi]
    step = min(plans[0].exchange, iterate - done)
    for r, p in enumerate(plans):
      lo, hi = p.valid_margins(done, margins_of)
      engine.sweep(cur[r], nxt[r], p.local_dims, step, lo, hi)
    torch.cuda.synchronize()
    cur, nxt = nxt, [torch.zeros_like(a) for a in nxt]
    done += step
  got = np.zeros_like(full)
  for r, p in enumerate(plans):
    got[p.start:p.stop] = cur[r][p.ghost_lo:p.ghost_lo + p.own].cpu().numpy()
  want = oracle(app).run([full], iterate=iterate)[spec['outputs'][0]]
  sl = oracle(app).valid_slices(dims, iterate)
  assert want[sl].size > 0
  assert np.array_equal(got[sl], want[sl])


@pytest.mark.parametrize('app,dims,world,exchange,iterate,bands', [
    ('jacobi2d', (1500, 611), 3, 5, 17, False), ('jacobi2d', (1500, 611), 4, 24, 48, True),
    ('jacobi2d', (900, 1400), 8, 48, 300, True),
    # cfg5's proportions on 8 ranks: 20-plane slabs, 40 iterations - the valid range ends
    # with 10 planes per rank, partners beyond the nearest rank
    ('jacobi3d', (170, 150, 160), 8, 4, 40, False), ('jacobi3d', (170, 150, 160), 8, 8, 40, True),
    ('heat3d', (140, 64, 90), 3, 12, 24, True), ('skew2d', (1100, 900), 4, 7, 21, True)])
def test_recut_decomposition_with_the_hip_engine(app, dims, world, exchange, iterate, bands):
  """Slabs re-cut every super-step (soda_hip.runtime.dist.RecutPlan) with the REAL kernels:
  all ranks emulated on this one GPU - up to eight, more than processes may share it - the
  rows of RecutPlan.messages copied by hand where RCCL would move them, the sweeps those of
  run_recut (sub-arrays of the rows a rank reads, both outer sides declared valid; `bands`:
  cut into RecutPlan.pieces, each piece written by its last launch only).  Covers what the
  gloo tests cannot: soda_hip_sweep on those sub-arrays, wide 3-D planes whose first and
  last tile columns store the edge columns, ranks whose share shrinks to a few planes."""
  import torch
  from soda_hip.codegen import spec as specmod
  from soda_hip.runtime import dist as sdist
  prog = program(app)
  spec = prog.spec
  dim = spec['dim']
  full = np.random.default_rng(12).random(tuple(reversed(dims)), dtype=np.float32)
  table = specmod.iteration_margins(spec, iterate)
  zero = (tuple([0] * dim), tuple([0] * dim))
  margins_of = lambda k: zero if k == 0 else table[k - 1]      # noqa: E731
  engine = sdist.HipEngine(prog, torch)
  r_lo, r_hi = spec['radius']['lo'][-1], spec['radius']['hi'][-1]
  plans = [sdist.RecutPlan(list(dims), r, world, r_lo, r_hi, exchange, iterate)
           for r in range(world)]
  dev = torch.device('cuda', 0)
  cur, nxt = [], []
  for p in plans:
    a = torch.full(tuple(reversed(p.local_dims)), float('nan'), dtype=torch.float32, device=dev)
    a[p.ghost_lo:p.ghost_lo + p.own] = torch.from_numpy(full[p.start:p.stop]).to(dev)
    cur.append(a)
    nxt.append(torch.full_like(a, float('nan')))
  try:
    for s, (done, step) in enumerate(plans[0].steps):
      for r, p in enumerate(plans):          # what exchange_rows() does over RCCL
        for peer, rows in p.messages(s)[1]:
          a0, a1 = p.local(rows)
          b0, b1 = plans[peer].local(rows)
          cur[r][a0:a1] = cur[peer][b0:b1]
      for r, p in enumerate(plans):
        lo, hi = (list(v) for v in margins_of(done))
        lo[-1] = hi[-1] = 0
        out = (p.cuts[s][r], p.cuts[s][r + 1])
        pieces = p.pieces(s) if bands else None
        todo = [(o, True) for o in pieces[0] + [pieces[1]]] if pieces else \
            [(out, False)] if out[1] > out[0] else []
        for (o0, o1), final_only in todo:
          engine.sweep(cur[r], nxt[r], p.local_dims, step, lo, hi,
                       rows=(o0 - step * r_lo - p.base, o1 + step * r_hi - p.base),
                       final_only=final_only)
      torch.cuda.synchronize()
      cur, nxt = nxt, [torch.full_like(a, float('nan')) for a in nxt]
  finally:
    prog.set_out_final_only(False)
  got = np.zeros_like(full)
  for r, p in enumerate(plans):
    (g0, g1), (l0, l1) = p.final_rows, p.local(p.final_rows)
    got[g0:g1] = cur[r][l0:l1].cpu().numpy()
  want = oracle(app).run([full], iterate=iterate)[spec['outputs'][0]]
  sl = oracle(app).valid_slices(dims, iterate)
  assert want[sl].size > 0
  assert np.array_equal(got[sl], want[sl])


@pytest.mark.parametrize('app,dims,world,iterate,exchange,mode', [
    ('jacobi2d', (1300, 900), 2, 70, 24, 'serial'),
    ('jacobi2d', (1300, 1500), 3, 100, 20, 'overlap'),
    ('jacobi3d', (130, 70, 200), 2, 20, 8, 'overlap'),
    # super-steps of three launches (24 + 24 + 12; 3 x depth 4): a piece's
    # intermediate launches must not write the array its neighbours' pieces share
    ('jacobi2d', (1300, 1800), 3, 150, 60, 'overlap'),
    ('jacobi3d', (130, 70, 300), 3, 24, 12, 'overlap'),
    # slabs too thin for bands (90 own rows, 2 x 24 to send each way): the whole-slab
    # sweep is followed by an exchange on the side stream all the same
    ('jacobi2d', (1300, 270), 3, 60, 24, 'overlap')])
def test_multi_process_slabs_on_one_gpu(tmp_path, app, dims, world, iterate, exchange,
                                        mode):
  """The multi-rank driver end to end with the real kernels: `world` processes (a
  gloo group; all on this box's one GPU, so ghost rows go through the host instead
  of RCCL) each run soda_hip.runtime.dist.run_slab with the HIP engine - serial
  order, and the overlapping order (boundary bands first, exchange on a side
  stream beside the interior sweep).  Own rows put together equal the oracle's
  single-process result bit for bit."""
  import socket
  import subprocess
  import sys
  from conftest import ROOT
  s = socket.socket()
  s.bind(('127.0.0.1', 0))
  port = s.getsockname()[1]
  s.close()
  env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
             WORLD_SIZE=str(world), OMP_NUM_THREADS='2')
  procs = [subprocess.Popen(
      [sys.executable, os.path.join(ROOT, 'tests', 'dist_gpu_worker.py'), app,
       'x'.join(map(str, dims)), str(iterate), str(exchange), str(tmp_path), mode],
      env=dict(env, RANK=str(r), LOCAL_RANK=str(r))) for r in range(world)]
  for p in procs:
    assert p.wait(timeout=600) == 0
  spec = gpu_util.load_spec(app, iterate=iterate)
  full = np.random.default_rng(99).random(tuple(reversed(dims)), dtype=np.float32)
  orc = oracle(app)
  want = orc.run([full], iterate=iterate)[spec['outputs'][0]]
  got = np.zeros_like(want)
  n_exchanges = []
  for r in range(world):
    start, stop, ex, n_ex, ms = open(os.path.join(tmp_path, 'rank%d.txt' % r)).read().split()
    got[int(start):int(stop)] = np.load(os.path.join(tmp_path, 'rank%d.npy' % r))
    n_exchanges.append(int(n_ex))
  sl = orc.valid_slices(tuple(dims), iterate)
  assert want[sl].size > 0 and np.array_equal(got[sl], want[sl])
  assert min(n_exchanges) >= -(-iterate // exchange)


@pytest.mark.parametrize('app,shape,iterate', [
    ('jacobi2d', (700, 1300), 50), ('seidel2d', (500, 1100), 30),
    ('jacobi3d', (150, 140, 130), 10)])
def test_tuned_split_changes_speed_only(app, shape, iterate):
  """soda_hip_plan_tune runs the candidate splits of `iterate` as whole sweeps and
  keeps the fastest for these extents: the schedule afterwards still adds up to
  `iterate`, the result is still the oracle's, and other extents are unaffected."""
  from soda_hip.runtime import capi
  spec = gpu_util.load_spec(app)
  prog = gpu_util.open_prebuilt(app)      # a plan of its own: tuning is plan state
  try:
    (a,) = gpu_util.random_inputs(spec, shape)
    dims = list(reversed(shape))
    din, dout = host.DeviceArray(a.nbytes), host.DeviceArray(a.nbytes)
    din.upload(a)
    dout.zero()
    before = [k['depth'] for k, _ in prog.schedule(dims, iterate)]
    prog.tune([din.ptr], [dout.ptr], dims, iterate)
    after = [k['depth'] for k, _ in prog.schedule(dims, iterate)]
    assert sum(after) == iterate == sum(before)
    assert after == sorted(after, reverse=True)
    other = [k['depth'] for k, _ in prog.schedule([d + 8 for d in dims], iterate)]
    assert sum(other) == iterate
    prog.sweep([din.ptr], [dout.ptr], dims, iterate)
    got = dout.download(a.shape, a.dtype)
    orc = oracle(app)
    want = orc.run([a], iterate=iterate)[spec['outputs'][0]]
    sl = orc.valid_slices(tuple(dims), iterate)
    assert want[sl].size > 0 and np.array_equal(got[sl], want[sl])
    # a GIVEN split (soda_hip_plan_set_split: what profiling passes repeat a timed
    # run's schedule with): used as given, same result; rejected when it does not
    # add up or names a depth the blob has no kernel of; an empty one gives the
    # choice back to the scheduler
    given = [1] * (iterate % 2) + [2] * (iterate // 2)
    prog.set_split(dims, iterate, given)
    assert [k['depth'] for k, _ in prog.schedule(dims, iterate)] == given
    dout.zero()
    prog.sweep([din.ptr], [dout.ptr], dims, iterate)
    assert np.array_equal(dout.download(a.shape, a.dtype)[sl], want[sl])
    with pytest.raises(capi.SodaHipError, match='add up'):
      prog.set_split(dims, iterate, [2] * (iterate // 2 + 1))
    with pytest.raises(capi.SodaHipError, match='no fused kernel of depth 7'):
      prog.set_split(dims, iterate, [7])
    prog.set_split(dims, iterate, [])
    assert sum(k['depth'] for k, _ in prog.schedule(dims, iterate)) == iterate
    assert max(k['depth'] for k, _ in prog.schedule(dims, iterate)) > 2
    din.free()
    dout.free()
  finally:
    prog.close()


_STANDIN = {}


def build_rccl_standin(tmp_path):
  """tests/rccl_standin: ncclSend / ncclRecv / groups for ranks that are threads of
  one process on one GPU (a TEST library; soname librccl.so so that libsoda_hip's
  dlopen resolves to it inside the worker process only).  Built once per test session."""
  import subprocess
  import tempfile
  from conftest import ROOT
  if _STANDIN.get('path') and os.path.exists(_STANDIN['path']):
    return _STANDIN['path']
  _STANDIN['dir'] = tempfile.TemporaryDirectory(prefix='rccl_standin_')
  out = _STANDIN['path'] = os.path.join(_STANDIN['dir'].name, 'librccl.so')
  subprocess.check_call([
      'g++', '-O1', '-fPIC', '-shared', '-std=c++17', '-Wl,-soname,librccl.so',
      '-D__HIP_PLATFORM_AMD__', '-I/opt/rocm/include',
      os.path.join(ROOT, 'tests', 'rccl_standin', 'rccl_standin.cpp'), '-o', out,
      '-L/opt/rocm/lib', '-Wl,-rpath,/opt/rocm/lib', '-lamdhip64', '-lpthread'])
  return out


@pytest.mark.parametrize('app,dims,world,iterate,exchange,order', [
    ('jacobi2d', (1300, 900), 2, 70, 24, 0),
    ('jacobi2d', (700, 1500), 3, 100, 60, 0),  # three launches per super-step
    ('skew2d', (900, 800), 3, 30, 8, 0),       # ghost regions 2 rows deep per iteration
                                               # on one side, 1 on the other
    ('jacobi3d', (130, 70, 200), 2, 20, 8, 0),
    ('jacobi3d', (140, 150, 90), 3, 9, 100, 0),   # period clamped to the smallest slab
    # bands first, exchange on the plan's second stream beside the interior sweep
    # (SODA_HIP_SLAB_BANDS_FIRST): super-steps of three and more launches, whose
    # intermediate launches must not touch rows that are being sent
    ('jacobi2d', (700, 1500), 3, 200, 60, 1),
    ('jacobi2d', (1300, 2400), 2, 130, 44, 1),
    ('skew2d', (900, 1200), 3, 50, 12, 1),
    ('jacobi3d', (130, 70, 200), 2, 20, 8, 1),
    ('jacobi2d', (600, 900), 3, 250, 100, 1)])  # slabs too thin for bands: whole sweeps
def test_run_slab_with_two_and_three_ranks_over_the_rccl_standin(tmp_path, app, dims,
                                                                 world, iterate, exchange,
                                                                 order):
  run_slab_over_the_standin(tmp_path, build_rccl_standin(tmp_path), app, dims, world,
                            iterate, exchange, order)


@pytest.mark.parametrize('app,dims,world,iterate,exchange,order', [
    ('jacobi2d', (1300, 900), 2, 70, 24, 0),
    ('jacobi2d', (700, 1500), 3, 100, 60, 0),
    ('skew2d', (900, 800), 3, 30, 8, 0),          # one-sided-ish window: the cuts drift
    ('jacobi2d', (700, 1600), 4, 200, 48, 0),
    ('jacobi2d', (700, 1600), 4, 200, 48, 1),     # bands first
    ('jacobi2d', (1300, 2400), 2, 130, 44, 1),
    ('skew2d', (900, 1200), 3, 50, 12, 1),
    # cfg5's proportions: 48-plane slabs, 64 iterations - with the static cut the first and
    # last ranks run out of valid planes after 48; here every rank keeps a quarter of what
    # is left, and rows change owner every super-step
    ('jacobi3d', (160, 160, 192), 4, 64, 8, 0),
    ('jacobi3d', (160, 160, 192), 4, 64, 16, 1),
    # a period longer than the slabs are thick: partners beyond the nearest rank
    ('jacobi2d', (600, 400), 4, 150, 120, 0)])
def test_run_slab_recut_over_the_rccl_standin(tmp_path, app, dims, world, iterate, exchange,
                                              order):
  """soda_hip_slab.cut = SODA_HIP_SLAB_CUT_RECUT below the C ABI: every super-step the rows
  its output level defines are cut evenly again, ghost rows and rows changing owner travel
  in one group.  The ranks' rows of the result tile the final valid range and equal the
  oracle bit for bit, in both orders."""
  run_slab_over_the_standin(tmp_path, build_rccl_standin(tmp_path), app, dims, world,
                            iterate, exchange, order, cut='recut')


def run_slab_over_the_standin(tmp_path, standin, app, dims, world, iterate, exchange,
                              order, cut='static'):
  """soda_hip_run_slab - the C slab driver: ncclSend / ncclRecv inside one group per
  super-step on the caller's stream, then the sweep - with world > 1.  RCCL refuses two
  ranks on one GPU and this box has one, so the ranks are host threads over a
  test-only stand-in for librccl.so (stream-ordered device-to-device copies matched
  through a mailbox).  Own rows put together equal the oracle bit for bit; the
  stand-in's counters prove that the ghost rows really travelled."""
  import subprocess
  import sys
  from conftest import ROOT
  from soda_hip.codegen import spec as specmod
  r = subprocess.run(
      [sys.executable, os.path.join(ROOT, 'tests', 'rccl_standin_worker.py'), standin,
       app, 'x'.join(map(str, dims)), str(world), str(iterate), str(exchange),
       str(tmp_path), str(order)] + (['cut=recut'] if cut == 'recut' else []),
      capture_output=True, text=True, timeout=600)
  assert r.returncode == 0, r.stderr[-2000:]
  spec = gpu_util.load_spec(app, iterate=iterate)
  dt = np.dtype(specmod.NUMPY_NAME[spec['inputs'][0]['c_type']])
  rng = np.random.default_rng(99)
  shape = tuple(reversed(dims))
  full = rng.random(shape, dtype=np.float32).astype(dt)
  orc = gpu_util.make_oracle(gpu_util.load_spec(app))
  want = orc.run([full], iterate=iterate)[spec['outputs'][0]]
  got = np.zeros_like(want)
  held = []
  for rank in range(world):
    first, last, period, count, messages, nbytes = map(int, open(
        os.path.join(tmp_path, 'rank%d.txt' % rank)).read().split())
    got[first:last] = np.load(os.path.join(tmp_path, 'rank%d.npy' % rank))
    held.append((first, last))
    assert count == -(-iterate // period)
  sl = orc.valid_slices(tuple(dims), iterate)
  assert want[sl].size > 0 and np.array_equal(got[sl], want[sl])
  r_lo, r_hi = spec['radius']['lo'][-1], spec['radius']['hi'][-1]
  row_bytes = int(np.prod(dims[:-1])) * dt.itemsize
  if cut == 'recut':
    # the ranks' rows tile the rows still valid, as evenly as they divide, and the rows
    # that travelled are the ones RecutPlan (the Python driver's geometry) lists
    from soda_hip.runtime import dist as sdist
    assert held[0][0] == iterate * r_lo and held[-1][1] == dims[-1] - iterate * r_hi
    assert all(a[1] == b[0] for a, b in zip(held, held[1:]))
    assert max(b - a for a, b in held) - min(b - a for a, b in held) <= 1
    plans = [sdist.RecutPlan(list(dims), r, world, r_lo, r_hi, period, iterate)
             for r in range(world)]
    assert held == plans[0].final
    sent = [rows for p in plans for i in range(len(p.steps)) for _, rows in p.messages(i)[0]]
    assert messages == len(sent) > 0
    assert nbytes == sum(b - a for a, b in sent) * row_bytes
    return
  # per exchange and interior boundary: period * r_lo rows up and period * r_hi down
  # (a one-sided window sends nothing one way)
  assert messages == count * (world - 1) * ((r_lo > 0) + (r_hi > 0))
  assert nbytes == count * (world - 1) * period * (r_lo + r_hi) * row_bytes


@pytest.mark.parametrize('order,fail_rank,fail_at,options', [
    (0, 1, 1, []), (1, 0, 1, []), (1, 2, 0, ['cut=recut']),
    (0, 1, 1, ['abort=lib']), (1, 2, 1, ['abort=lib', 'cut=recut'])])
def test_a_rank_that_fails_mid_run_does_not_leave_its_peers_blocked(tmp_path, order,
                                                                    fail_rank, fail_at,
                                                                    options):
  """A rank that fails after the first message (a failed launch, allocation or exchange;
  here SODA_HIP_FAIL_RANK / _SUPERSTEP, a test hook behind SODA_HIP_TUNING) must not leave
  its peers waiting in ncclRecv for rows that will never come.  ncclCommAbort is LOCAL to a
  rank (the stand-in's is too), so what unblocks the peers is the driver of all ranks of the
  process aborting every communicator once (the worker's abort_all = the generated
  `<app>_multi_gpu`'s); with soda_hip_slab.abort_on_error the library has aborted the
  failing rank's own before returning, and the driver must leave that one alone.  Every
  rank of a three-rank group returns an error within the time limit, in both orders and
  both cuts, and no communicator is aborted twice."""
  import subprocess
  import sys
  from conftest import ROOT
  standin = build_rccl_standin(tmp_path)
  r = subprocess.run(
      [sys.executable, os.path.join(ROOT, 'tests', 'rccl_standin_worker.py'), standin,
       'jacobi2d', '700x1500', '3', '200', '48', str(tmp_path), str(order),
       'expect-failure'] + options, capture_output=True, text=True, timeout=240,
      env=dict(os.environ, SODA_HIP_TUNING='1', SODA_HIP_FAIL_RANK=str(fail_rank),
               SODA_HIP_FAIL_SUPERSTEP=str(fail_at)))
  assert r.returncode == 0, r.stderr[-2000:]
  lines = open(os.path.join(tmp_path, 'errors.txt')).read().splitlines()
  assert len(lines) == 3
  for rank, text in enumerate(lines):
    assert text.split()[0] == str(rank) and 'soda_hip_run_slab' in text, lines
    if rank != fail_rank:     # the peers' exchange failed on THEIR aborted communicator
      assert 'communicator aborted' in text, lines
  assert 'injected failure of rank %d at super-step %d' % (fail_rank, fail_at) in \
      lines[fail_rank]
  # the library's own abort happens exactly when asked for
  assert ('(communicator aborted)' in lines[fail_rank]) == ('abort=lib' in options)
  assert open(os.path.join(tmp_path, 'double_abort.txt')).read().strip() == '0'
  assert 'aborted twice' not in r.stderr


def test_run_slab_checks_its_arguments_before_anything_is_sent():
  """A bad `iterate`, order, cut or geometry is an error of the call, found before the first
  message: it is returned without touching the communicator (a poisoned pointer here: any
  use would crash), even with abort_on_error set (a caller that forgot to zero a
  new field must not lose its communicator over it)."""
  import ctypes
  from soda_hip.runtime import capi
  lib = capi.lib()
  prog = gpu_util.open_prebuilt('jacobi2d')
  try:
    slab = capi.Slab()
    slab.rank, slab.world, slab.reach_lo, slab.reach_hi, slab.exchange = 0, 2, 1, 1, 8
    slab.dims[0], slab.dims[1] = 512, 400
    slab.own_first, slab.own_last = 0, 200
    slab.abort_on_error = 1
    poison = ctypes.c_void_p(0xdead0000)
    buf = host.DeviceArray(512 * 300 * 4)
    result = ctypes.c_void_p()

    def run(iterate):
      return lib.soda_hip_run_slab(prog.handle, ctypes.byref(slab), poison, buf.ptr, buf.ptr,
                                   buf.ptr, iterate, None, ctypes.byref(result), None)
    assert run(0) == -8 and 'iterate' in lib.soda_hip_last_error().decode()
    slab.order = 7
    assert run(4) == -8 and 'order' in lib.soda_hip_last_error().decode()
    slab.order, slab.cut = 0, 9
    assert run(4) == -8 and 'cut' in lib.soda_hip_last_error().decode()
    slab.cut, slab.exchange = 0, 0
    assert run(4) == -8
    slab.exchange, slab.cut, slab.own_last = 8, capi.SLAB_CUT_RECUT, 150
    assert run(4) == -8 and 'even' in lib.soda_hip_last_error().decode()
    assert 'communicator aborted' not in lib.soda_hip_last_error().decode()
    buf.free()
  finally:
    prog.close()


def test_bench_py_as_the_driver_launches_it_for_two_gpus():
  """`python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 ... bench.py
  --gpus 2 --steps K --warmup W`, the driver's command for N > 1, on this box's ONE
  GPU: SODA_DIST_BACKEND=gloo lets two ranks share the device (RCCL refuses that),
  everything else - slabs, ghost exchange inside the timed region, barrier and
  max-over-ranks timing, the one JSON line of rank 0 - is the production path."""
  import socket
  import subprocess
  import sys
  from conftest import ROOT
  s = socket.socket()
  s.bind(('127.0.0.1', 0))
  port = s.getsockname()[1]
  s.close()
  r = subprocess.run(
      [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node',
       '2', '--master-addr', '127.0.0.1', '--master-port', str(port),
       os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1',
       '--size', '4096', '3000', '--iterate', '120', '--cpu-seconds', '2',
       '--cpu-baseline-at-all-n'],
      capture_output=True, text=True, timeout=600,
      env=dict(os.environ, SODA_DIST_BACKEND='gloo', OMP_NUM_THREADS='2'))
  assert r.returncode == 0, r.stderr[-2000:]
  lines = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
  assert len(lines) == 1, r.stdout[-2000:]
  d = json.loads(lines[0])
  assert d['n_gpus'] == 2 and d['steps'] == 2 and d['warmup'] == 1
  assert d['metric'] == 'gcell_updates_per_s' and d['unit'] == 'Gcell-updates/s'
  assert d['scaling'] == 'strong' and d['higher_is_better'] is True
  assert d['vs_baseline'] is None and d['dtype'] == 'f32' and d['data'] == 'synthetic'
  c = d['config']
  assert c['dims'] == [4096, 3000] and c['iterate'] == 120
  assert c['exchanges_per_step'] >= 1 and c['exchange_every'] >= 1
  assert c['ghost_rows'] == [c['exchange_every'], c['exchange_every']]
  # before the timed region the chosen cut, period and order ran once on a small grid and
  # every rank's rows equalled the one-rank sweep of rank 0 (a wrong ghost row prints no
  # throughput: test_bench_py_refuses_a_run_whose_ghost_rows_arrive_wrong)
  assert c['multi_rank_check'] == 'bit-exact' and c['slab_cut'].split()[0] in ('recut', 'static')
  assert c['multi_rank_check_grid'].startswith('4096x') and c['multi_rank_check_iterate'] >= 24
  # the exchange period and the order (serial / overlapped) are the fastest of the
  # candidates timed during warm-up, and the table is on the line
  table = c['exchange_candidates_ms']
  assert c['exchange_choice'].startswith('measured')
  assert sorted({r['exchange'] for r in table}) == [24, 48, 96, 120]
  # eight (period, order) pairs under the re-cut, then the chosen pair under the static cut
  assert len(table) == 9 and all(r['ms'] > 0 for r in table)
  assert [r['cut'] for r in table] == ['recut'] * 8 + ['static']
  assert {r['overlapped'] for r in table} == {False, True}
  best = min(table[:8], key=lambda r: r['ms'])
  assert (table[8]['exchange'], table[8]['overlapped']) in {
      (r['exchange'], r['overlapped']) for r in table[:8]}
  if c['slab_cut'].startswith('recut'):
    assert (c['exchange_every'], c['exchange_overlapped']) == (best['exchange'],
                                                               best['overlapped'])
  else:
    assert table[8]['ms'] < 0.98 * min(r['ms'] for r in table[:8]
                                       if (r['exchange'], r['overlapped']) ==
                                       (table[8]['exchange'], table[8]['overlapped']))
  # the same steps without the exchanges: never (much) slower than with them
  assert 0 < c['compute_only_ms_per_step'] < 1.25 * d['ms_per_step']
  # whole-job throughput on VALID updates of the whole grid, both ranks' rows
  spec = gpu_util.load_spec('jacobi2d', iterate=120)
  from soda_hip.codegen import spec as specmod
  valid = specmod.valid_cells(spec, [4096, 3000], 120)
  assert c['valid_cell_updates'] == valid
  assert abs(d['value'] - valid / (d['ms_per_step'] * 1e-3) / 1e9) < 1e-6 * d['value']
  rf = d['roofline']
  assert rf['kernel'].startswith('jacobi2d_fused_k') and 0 < rf['frac'] < 1.5
  # with --cpu-baseline-at-all-n the CPU figure stands beside this point of the curve too
  # (by default it is on the N = 1 line only): rank 0 times the oracle port on the whole
  # grid AFTER the timed region, the others wait
  cpu = d['cpu_baseline']
  assert cpu['kind'] == 'port' and cpu['value'] > 0 and len(cpu['samples']) == 3
  assert c['super_step_schedule']


@pytest.mark.parametrize('cut', ['recut', 'static'])
def test_bench_py_refuses_a_run_whose_ghost_rows_arrive_wrong(cut):
  """The N > 1 bench line verifies itself: with one cell of the rows rank 1 receives damaged
  in every exchange (SODA_DIST_CORRUPT_GHOST, a hook that exists under SODA_HIP_TUNING=1
  only) the self-check before the timed region finds differing cells, rank 0 prints a line
  WITHOUT a value and every rank exits non-zero - the driver's scaling run is the first
  time real RCCL moves these rows, and a wrong row must not yield a throughput."""
  import socket
  import subprocess
  import sys
  from conftest import ROOT
  s = socket.socket()
  s.bind(('127.0.0.1', 0))
  port = s.getsockname()[1]
  s.close()
  cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node',
         '2', '--master-addr', '127.0.0.1', '--master-port', str(port),
         os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0',
         '--size', '4096', '1500', '--iterate', '60', '--cpu-seconds', '0',
         '--no-exchange-tune', '--no-tune'] + (['--static-cut'] if cut == 'static' else [])
  env = dict(os.environ, SODA_DIST_BACKEND='gloo', OMP_NUM_THREADS='2')
  bad = subprocess.run(cmd, capture_output=True, text=True, timeout=600,
                       env=dict(env, SODA_HIP_TUNING='1', SODA_DIST_CORRUPT_GHOST='1'))
  assert bad.returncode != 0
  lines = [l for l in bad.stdout.splitlines() if l.startswith('{"metric"')]
  assert len(lines) == 1, bad.stdout[-2000:] + bad.stderr[-2000:]
  d = json.loads(lines[0])
  assert d['value'] is None and d['error'] == 'multi-rank self-check failed'
  assert d['config']['multi_rank_check'].endswith('cells differ')
  assert int(d['config']['multi_rank_check'].split()[0]) > 0
  # the hook is dead without SODA_HIP_TUNING=1: the same command then passes
  good = subprocess.run(cmd, capture_output=True, text=True, timeout=600,
                        env=dict(env, SODA_DIST_CORRUPT_GHOST='1'))
  assert good.returncode == 0, good.stderr[-2000:]
  d = json.loads([l for l in good.stdout.splitlines() if l.startswith('{"metric"')][0])
  assert d['config']['multi_rank_check'] == 'bit-exact' and d['value'] > 0
  assert d['config']['slab_cut'].startswith(cut)


def test_bench_py_falls_back_to_the_conservative_configuration():
  """When the self-check fails for the configuration the warm-up chose, bench.py checks the
  most conservative one in its place - static cut, serial order - and goes on with it only
  if THAT is bit-exact, saying so on the line.  Here only the re-cut's exchanges are damaged
  (SODA_DIST_CORRUPT_CUT=recut): the run ends with rc 0, slab_cut static, serial order, and
  multi_rank_check names what failed."""
  import socket
  import subprocess
  import sys
  from conftest import ROOT
  s = socket.socket()
  s.bind(('127.0.0.1', 0))
  port = s.getsockname()[1]
  s.close()
  r = subprocess.run(
      [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
       '--master-addr', '127.0.0.1', '--master-port', str(port),
       os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0',
       '--size', '4096', '1500', '--iterate', '60', '--cpu-seconds', '0', '--no-tune',
       '--recut'], capture_output=True, text=True, timeout=600,
      env=dict(os.environ, SODA_DIST_BACKEND='gloo', OMP_NUM_THREADS='2', SODA_HIP_TUNING='1',
               SODA_DIST_CORRUPT_GHOST='1', SODA_DIST_CORRUPT_CUT='recut'))
  # (--recut: the warm-up does not get to prefer the static cut by timing; being told the
  # cut is no reason to refuse the fallback - only --static-cut is already the fallback)
  assert r.returncode == 0, r.stderr[-2000:]
  d = json.loads([l for l in r.stdout.splitlines() if l.startswith('{"metric"')][0])
  c = d['config']
  assert d['value'] > 0 and c['slab_cut'] == 'static' and c['exchange_overlapped'] is False
  assert c['multi_rank_check'].startswith('bit-exact with the static cut and the serial order')
  assert 're-cut cut' in c['multi_rank_check'] and 'cells differ' in c['multi_rank_check']
  assert 'self-check FAILED' in r.stderr


def test_bench_py_for_the_three_dimensional_multi_gpu_config():
  """BASELINE cfg5 is an 8-GPU config (jacobi3d 512^3 x 200 in 64-plane slabs: the edge
  ranks' own planes leave the valid box after 64 iterations while the middle ranks go
  on).  The driver's command line for it, rehearsed on this box's ONE GPU with four gloo
  ranks and the same proportions (128 x 128 x 160, 52 iterations, 40-plane slabs): the line, the candidate
  table, the compute-only time, and nobody stalls on a rank whose boxes are empty."""
  import socket
  import subprocess
  import sys
  from conftest import ROOT
  s = socket.socket()
  s.bind(('127.0.0.1', 0))
  port = s.getsockname()[1]
  s.close()
  r = subprocess.run(
      [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node',
       '4', '--master-addr', '127.0.0.1', '--master-port', str(port),
       os.path.join(ROOT, 'bench.py'), '--gpus', '4', '--steps', '1', '--warmup', '0',
       '--app', 'jacobi3d', '--size', '128', '128', '160', '--iterate', '52',
       '--cpu-seconds', '0', '--no-tune'],
      capture_output=True, text=True, timeout=900,
      env=dict(os.environ, SODA_DIST_BACKEND='gloo', OMP_NUM_THREADS='2'))
  assert r.returncode == 0, r.stderr[-2000:]
  lines = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
  assert len(lines) == 1, r.stdout[-2000:]
  d = json.loads(lines[0])
  c = d['config']
  assert d['n_gpus'] == 4 and c['dims'] == [128, 128, 160] and c['iterate'] == 52
  assert c['parallelism'] == 'outer-dim slabs x4' and d['scaling'] == 'strong'
  table = c['exchange_candidates_ms']
  # periods 4, 8, 16, 32 (1, 2, 4, 8 x the deepest 3-D kernel), serial and overlapped
  assert sorted({row['exchange'] for row in table}) == [4, 8, 16, 32]
  assert len(table) == 9 and all(row['ms'] > 0 and row['repeats'] >= 3 for row in table)
  assert [row['cut'] for row in table] == ['recut'] * 8 + ['static']
  assert (c['exchange_every'], c['exchange_overlapped']) in {
      (row['exchange'], row['overlapped']) for row in table}
  assert c['exchanges_per_step'] == -(-52 // c['exchange_every'])
  assert 0 < c['compute_only_ms_per_step'] < 1.5 * d['ms_per_step']
  assert c['multi_rank_check'] == 'bit-exact' and c['slab_cut'].split()[0] in ('recut', 'static')
  from soda_hip.codegen import spec as specmod
  valid = specmod.valid_cells(gpu_util.load_spec('jacobi3d', iterate=52), [128, 128, 160], 52)
  assert abs(d['value'] - valid / (d['ms_per_step'] * 1e-3) / 1e9) < 1e-6 * d['value']
  assert d['roofline']['kernel'].startswith('jacobi3d_fused_k')


def test_bench_py_one_rank_group_over_real_rccl():
  """The N > 1 code path of bench.py with torch's `nccl` backend (= RCCL) for real, as far
  as one GPU allows: a one-rank process group (`--force-dist`) - RCCL initialisation
  with a device id, the barrier and the MAX all-reduce on device tensors that bracket the
  timed region, the slab plan with no neighbours, the JSON line.  What it cannot cover is
  a send / receive pair between two devices (the gloo rehearsal above and the RCCL
  stand-in cover their ordering and contents)."""
  import subprocess
  import sys
  from conftest import ROOT
  r = subprocess.run(
      [sys.executable, os.path.join(ROOT, 'bench.py'), '--force-dist', '--steps', '2',
       '--warmup', '1', '--size', '4096', '2000', '--iterate', '60', '--cpu-seconds', '0'],
      capture_output=True, text=True, timeout=600,
      env={k: v for k, v in os.environ.items()
           if k not in ('SODA_DIST_BACKEND', 'RANK', 'WORLD_SIZE', 'LOCAL_RANK')})
  assert r.returncode == 0, r.stderr[-2000:]
  lines = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
  assert len(lines) == 1, r.stdout[-2000:]
  d = json.loads(lines[0])
  c = d['config']
  assert d['n_gpus'] == 1 and c['parallelism'] == 'outer-dim slabs x1'
  assert c['ghost_rows'] == [c['exchange_every'], c['exchange_every']]
  assert c['multi_rank_check'] == 'bit-exact'
  assert c['exchange_choice'] == 'given' and c['exchange_candidates_ms'] == []
  assert 0 < c['compute_only_ms_per_step'] < 1.25 * d['ms_per_step']
  spec = gpu_util.load_spec('jacobi2d', iterate=60)
  from soda_hip.codegen import spec as specmod
  valid = specmod.valid_cells(spec, [4096, 2000], 60)
  assert abs(d['value'] - valid / (d['ms_per_step'] * 1e-3) / 1e9) < 1e-6 * d['value']


def test_bench_py_headline_line_carries_the_other_single_gpu_configs():
  """The driver's own command (`python bench.py --steps 2 --warmup 1`) prints the headline
  workload and, under config.other_configs, BASELINE configs 2, 3 and 5 timed on the same
  device in the same process ("plus cfg 2, 3, 5 single-point numbers")."""
  import subprocess
  import sys
  from conftest import ROOT
  r = subprocess.run(
      [sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '2', '--warmup', '1',
       '--cpu-seconds', '0'], capture_output=True, text=True, timeout=900,
      env={k: v for k, v in os.environ.items()
           if k not in ('SODA_DIST_BACKEND', 'RANK', 'WORLD_SIZE', 'LOCAL_RANK')})
  assert r.returncode == 0, r.stderr[-2000:]
  lines = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
  assert len(lines) == 1, r.stdout[-2000:]
  d = json.loads(lines[0])
  assert d['config']['workload'] == 'jacobi2d.soda float32 16384x16384, iterate 1000'
  assert d['roofline']['kernel'].startswith('jacobi2d_fused_k')
  others = d['config']['other_configs']
  assert [o['config'] for o in others] == ['cfg2', 'cfg3', 'cfg5']
  assert [o['workload'] for o in others] == [
      'jacobi2d.soda float32 8192x8192, iterate 100',
      'blur.soda uint16 16384x16384, iterate 1',
      'jacobi3d.soda float32 512x512x512, iterate 200']
  from soda_hip.codegen import spec as specmod
  for o, (app, dims, iterate) in zip(others, (('jacobi2d', [8192, 8192], 100),
                                              ('blur', [16384, 16384], 1),
                                              ('jacobi3d', [512, 512, 512], 200))):
    valid = specmod.valid_cells(gpu_util.load_spec(app, iterate=iterate), dims, iterate)
    assert abs(o['gcell_updates_per_s'] - valid / (o['ms'] * 1e-3) / 1e9) < 1e-6 * valid
    assert o['steps'] == 30 and o['warmup'] == 10 and 0 < o['ms'] < 50
    rf = o['roofline']
    assert rf['bound'] in ('hbm', 'valu') and 0 < rf['frac'] <= 1.0
    assert rf['frac_algorithmic'] > 0 and rf['kernel_avg_us'] * rf['kernel_launches'] \
        <= 1.001 * o['ms'] * 1e3


def test_denormals_signed_zeros_and_infinities():
  """IEEE corner cases: subnormal inputs (no flush-to-zero on either side),
  negative zeros, infinities (inf - inf = NaN must appear in the same cells)."""
  rng = np.random.default_rng(21)
  shape = (90, 700)
  a = (rng.random(shape, dtype=np.float32) * np.float32(3e-39)).astype(np.float32)
  a[rng.random(shape) < 0.05] = np.float32(-0.0)
  a[10, 100] = np.inf
  a[40, 300] = -np.inf
  a[41, 300] = np.inf
  assert (np.abs(a[np.isfinite(a)]) < np.finfo(np.float32).tiny).all()
  prog = program('jacobi2d')
  # 16 and 28 iterations: the packed (v_pk_*_f32) depth-16 and depth-12 kernels;
  # jacobi averages, so the values stay subnormal all the way
  for it, depth, deepest in ((1, 0, 1), (5, 0, 4), (5, 1, 1), (16, 0, 16), (28, 0, 16)):
    prog.set_max_depth(depth)
    got, timing = prog.run_numpy([a], iterate=it, timed=True)
    got = got[0]
    assert timing['max_depth'] == deepest
    want = oracle('jacobi2d').run([a], iterate=it)['t0']
    sl = oracle('jacobi2d').valid_slices((shape[1], shape[0]), it)
    assert np.array_equal(np.isnan(got[sl]), np.isnan(want[sl]))
    finite = ~np.isnan(want[sl])
    # bit patterns, so that -0.0 vs +0.0 and subnormals are told apart
    assert np.array_equal(got[sl][finite].view(np.uint32),
                          want[sl][finite].view(np.uint32))
    assert (got[sl][finite] != 0).any()
    tiny = np.abs(got[sl][finite])
    assert ((tiny > 0) & (tiny < np.finfo(np.float32).tiny)).any()
  prog.set_max_depth(0)


@pytest.mark.parametrize('app', ['jacobi3d', 'heat3d'])
def test_denormals_in_the_deep_3d_kernel(app):
  """The same corner cases through the depth-4 3-D kernel (scalar for jacobi3d,
  packed pair-rows for heat3d)."""
  rng = np.random.default_rng(22)
  shape = (40, 45, 90)
  a = (rng.random(shape, dtype=np.float32) * np.float32(3e-39)).astype(np.float32)
  a[rng.random(shape) < 0.05] = np.float32(-0.0)
  a[10, 20, 30] = np.inf
  a[20, 21, 50] = -np.inf
  a[20, 22, 50] = np.inf
  prog = program(app)
  got, timing = prog.run_numpy([a], iterate=4, timed=True)
  assert timing['max_depth'] == 4
  name = gpu_util.load_spec(app)['outputs'][0]
  want = oracle(app).run([a], iterate=4)[name]
  sl = oracle(app).valid_slices(tuple(reversed(shape)), 4)
  assert np.array_equal(np.isnan(got[0][sl]), np.isnan(want[sl]))
  finite = ~np.isnan(want[sl])
  assert np.array_equal(got[0][sl][finite].view(np.uint32),
                        want[sl][finite].view(np.uint32))
  tiny = np.abs(got[0][sl][finite])
  assert ((tiny > 0) & (tiny < np.finfo(np.float32).tiny)).any()


def test_grid_beyond_2_to_31_elements():
  """64-bit indexing: 50000 x 43000 floats = 2.15e9 cells (8.6 GB per array),
  29 iterations (one depth-16, one depth-12 - both fed by LDS-direct loads - and
  one depth-1 launch).  Bands of rows at the top,
  just below and above the 2^31-element boundary and at the bottom are compared
  with the oracle run on sub-grids containing their dependency cones."""
  w, h, it = 50000, 43000, 29
  assert w * h > 2 ** 31
  prog = program('jacobi2d')
  cols = (np.arange(w, dtype=np.int64) * 7 % 1013).astype(np.float32) / np.float32(1013)
  rows = (np.arange(h, dtype=np.int64) * 13 % 997).astype(np.float32) / np.float32(997)
  a = np.empty((h, w), dtype=np.float32)
  np.add(rows[:, None], cols[None, :], out=a)
  din = host.DeviceArray(a.nbytes)
  dout = host.DeviceArray(a.nbytes)
  try:
    din.upload(a)
    prog.set_max_depth(0)
    prog.sweep([din.ptr], [dout.ptr], [w, h], it)
    host.capi.check(host.capi.lib().soda_hip_stream_synchronize(None))
    boundary_row = 2 ** 31 // w          # first row whose cells pass 2^31
    for first in (it, boundary_row - 8, boundary_row + 1, h - it - 16):
      band = np.empty((16, w), dtype=np.float32)
      host.capi.check(host.capi.lib().soda_hip_memcpy_d2h(
          band.ctypes.data, dout.ptr + first * w * 4, band.nbytes, None))
      host.capi.check(host.capi.lib().soda_hip_stream_synchronize(None))
      sub = np.ascontiguousarray(a[first - it:first + 16 + it])
      want = oracle('jacobi2d').run([sub], iterate=it)['t0'][it:it + 16]
      assert np.array_equal(band[:, it:w - it], want[:, it:w - it]), first
      assert band[:, it:w - it].std() > 0
  finally:
    din.free()
    dout.free()
    prog.close()
    _PROGRAMS.pop('jacobi2d', None)


@pytest.mark.parametrize('fixture', ['layout.jacobi2d.iter1.40x21.npz',
                                     'layout.blur.iter1.50x9.npz',
                                     'layout.jacobi3d.iter2.20x18x6.npz'])
def test_run_on_data_in_the_reference_dram_layout(fixture):
  """Buffers the reference's own tiling loops produced (tests/golden/layout.*)
  go in; what comes out, read back the way the reference's copy-back loops read
  its device buffers, equals the oracle on every copied-back valid cell."""
  import json
  from soda_hip import frontend
  from soda_hip.codegen import kernel, spec as specmod
  from soda_hip.runtime import layout
  golden = os.path.join(gpu_util.ROOT, 'tests', 'golden')
  with open(os.path.join(golden, 'layout_manifest.json')) as f:
    meta = json.load(f)[fixture]
  data = np.load(os.path.join(golden, fixture))
  st = frontend.load(os.path.join(gpu_util.ROOT, 'tests', 'samples',
                                  meta['app'] + '.soda'), iterate=meta['iterate'])
  spec = specmod.spec_from_stencil(st)
  prog = host.open_program(source=kernel.generate(spec)[0], spec=spec)
  in_buffers = {n: {b: data['inbuf_%s_%d' % (n, b)] for b in meta['banks_in']}
                for n in st.input_names}
  out = layout.run_in_reference_layout(prog, st, in_buffers, meta['dims'],
                                       meta['tile_size'], meta['banks_in'],
                                       meta['banks_out'])
  c = layout.stencil_constants(st, meta['tile_size'])
  inputs = [data['in_' + n] for n in st.input_names]
  orc = gpu_util.make_oracle(spec)
  want = orc.run(inputs, iterate=meta['iterate'])
  sl = orc.valid_slices(tuple(meta['dims']), meta['iterate'])
  for name in st.output_names:
    lay = layout.TiledLayout(meta['dims'], meta['tile_size'], c['stencil_dim'],
                             meta['burst_width'], want[name].dtype.itemsize * 8,
                             meta['banks_out'], input_banks=meta['banks_in'])
    got = lay.unpack(out[name], c['copy_back_offset'], c['stencil_offset'][name],
                     window_dim=c['copy_back_dim'])
    assert np.array_equal(got[sl], want[name][sl]) and want[name][sl].size
  prog.close()
  prog.blob.unload()


def test_streaming_launches_of_the_depth1_kernel_use_the_measured_chunk_and_cap():
  """soda_hip_kernel.stream_chunk (ABI 6) and stream_wgs_per_cu (ABI 5): a launch of the
  depth-1 kernel whose box does not fit the Infinity Cache walks the chunk length that
  tools/calibrate.py measured as the fastest for it (short chunks, many workgroups
  dispatched in address order) under the measured cap on workgroups per CU, while a
  cache-resident launch keeps the launcher's own rule (the longest chunk that fills the
  chip in whole rounds, no cap); results are the same either way (chunk and cap decide
  where rows are computed, not what)."""
  import re
  import subprocess
  import sys
  from conftest import ROOT
  code = (
      'import sys, numpy as np\n'
      'sys.path[:0] = [%r, %r, %r]\n'
      'import gpu_util\n'
      'from soda_hip.runtime import host\n'
      'prog = gpu_util.open_prebuilt("jacobi2d")\n'
      'prog.set_max_depth(1)\n'
      'for n in (3072, 12288):\n'
      '  a = host.DeviceArray(n * n * 4); a.zero()\n'
      '  b = host.DeviceArray(n * n * 4); b.zero()\n'
      '  prog.sweep_timed([a.ptr], [b.ptr], [n, n], 1, warmup=0, repeats=1)\n'
      '  a.free(); b.free()\n' % (ROOT, os.path.join(ROOT, 'soda-compiler_amd'),
                                   os.path.join(ROOT, 'tests')))
  r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True,
                     env=dict(os.environ, SODA_HIP_TUNING='1', SODA_HIP_LAUNCH_TRACE='1'),
                     timeout=300)
  assert r.returncode == 0, r.stderr[-2000:]
  launches = re.findall(r'launch\s+\d+ jacobi2d_fused_k1\s+[\d.]+ us \(model\s+[\d.]+\)  box (\d+) x '
                        r'(\d+) x \d+  grid (\d+) x (\d+) x \d+  chunk (\d+)  fill \d+  '
                        r'resident \d+  lds (\d+)', r.stderr)
  assert len(launches) == 2, r.stderr[-2000:]
  entry = [k for k in program('jacobi2d').kernels if k['name'] == 'jacobi2d_fused_k1'][0]
  assert entry['exact'] == 1
  cus = host.device_info(0)['compute_units']
  small, big = [(int(gx) * int(gy), int(chunk), int(lds))
                for _, _, gx, gy, chunk, lds in launches]
  assert small[0] > 2 * cus and small[2] == 0          # 3072^2 (72 MiB): the chip is filled
  if entry.get('stream_chunk', 0) > 0:      # 12288^2: the measured chunk, many workgroups
    assert big[1] == entry['stream_chunk'] and big[0] > 8 * cus
  else:
    assert big[1] > 4 * small[1]
  if entry.get('stream_wgs_per_cu', 0) > 0:     # the cap: dynamic LDS nobody touches
    cap = entry['stream_wgs_per_cu']
    assert 160 * 1024 // (cap + 1) < big[2] <= 160 * 1024 // cap
  else:
    assert big[2] == 0
