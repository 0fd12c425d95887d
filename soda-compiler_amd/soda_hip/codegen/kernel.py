"""HIP kernel printer: program spec -> one translation unit of gfx950 kernels.

Counterpart of the reference's HLS kernel printer (reference
src/soda/codegen/xilinx/hls_kernel.py:146-189): same input (the analysed
stencil), and the arithmetic of every stage is the same expression text
(hls_kernel.py:487-489 `WriteData(..., c_type(expr.c_expr))`); everything around
it is designed for a GPU instead of an FPGA dataflow region.

The unit holds
  * one per-stage kernel per stage (kernel_stage.py), always;
  * fused kernels of depth 1, 2, 4, ... (kernel_stream2d.py) when the program
    is in their scope;
  * `soda_hip_meta`, JSON describing the program and the kernel table, which
    libsoda_hip.so reads back from the loaded blob.
"""
import os
import re
import subprocess
import tempfile

from .. import __version__
from . import (kernel_common, kernel_stage, kernel_stream2d, kernel_stream2d_wp,
               kernel_stream3d, kernel_stream3d_blk, kernel_stream3d_wp)
from . import spec as specmod

DEFAULT_MAX_DEPTH = 12
# Wavefronts per strip for deep fused 2-D kernels: 0 = single-wave form only,
# N > 1 = always the wave-pipelined form (kernel_stream2d_wp), -1 = choose: the
# wave-pipelined form (4 wavefronts, packed pairs when the program allows) where
# the single-wave form would need more than AUTO_WP_VGPRS registers or does not
# fit at all.  Measured on MI355X at 16384x16384 (tools/tune.py): jacobi2d depth
# 12 (188 VGPRs) single-wave 543 us vs 545-570 us pipelined; seidel2d depth 12
# (236 VGPRs) 684 us vs 588 us; blur depth 8 (single-wave impossible, depth 4 is
# its limit) 4402 vs 3499 Gcell/s.
WAVE_GROUPS = -1
AUTO_WP_VGPRS = 200
AUTO_WP_GROUPS = 4
AUTO_WP_BUDGET = 200
AUTO_WP_RING = 6               # LDS ring slots of the packed form (4 rows in flight)
AUTO_WP_SQUEEZE_VGPRS = 152
# Depths beyond 12 for programs the packed wave-pipelined form covers.  16 runs at
# four workgroups per CU (128 VGPRs, 12-slot ring); 20 and 24 keep 5 / 6 levels
# per wavefront and are capped at 168 VGPRs = three workgroups per CU.  Measured
# per iteration, jacobi2d, same box (tools/chunk_sweep.py; us at 16384^2 / 8192^2 /
# the 16384 x 2432 slab of an 8-GPU run): depth 16 36.7 / 9.89 / 6.27, depth 20
# 34.6 / 9.46 / 5.90, depth 24 33.2 / 9.17 / 5.83.  Past 24 the four-wavefront
# form spills (depth 28: 49.7), 5 or 6 wavefronts per workgroup load the four
# SIMDs unevenly (depth 20 on 5: 41.7, depth 32 on 6: 47.2) and 7 or 8
# (depth 28 / 32: 33.9 / 34.0) are no better than depth 24 on four.
PACKED_DEEP_DEPTHS = (16, 20, 24)
PACKED_DEEP_DEPTH = PACKED_DEEP_DEPTHS[0]
PACKED_DEEP_WAVES_PER_EU = 3      # for the depths above 16
# packable programs: the packed + ring form also replaces the single-wave form
# from this depth on (jacobi2d 16384^2 per launch: depth 12 570 vs 617 us, depth 8
# 501 vs 470 us - the single-wave form stays below)
PACKED_FROM_DEPTH = 12
WAVE_PIPELINE_MIN_DEPTH = 4
# Shallow fused kernels are HBM-bound: their strips start and end on 128-byte
# lines (kernel_stream2d.geometry, align='full'; +7..10 % measured at depth 1-2,
# nothing at 4, a loss from depth 8 on where the VALU bounds and the extra halo
# columns cost more than the alignment saves).
ALIGN_FULL_MAX_DEPTH = 2
# ... and their stores bypass the caches on arrays beyond the Infinity Cache
NT_AUTO_MAX_DEPTH_2D = 2
# rows a wavefront of the seam-free depth-1 form keeps in flight (16384^2 with two
# workgroups per CU, 3 / 6 / 9 rows: jacobi2d 419 / 414 / 415 us, sobel2d 270 / 253 / 242)
EXACT_PREFETCH = 6
# 3-D: depths beyond the single-wave form, built wave-pipelined (kernel_stream3d_wp)
DEEP_3D_DEPTHS = (4,)
BLOCK_3D_SHALLOW_DEPTHS = (1, 2)
# 3-D programs light on arithmetic get TWO depth-4 kernels: the wave-pipelined one
# (64 x 32 tiles, three workgroups per CU) and the block form (128 x 64 tiles, one
# 8-wavefront workgroup per CU, input planes prefetched one ahead); the run-time
# prices both per launch.  jacobi3d per depth-4 launch, same call: 512^3 375 vs
# 314 us, 256^3 49 vs 55, 128^3 27 vs 45; cfg5 (boxes 504^3 .. 112^3) 6.44 vs 6.04
# ms with either alone, 5.87 ms with the per-launch choice.  heat3d (packed pair-rows
# in the wave-pipelined form): 392 vs 378 us per 512^3 launch.
DEEP_3D_FORM = 'both'
BLOCK_3D_OPTIONS = dict(stack=8, prefetch=1, vgpr_budget=300, nt=4, mask_loads=1,
                        wide_stores=2, edge=1)
# The block form's input planes: through a two-slot LDS ring (LDS-direct loads, no
# prefetch registers) where the program's edge rows leave the LDS for it, else one
# plane ahead in registers.  Per depth-4 launch inside the 512^3 array of cfg5 (same
# chunking): box 496 242 vs 247 us, 456 205 vs 219, 416 152 vs 166, 352 117 vs 130,
# 256 53 vs 74; heat3d 512^3 x20, block form alone: 1.94 vs 1.96 ms
# (profiles/r03_blk_variants.txt); the ring's freed registers make room for packed
# pair-rows in heavy programs.
# mask_loads: ragged tiles fetch only what a stored cell depends on (the buffer form of the
# LDS-direct load; cfg5 per launch: box 424 161 -> 157 us, 400 140 -> 135, 368 128 -> 120,
# 360 118 -> 108; boxes that fill their tiles unchanged; the sweep -1.5 %)
# wide_stores: row segments leave in whole 64-byte pieces - the cells between the box
# and the next 64-byte boundary (unspecified by contract, read by nobody) are stored
# along, in launches beyond the Infinity Cache (2; 1 = always: boxes of 232^3 .. 336^3
# +1 %): cfg5 with this form alone 4.95 -> 4.79 ms, box 496 225 -> 210 us, 440 172 -> 156
# lean_fill: the levels a chunk's first steps do not need are skipped there (a second,
# guarded copy of the row loop for the trips at a chunk's ends; round 4): cfg5 with this
# form alone 4.81 -> 4.66 ms, per launch box 400 130.5 -> 125.6 us, 256 55.0 -> 48.8,
# 224 42.8 -> 37.3 (profiles/r04_blk_lean_fill.txt)
# edge: the first and last tile of a row store the 8 valid columns the 64-byte alignment
# drops (kernel_stream3d_blk.emit; soda_hip_kernel.edge_slack): cfg5's boxes 464, 456, 336,
# 328, 240, 232 and 112 lose a tile column (round 6)
BLOCK_3D_RING_OPTIONS = dict(stack=8, prefetch=0, ring=2, vgpr_budget=300, nt=4,
                             mask_loads=1, wide_stores=2, lean_fill=1, edge=1)
# ... for programs light enough on arithmetic: jacobi3d (weight 7) 417 us per
# depth-4 launch against 2 x 374 us at depth 2, heat3d (15) 622 us against
# 2 x 411 us; heavier programs are VALU-bound at depth 2 already
DEEP_3D_MAX_WEIGHT = 20
PACKED_3D_LIGHT_WEIGHT = 10
# ... and only programs that are light on arithmetic (denoise2d, ~70 weighted
# operations per cell, is VALU-bound at depth 1 and loses 19 % to the narrower
# aligned strips; blur 20, sobel2d 28, jacobi2d 5 gain)
ALIGN_FULL_MAX_WEIGHT = 40

# generator options of the fused 2-D forms that `generate` passes through:
# those both forms understand, and those only the wave-pipelined form has
SHARED_2D_OPTIONS = ('skip_fill', 'vgpr_budget', 'max_period', 'align', 'waves_per_eu')
WP_ONLY_OPTIONS = ('pairs', 'ring', 'split', 'prio', 'seam_probe')

HIPCC_FLAGS = ['-x', 'hip', '--offload-arch=gfx950', '--cuda-device-only',
               '--no-gpu-bundle-output', '-O3', '-ffp-contract=off',
               '-fno-slp-vectorize', '-fwrapv', '-std=c++17']


_DOUBLE_LITERAL = re.compile(
    r'(?<![\w.])(?:\d+\.\d*|\.\d+|\d+(?=[eE]))(?:[eE][+-]?\d+)?(?![\w.])')


def dpp_combine_is_safe(spec):
  """True for programs whose arithmetic is float32 only: every tensor `float`,
  no double literal, no math call (those are the C double functions), no cast."""
  if any(t != 'float' for t in specmod.tensor_c_types(spec).values()):
    return False
  for stage in spec['stages']:
    for text in [stage['expr']] + [l['expr'] for l in stage['lets']]:
      plain = specmod.LOAD_RE.sub('x', text)
      if _DOUBLE_LITERAL.search(plain) or kernel_common.used_functions(
          dict(stages=[dict(expr=text, lets=[])])) or 'static_cast<' in plain:
        return False
  return True


def extra_flags(spec):
  """Per-program compiler flags beyond HIPCC_FLAGS.

  ROCm 7.2's DPP-combine pass (GCNDPPCombine) is only trusted on pure-float32
  programs.  Found by the parity tests:
    * INTEGER subtraction whose operand is a wave-shift DPP move is miscompiled
      (`v_mov_b32_dpp` folded into `v_sub[rev]_u32/_u16`: every lane wrong;
      sobel2d at depth >= 2, `o = l - a(2,0) + a(-2,0)` on int32);
    * a DPP move folded into `v_cvt_f64_f32` (a float operand meeting a double
      literal) yields an instruction the verifier rejects: the compile fails.
  -O0 and -amdgpu-dpp-combine=false are correct in both cases.  Float32
  add/sub/mul with a DPP operand is verified bit-exact and the fusion is worth
  ~10 % on the jacobi kernels, so float32-only programs keep the pass."""
  tuning = []
  if spec['dim'] == 2 and \
      kernel_stream2d_wp.packable(specmod.inline_pointwise(spec)):
    # The packed kernels run at a register cap under which the default scheduler
    # serialises whole level-rows on one accumulator pair; the ILP-first strategy
    # keeps the four cells of a row interleaved.  Scheduling only: same results.
    # Measured per launch on 16384^2: jacobi2d depth 16 566 -> 557 us, depth 12
    # 508 -> 505, seidel2d depth 16 620 -> 615, depth 8 506 -> 489; the other
    # depths within +-0.5 %.  Not for 3-D programs: hipcc then gives the packed
    # heat3d kernel 276 registers instead of 230 (one workgroup per CU, 647 us
    # per launch instead of 400).
    tuning = ['-mllvm', '-amdgpu-sched-strategy=max-ilp']
  if dpp_combine_is_safe(spec):
    return tuning
  return ['-mllvm', '-amdgpu-dpp-combine=false'] + tuning


FLAGS_MARK = '// SODA-HIP-FLAGS:'


def flags_from_text(text):
  """Extra flags recorded in generated kernel text (for run-time compiles)."""
  for line in text.splitlines()[:400]:
    if line.startswith(FLAGS_MARK):
      return line[len(FLAGS_MARK):].split()
  return []


# Issue cost of an operation in units of one f32 add (2 SIMD cycles per wave64
# instruction on gfx950).  Double precision runs at half the rate; a double division
# and a double square root compile to ~13 double instructions each (denoise2d's
# `1.0f / sqrt(...)`: 12 v_fma_f64, 3 v_mul_f64, 2 v_div_scale_f64, v_rsq_f64,
# v_rcp_f64, v_div_fmas_f64, v_div_fixup_f64, 2 v_ldexp_f64, 2 conversions per cell);
# a float division to ~10 single ones; an integer division by a literal to a
# multiply-high and shifts.  Other math calls are the double library functions.
OP_COST = dict(f32=1, f64=2, div_f32=10, div_f64=30, div_int=8, sqrt=30, cheap_call=4,
               call=60)
_CHEAP_CALLS = ('fabs', 'fmax', 'fmin', 'floor', 'ceil', 'trunc', 'round', 'rint',
                'nearbyint', 'copysign', 'abs', 'min', 'max', 'select')


def arithmetic_weight(spec):
  """VALU cost of one iteration per cell, in f32-add equivalents: the operators of
  all stages priced by OP_COST.  An expression is priced as DOUBLE arithmetic when
  it holds a double literal, a math call (the C double functions) or a
  double tensor - as the C++ promotion rules make (most of) it."""
  types = specmod.tensor_c_types(spec)
  weight = 0
  for stage in spec['stages']:
    texts = [(let['c_type'], let['expr']) for let in stage['lets']] + \
        [(stage['c_type'], stage['expr'])]
    for c_type, source in texts:
      operands = [types.get(t) for t, _ in specmod.LOAD_RE.findall(source)]
      text = re.sub(r'\{[^}]*\}', 'L', kernel_common.device_expr(source))
      calls = re.findall(r'soda_fn_(\w+)', text)
      heavy = [c for c in calls if c not in _CHEAP_CALLS]
      double = c_type == 'double' or 'double' in operands or bool(heavy) or \
          bool(_DOUBLE_LITERAL.search(text))
      integer = not double and c_type not in ('float', 'double', '_Float16')
      unit = OP_COST['f64'] if double else OP_COST['f32']
      weight += unit * len(re.findall(r'(?<=[\w)\s])[-+*](?=[\s\w(])', text))
      weight += text.count('/') * OP_COST[
          'div_f64' if double else 'div_int' if integer else 'div_f32']
      for c in calls:
        weight += OP_COST['sqrt'] if c == 'sqrt' else \
            OP_COST['cheap_call'] if c in _CHEAP_CALLS else OP_COST['call']
  return weight


def default_cols(spec):
  """Columns per lane: the widest vector (<= 16 bytes) the DSL's burst width
  allows; `burst width: 512` (64 bytes per FPGA burst) gives 16-byte lanes."""
  elem = specmod.ELEM_SIZE[spec['inputs'][0]['c_type']]
  vec = max(elem, min(16, spec['burst_width'] // 8))
  return max(1, vec // elem)


def fused_depths(spec, max_depth):
  if len(spec['inputs']) == 1 and len(spec['outputs']) == 1 and \
      spec['inputs'][0]['c_type'] == specmod.tensor_c_types(spec)[spec['outputs'][0]]:
    # no point in a kernel deeper than the program iterates
    # 12 is the measured sweet spot for 4-byte 5-point programs on MI355X
    # (two waves per SIMD at ~200 VGPRs; 16 drops to one wave, 8 is HBM-bound)
    return [d for d in (1, 2, 4, 8, 12)
            if d <= max_depth and d <= max(1, spec['iterate'])]
  return [1]


# What a workgroup of the wave-pipelined forms spends per streamed row besides
# arithmetic (barrier, ring and hand-off traffic), as VALU-cycle equivalents
# summed over its wavefronts.  Fitted to jacobi2d at 16384^2, us per step of a full
# chip: depth 16 0.98 (four workgroups per CU), 20 0.84, 24 0.96 (three per CU)
# = workgroups per CU x (0.010 us x depth + 0.080 us).
WP_STEP_FIXED_CYCLES = 640


CALIBRATION_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)),
                                'calibration.json')
CALIBRATED_FIELDS = ('step_ns_full', 'step_ns_one', 'stream_gbps', 'fade_lo_mib',
                     'fade_hi_mib', 'stream_chunk')
# ... and one the generator sets a default for, which a measurement may replace (the cap
# on workgroups per CU that goes with the measured chunk)
MEASURED_OVER_DEFAULT = ('stream_wgs_per_cu',)
_calibration = None


def calibration_key(entry, spec):
  """Identity of a generated kernel for the calibration table: the program it
  computes and every figure of its table entry except the cost figures themselves -
  a change of the generator that changes the kernel's shape makes its measurement
  stale (the scheduler then falls back to the model) until tools/calibrate.py has
  run again."""
  import hashlib
  import json
  # (edge_slack changes which tiles a launch has, not what a workgroup's step costs: the
  # streamed loop is the same text with and without it)
  shape = {k: v for k, v in entry.items()
           if k not in CALIBRATED_FIELDS + MEASURED_OVER_DEFAULT +
           ('step_valu', 'step_bytes', 'edge_slack')}
  digest = hashlib.sha1(json.dumps(shape, sort_keys=True).encode()).hexdigest()[:12]
  return '%s/%s/%s' % (kernel_common.program_hash(spec)[:12], entry['name'], digest)


def calibration():
  global _calibration
  if _calibration is None:
    import json
    try:
      with open(CALIBRATION_FILE) as f:
        _calibration = json.load(f).get('kernels', {})
    except (OSError, ValueError):
      _calibration = {}
  return _calibration


def annotate_cost(entry, spec):
  """Cost figures of a streaming kernel for the run-time scheduler
  (soda_hip_kernel.step_valu / step_bytes, include/soda_hip.h): VALU issue
  cycles and HBM bytes of ONE workgroup per streamed row or plane.  A wave64
  instruction on `n` lane-operations' worth of cells costs 2 cycles per cell it
  covers per lane (a packed instruction covers two cells in 4 cycles: the same)."""
  elem = specmod.ELEM_SIZE[spec['inputs'][0]['c_type']]
  weight = max(1, arithmetic_weight(spec))
  waves = entry['block'][0] // kernel_stream2d.LANES
  n_in, n_out = len(spec['inputs']), len(spec['outputs'])
  if spec['dim'] == 2 and 'groups' in entry:      # wave-pipelined: one strip(-pair)
    lane_cells = entry['cols'] * (2 if entry.get('pairs') else 1)
    valu = entry['depth'] * weight * lane_cells * 2 + WP_STEP_FIXED_CYCLES
    cells_in, cells_out = kernel_stream2d.LANES * lane_cells, entry['tile'][0]
  elif spec['dim'] == 2:                          # one strip per wavefront
    valu = waves * entry['depth'] * weight * entry['cols'] * 2
    cells_in = waves * kernel_stream2d.LANES * entry['cols']
    cells_out = entry['tile'][0]
  elif 'groups' in entry:                         # 3-D, one level per wavefront
    lane_cells = entry['rows'] * entry['cols']
    valu = entry['depth'] * weight * lane_cells * 2 + WP_STEP_FIXED_CYCLES
    cells_in = kernel_stream2d.LANES * lane_cells
    cells_out = entry['tile'][0] * entry['tile'][1]
  else:                                           # 3-D, one tile per wavefront
    lane_cells = entry['rows'] * entry['cols']
    valu = waves * entry['depth'] * weight * lane_cells * 2
    cells_in = waves * kernel_stream2d.LANES * lane_cells
    cells_out = entry['tile'][0] * entry['tile'][1]
  entry['step_valu'] = int(valu)
  entry['step_bytes'] = int((n_in * cells_in + n_out * cells_out) * elem)
  # measured step times of exactly this kernel, when it has been calibrated
  # (soda_hip_kernel.step_ns_full / step_ns_one / stream_gbps)
  measured = calibration().get(calibration_key(entry, spec))
  if measured:
    for field in CALIBRATED_FIELDS:
      entry[field] = int(measured.get(field, 0))
    for field in MEASURED_OVER_DEFAULT:
      if field in measured:
        entry[field] = int(measured[field])
  return entry


def prefixed_options(options, prefix, emit):
  """The `<prefix>name=value` entries of `options` as keyword arguments of `emit`;
  a name `emit` does not take is an error, not a silently ignored knob."""
  import inspect
  known = inspect.signature(emit).parameters
  out = {}
  for key, value in options.items():
    if key.startswith(prefix):
      if key[len(prefix):] not in known:
        raise TypeError('%s: %s() has no option `%s`' % (key, emit.__module__,
                                                         key[len(prefix):]))
      out[key[len(prefix):]] = value
  return out


def generate(spec, max_depth=None, cols=None, chunk_rows=None, prefetch=None,
             fused=True, depths=None, inline=True, wave_groups=None,
             **fused_options):
  """Returns (kernel text, kernel table).  `depths` overrides the default set
  of fused depths (depth 1 is always included: the scheduler needs it)."""
  max_depth = DEFAULT_MAX_DEPTH if max_depth is None else max_depth
  # kernels are generated from the LOWERED program (pointwise-only locals folded
  # into their readers); the blob is still identified by the source program
  source = spec
  spec = specmod.inline_pointwise(spec) if inline else spec
  parts = [kernel_common.prelude(source, __version__)]
  if extra_flags(source):
    parts.append('%s %s\n' % (FLAGS_MARK, ' '.join(extra_flags(source))))
  wrappers = kernel_common.math_wrappers(kernel_common.used_functions(spec))
  if wrappers:
    parts.append('// math calls resolve as in the reference CPU path: C double '
                 'functions\n' + '\n'.join(wrappers) + '\n')
  text, table = kernel_stage.emit(spec)
  parts.append(text)
  notes = []
  if fused and spec['dim'] == 2:
    wanted = list(fused_depths(spec, max_depth))
    # one level deeper for programs the packed wave-pipelined form covers: it
    # is the only form with the registers for it and, fed through the LDS ring,
    # the only one that gains from it (jacobi2d 16384^2: depth 12 single-wave
    # 46.0 us per iteration, depth 16 packed + ring 39.5)
    if (len(wanted) > 1 and max_depth >= DEFAULT_MAX_DEPTH and
        len(spec['inputs']) == len(spec['outputs']) == 1 and
        (WAVE_GROUPS if wave_groups is None else wave_groups) == -1 and
        kernel_stream2d_wp.packable(spec)):
      wanted += [d for d in PACKED_DEEP_DEPTHS if spec['iterate'] >= d]
    if depths is not None:
      wanted = sorted(set([1] + [d for d in depths if len(wanted) > 1 or d == 1]))
    for depth in wanted:
      groups = WAVE_GROUPS if wave_groups is None else wave_groups
      common = dict(cols=cols if cols else default_cols(spec),
                    chunk_rows=chunk_rows or 256,
                    prefetch=3 if prefetch is None else prefetch)
      if 'align' not in fused_options:
        # (deeper kernels keep the widest strips: strips on 64-byte pieces,
        # align='store64', gain 1.8 % on cfg2 in short runs and LOSE 1.3 % under the
        # bench protocol's sustained load - 0.949 vs 0.961 ms, three alternating pairs
        # in one call; cfg4 29.17 vs 29.22 ms)
        common['align'] = 'full' if (
            depth <= ALIGN_FULL_MAX_DEPTH and
            arithmetic_weight(spec) <= ALIGN_FULL_MAX_WEIGHT) else 'none'
      single = piped = None
      if groups <= 1 or depth < WAVE_PIPELINE_MIN_DEPTH or groups == -1:
        # the memory-bound depths store around the caches when a launch's box
        # does not fit the Infinity Cache (kernel_stream2d.emit: nontemporal)
        options = dict({'nontemporal': 4} if depth <= NT_AUTO_MAX_DEPTH_2D else {},
                       **{k: v for k, v in fused_options.items()
                          if k not in WP_ONLY_OPTIONS and k != 'nt'})
        # ... and where no STAGE is read across lanes (depth 1 of the samples) their
        # strips do not overlap at all (align='exact': whole 128-byte lines in and
        # out, the seam columns from one extra vector load per row and side)
        aligns = ['exact', 'full'] if common.get('align') == 'full' else \
            [options.pop('align', None) or common['align']]
        for k, how in enumerate(aligns):
          shape = dict(common, align=how)
          if how == 'exact' and prefetch is None:
            # six rows in flight per wavefront (two workgroups per CU on the arrays
            # that matter, soda_hip_kernel.stream_wgs_per_cu)
            shape['prefetch'] = EXACT_PREFETCH
          try:
            single = kernel_stream2d.emit(spec, depth, **shape, **options)
            break
          except kernel_stream2d.NotFusable as e:
            if k == len(aligns) - 1:
              notes.append('depth %d not fused: %s' % (depth, e))
      want_piped = depth >= WAVE_PIPELINE_MIN_DEPTH and (
          groups > 1 or (groups == -1 and (
              single is None or single[1]['est_vgprs'] > AUTO_WP_VGPRS or
              (depth >= PACKED_FROM_DEPTH and kernel_stream2d_wp.packable(spec)))))
      if want_piped:
        options = {k: v for k, v in fused_options.items()
                   if k in SHARED_2D_OPTIONS + WP_ONLY_OPTIONS}
        if groups == -1:
          options.setdefault('vgpr_budget', AUTO_WP_BUDGET)
          lane_bytes = common['cols'] * specmod.ELEM_SIZE[spec['inputs'][0]['c_type']]
          if lane_bytes == 16:
            # input rows through the LDS ring: no prefetch registers (see
            # kernel_stream2d_wp.emit); needs 16-byte lanes
            options.setdefault('ring', AUTO_WP_RING)
          # packable programs: one 512-column strip per wavefront, its two halves
          # sharing register pairs (pairs=2; jacobi2d depth 16 per launch on
          # 16384^2: 597 us against 627 us for two 256-column strips, pairs=1)
          options.setdefault('pairs', 0 if not kernel_stream2d_wp.packable(spec)
                             else 2 if options.get('ring') else 1)
          if depth > PACKED_DEEP_DEPTH and options['pairs'] == 2:
            # 5-6 levels per wavefront: three workgroups per CU (168 VGPRs;
            # depth 24 left alone takes 174 = two per CU: 34.1 vs 33.2 us per
            # iteration at 16384^2)
            options.setdefault('waves_per_eu', PACKED_DEEP_WAVES_PER_EU)
        try:
          piped = kernel_stream2d_wp.emit(
              spec, depth, groups=AUTO_WP_GROUPS if groups == -1 else groups,
              **common, **options)
          squeeze = (groups == -1 and options.get('ring') and
                     'waves_per_eu' not in options and
                     128 < piped[1]['est_vgprs'] <= AUTO_WP_SQUEEZE_VGPRS)
          if squeeze and options.get('pairs') == 2 and options['ring'] == AUTO_WP_RING \
              and 'max_period' not in options:
            # a few registers over four workgroups per CU: cap them at 128.  In
            # the wide form that pays only together with a 12-slot ring (10 rows
            # in flight) unrolled over 12 rows: jacobi2d depth 16, 16384^2, per
            # launch 578 us uncapped with ring 6, 830 us capped with ring 6,
            # 560 us capped with ring 12.
            piped = kernel_stream2d_wp.emit(
                spec, depth, groups=AUTO_WP_GROUPS, waves_per_eu=4, **common,
                **dict(options, ring=2 * AUTO_WP_RING, max_period=2 * AUTO_WP_RING))
          elif squeeze and not options.get('pairs'):
            # the scalar form: let the compiler spill the few registers over the cap
            # (not the two-strip packed form: its scalar DPP adds then spill 45
            # registers and lose 60 %)
            piped = kernel_stream2d_wp.emit(
                spec, depth, groups=AUTO_WP_GROUPS, waves_per_eu=4, **common,
                **options)
        except kernel_stream2d.NotFusable as e:
          notes.append('depth %d not wave-pipelined: %s' % (depth, e))
          if single is None and groups > 1:
            try:
              single = kernel_stream2d.emit(
                  spec, depth, **common,
                  **{k: v for k, v in fused_options.items()
                     if k not in WP_ONLY_OPTIONS and k != 'nt'})
            except kernel_stream2d.NotFusable as e2:
              notes.append('depth %d not fused: %s' % (depth, e2))
      if piped is None and single is None:
        continue
      ftext, entry = piped if piped is not None else single
      parts.append(ftext)
      table.append(annotate_cost(entry, spec))
  if fused and spec['dim'] == 3:
    wanted3 = [d for d in (1, 2) if d <= max(1, spec['iterate'])] \
        if len(spec['inputs']) == len(spec['outputs']) == 1 else [1]
    if depths is not None:
      wanted3 = sorted(set([1] + list(depths))) if len(wanted3) > 1 else [1]
    for depth in wanted3:
      # rows per lane: as many as the register file allows (taller tiles waste
      # less on the y halo)
      options = {k: v for k, v in fused_options.items()
                 if not k.startswith(('wp_', 'blk_')) and
                 k not in ('deep3d', 'deep3d_from', 'nontemporal')}
      # (rows, columns) per lane: the tallest tile the register file allows (taller
      # tiles waste less on the y halo).  Programs with several live tensors
      # (denoise3d, lowered to g and output over the inputs f and u) fit with one
      # column per lane: 12 rows (156 VGPRs, three wavefronts per SIMD) before 16
      # (235 VGPRs, two) - denoise3d per sweep at 256^3 / 512^3: 160 / 971 us against
      # 200 / 999 us, the two per-stage launches 165 / 1281 us
      narrow = [] if cols or len(spec['inputs']) == len(spec['outputs']) else \
          [(12, 1), (16, 1)]      # (iteration chains go deep in the forms below)
      shapes = [(options.pop('rows'), cols or 2)] if 'rows' in options else \
          [(16, cols or 2), (12, cols or 2)] + narrow
      error = None
      options.setdefault('nt', 4)
      for rows, lane_cols in shapes:
        try:
          ftext, entry = kernel_stream3d.emit(spec, depth, rows=rows, cols=lane_cols,
                                              **options)
        except kernel_stream2d.NotFusable as e:
          error = e
          continue
        parts.append(ftext)
        table.append(annotate_cost(entry, spec))
        error = None
        break
      if error is not None:
        notes.append('depth %d not fused: %s' % (depth, error))
    # deeper than one wavefront's registers allow: one level per wavefront
    # The block form serves the shallow depths as well (deep3d_from=3: not): next to the
    # single-wave kernels above, which stay for arrays below its 128 x 64 tile and for
    # programs it does not take.  jacobi3d per launch, depth 1 / 2: 512^3 470 / 395 us
    # single-wave, 221 / 222 us block form (one read and one write of the array at 5.3
    # TB/s: 203 us); 256^3 67 / 65 against 33 / 35; heat3d 512^3 depth 2 464 -> 231.
    deep_from = fused_options.get('deep3d_from', 1)
    deep = [d for d in sorted(set(depths if depths is not None else
                                  BLOCK_3D_SHALLOW_DEPTHS + DEEP_3D_DEPTHS))
            if d >= deep_from and d <= max(1, spec['iterate'])]
    if len(spec['inputs']) == len(spec['outputs']) == 1 and (
        depths is not None or arithmetic_weight(spec) <= DEEP_3D_MAX_WEIGHT):
      for depth in deep:
        form = fused_options.get('deep3d', DEEP_3D_FORM)
        if depth < 3 and form == 'wp':      # one level per wavefront needs >= 3 levels
          continue
        if form in ('blk', 'both'):
          # block form: all levels in every wavefront, edge rows through LDS
          # (kernel_stream3d_blk).  Named <app>_fused_k<d>b; with 'both' it ships
          # NEXT TO the wave-pipelined kernel and the run-time picks per launch
          given = prefixed_options(fused_options, 'blk_', kernel_stream3d_blk.emit)
          heavy = arithmetic_weight(spec) > PACKED_3D_LIGHT_WEIGHT
          ring_forms = [dict(BLOCK_3D_RING_OPTIONS)]
          if heavy and kernel_stream2d_wp.packable(spec):
            # heavy plain-float programs: packed pair-rows first (heat3d 512^3 x20,
            # block form alone, clocks warm: 1.76 ms plain, 1.42 packed - 13.2 k instead
            # of 22.6 k VALU instructions per unrolled loop, 254 VGPRs without spills
            # now that the ring took the prefetch registers; a hand-ordered scalar
            # instruction stream, round 3's kernel_asm, reached 1.49 and is gone)
            # (and exact store ranges: whole 64-byte pieces cost this kernel 3 %,
            # heat3d 512^3 x20 1.37 -> 1.41 ms, where they gain jacobi3d's 2 %)
            ring_forms.insert(0, dict(BLOCK_3D_RING_OPTIONS, pairs=1, wide_stores=0))
          attempts = [BLOCK_3D_OPTIONS] if 'prefetch' in given or 'ring' in given else \
              ring_forms + [BLOCK_3D_OPTIONS]
          try:
            for k, base in enumerate(attempts):
              options = dict(base)
              options.update(given)
              try:
                ftext, entry = kernel_stream3d_blk.emit(spec, depth, **options)
                break
              except kernel_stream2d.NotFusable:
                if k == len(attempts) - 1:
                  raise
            parts.append(ftext)
            table.append(annotate_cost(entry, spec))
            if form == 'blk' or depth < 3:
              continue
          except kernel_stream2d.NotFusable as e:
            notes.append('depth %d not in block form: %s' % (depth, e))
            if depth < 3:
              continue
        options = prefixed_options(fused_options, 'wp_', kernel_stream3d_wp.emit)
        options.setdefault('groups', min(depth * len(spec['stages']), 4))
        # (no non-temporal stores here: heat3d 512^3 x20 with this form alone 2.09 ms
        # without, 2.12 ms with wp_nt=4)
        if options.get('split', 2) == 2 and not options.get('loader') and \
            options.get('rows', 16) % 2 == 0 and kernel_stream2d_wp.packable(spec):
          # packed pair-rows (v_pk_*_f32) for programs heavy enough on arithmetic:
          # heat3d (weight 15) 459 us per launch scalar, 383-440 us packed at the
          # 230 VGPRs the compiler asks for (629 us capped at three workgroups per
          # CU: 360 spilled registers).  Light programs are bounded by memory and
          # keep the scalar form at three workgroups per CU: jacobi3d (weight 7),
          # cfg5 with the offline compiler: 6.23 ms scalar, 7.43 ms packed (36
          # spilled registers); hiprtc happens to favour the packed form (6.31 vs
          # 6.63 ms) - the shipped code objects are built by hipcc.
          options.setdefault('pairs', int(arithmetic_weight(spec) >
                                          PACKED_3D_LIGHT_WEIGHT))
          if options['pairs']:
            options.setdefault('waves_per_eu', 0)
        try:
          ftext, entry = kernel_stream3d_wp.emit(spec, depth, **options)
        except kernel_stream2d.NotFusable as e:
          notes.append('depth %d not wave-pipelined: %s' % (depth, e))
          continue
        parts.append(ftext)
        table.append(annotate_cost(entry, spec))
  if notes:
    parts.append(''.join('// %s\n' % n for n in notes))
  parts.append(kernel_common.meta_symbol(
      spec, table, extra=dict(program_hash=kernel_common.program_hash(source),
                              source_stages=[s['name'] for s in source['stages']])))
  return '\n'.join(parts), table


def compile_to_code_object(text, out_path, hipcc=None, extra_flags=()):
  """Offline build of the blob (the role `--xocl-hw-xo` plays in the reference:
  invoking the vendor tool chain from the compiler driver)."""
  hipcc = hipcc or os.environ.get('HIPCC') or '/opt/rocm/bin/hipcc'
  with tempfile.NamedTemporaryFile('w', suffix='.hip', delete=False) as f:
    f.write(text)
    src = f.name
  try:
    subprocess.check_call([hipcc] + HIPCC_FLAGS + flags_from_text(text) +
                          list(extra_flags) + [src, '-o', out_path])
  finally:
    os.unlink(src)
  return out_path
