"""Code generation on CPU: sodac CLI surface, printers, kernel text compiles for
gfx950 (hipcc cross-compiles without a GPU), metadata round-trips."""
import json
import os
import subprocess
import sys

import pytest

from soda_hip import frontend
from soda_hip.codegen import backend, kernel, kernel_common, kernel_stream2d
from soda_hip.codegen import spec as specmod
from soda_hip.runtime import host

from conftest import ROOT, SAMPLES

SODAC = os.path.join(ROOT, 'soda-compiler_amd', 'sodac')
APPS = ('blur', 'jacobi2d', 'jacobi3d', 'seidel2d', 'heat3d', 'sobel2d',
        'denoise2d', 'denoise3d')


def spec_of(app, **kw):
  return specmod.spec_from_stencil(
      frontend.load(os.path.join(SAMPLES, app + '.soda'), **kw))


def run_sodac(*argv, **kw):
  return subprocess.run([sys.executable, SODAC] + list(argv), capture_output=True,
                        text=True, **kw)


def test_sodac_writes_all_artifacts(tmp_path):
  r = run_sodac(os.path.join(SAMPLES, 'jacobi2d.soda'), '--hip-kernel',
                str(tmp_path / 'k.hip'), '--hip-host', str(tmp_path / 'h.py'),
                '--hip-header', str(tmp_path / 'h.h'), '--iterate', '12')
  assert r.returncode == 0, r.stderr
  text = (tmp_path / 'k.hip').read_text()
  assert 'jacobi2d_fused_k12' in text and 'jacobi2d_stage_t0' in text
  meta = kernel_common.read_meta_from_source(text)
  assert meta['app_name'] == 'jacobi2d' and meta['abi'] == kernel_common.ABI_VERSION
  assert meta['program_hash'] == kernel_common.program_hash(spec_of('jacobi2d'))
  assert [k['depth'] for k in meta['kernels'] if k['kind'] == 'fused'] == \
      [1, 2, 4, 8, 12]
  shim = (tmp_path / 'h.py').read_text()
  assert 'def jacobi2d(var_t1_buffer, var_t0_buffer, blob' in shim
  assert 'def jacobi2d_test(blob, dims' in shim
  compile(shim, 'h.py', 'exec')
  header = (tmp_path / 'h.h').read_text()
  assert 'int jacobi2d(buffer_t* var_t1_buffer, buffer_t* var_t0_buffer, ' \
         'const char* blob)' in header
  assert 'int jacobi2d_test(const char* blob, const int dims[4]);' in header
  # the header is valid C
  (tmp_path / 'x.c').write_text('#include "h.h"\nint main(void){return 0;}\n')
  subprocess.check_call(['gcc', '-fsyntax-only', '-I', str(tmp_path),
                         str(tmp_path / 'x.c')])


def test_sodac_stdout_and_stdin():
  with open(os.path.join(SAMPLES, 'blur.soda')) as f:
    r = run_sodac('-', '--hip-kernel', '-', stdin=f)
  assert r.returncode == 0
  assert 'blur_fused_k1' in r.stdout and 'soda_hip_meta' in r.stdout


def test_sodac_error_exit_codes(tmp_path):
  bad = tmp_path / 'bad.soda'
  bad.write_text('kernel: k\nburst width: 64\nunroll factor: 1\niterate: 1\n'
                 'input float: a(8, *)\noutput float: b(0, 0) = a(0, 0) +\n')
  r = run_sodac(str(bad), '--hip-kernel', '-')
  assert r.returncode == 1 and 'syntax error' in r.stderr
  r = run_sodac(os.path.join(SAMPLES, 'blur.soda'), '--iterate', '0',
                '--hip-kernel', '-')
  assert r.returncode == 1 and 'cannot iterate 0 times' in r.stderr
  odd = tmp_path / 'odd.soda'
  odd.write_text('kernel: k\nburst width: 64\nunroll factor: 1\niterate: 1\n'
                 'input uint5: a(8, *)\noutput uint5: b(0, 0) = a(0, 0)\n')
  r = run_sodac(str(odd), '--hip-kernel', '-')
  assert r.returncode == 1 and 'no native GPU representation' in r.stderr


def test_reference_flags_are_accepted():
  r = run_sodac(os.path.join(SAMPLES, 'jacobi2d.soda'), '--burst-width', '256',
                '--unroll-factor', '8', '--tile-size', '2000', '--dram-in', '1',
                '--dram-out', '2', '--iterate', '3', '-v', '--hip-kernel', '-')
  assert r.returncode == 0, r.stderr
  meta = kernel_common.read_meta_from_source(r.stdout)
  assert meta['spec']['iterate'] == 3 and meta['spec']['tile_size'] == [2000, 0]
  assert meta['spec']['burst_width'] == 256
  # the FPGA knobs are carried but unused, and sodac says so (once, as a warning)
  assert 'ignores --unroll-factor, --tile-size, --dram-in, --dram-out' in r.stderr
  assert r.stderr.count('WARNING') == 1
  quiet = run_sodac(os.path.join(SAMPLES, 'jacobi2d.soda'), '--iterate', '3',
                    '--hip-kernel', '-')
  assert quiet.returncode == 0 and 'WARNING' not in quiet.stderr


@pytest.mark.parametrize('app', APPS)
def test_every_sample_compiles_for_gfx950(app, tmp_path):
  """The reference's only test pipes each sample through the kernel printer and
  a syntax check (tests/test-compilation.sh); this is its HIP analogue, with a
  real gfx950 compile."""
  spec = spec_of(app, iterate=4 if app in ('jacobi2d', 'seidel2d') else None)
  text, table = kernel.generate(spec, max_depth=4)
  out = tmp_path / (app + '.hsaco')
  kernel.compile_to_code_object(text, str(out))
  assert out.stat().st_size > 1000
  assert open(out, 'rb').read(4) == b'\x7fELF'
  # one per-stage kernel per stage of the LOWERED program (locals that are only
  # read at offset 0 are folded into their readers)
  stages = [k for k in table if k['kind'] == 'stage']
  assert len(stages) == len(specmod.inline_pointwise(spec)['stages'])


@pytest.mark.parametrize('options', [
    dict(wave_groups=4), dict(wave_groups=4, pairs=1, vgpr_budget=250),
    dict(wave_groups=4, pairs=2, vgpr_budget=250, ring=6),
    dict(wave_groups=4, pairs=2, vgpr_budget=250, ring=12, max_period=12)])
def test_wave_pipelined_forms_compile(options, tmp_path):
  """The experimental wave-pipelined form of the fused 2-D kernel (one wavefront
  per group of levels, LDS hand-off) and its packed-pair variant build for
  gfx950; the packed variant is refused for anything but plain float programs."""
  from soda_hip.codegen import kernel_stream2d, kernel_stream2d_wp
  spec = spec_of('jacobi2d', iterate=8)
  text, table = kernel.generate(spec, depths=[8], **options)
  fused = [k for k in table if k['kind'] == 'fused' and k['depth'] == 8]
  assert fused and fused[0]['groups'] == 4 and fused[0]['block'] == [256, 1, 1]
  assert fused[0]['pairs'] == options.get('pairs', 0)
  if options.get('pairs') == 1:     # two strips per wavefront
    assert fused[0]['tile'][0] == 2 * fused[0]['w_out'] == 2 * (256 - 16)
  if options.get('pairs') == 2:     # one strip of twice the width
    assert fused[0]['tile'][0] == fused[0]['w_out'] == 512 - 16
    assert fused[0]['min_extent'] == [512, 1] and 'pk_wide_below(' in text
  if options.get('pairs') == 1:     # lane-crossing operands as two scalar DPP adds
    assert 'pk_from_lane_below(' in text
  # options that were measured null twice are gone: asking for one is an error
  with pytest.raises(TypeError):
    kernel.generate(spec, depths=[8], wave_groups=4, sync=3)
  out = tmp_path / 'wp.hsaco'
  kernel.compile_to_code_object(text, str(out))
  assert open(out, 'rb').read(4) == b'\x7fELF'
  assert kernel_stream2d_wp.packable(spec)
  for app in ('blur', 'sobel2d', 'denoise2d'):
    assert not kernel_stream2d_wp.packable(specmod.inline_pointwise(spec_of(app)))
  with pytest.raises(kernel_stream2d.NotFusable):
    kernel_stream2d_wp.emit(specmod.inline_pointwise(spec_of('blur', iterate=8)), 8,
                            pairs=1)
  with pytest.raises(kernel_stream2d.NotFusable):    # wide strips need the ring
    kernel_stream2d_wp.emit(spec, 8, pairs=2)


def test_default_depth_sets():
  """jacobi2d and seidel2d (plain float programs) get depth-16, -20 and -24
  kernels (as many as fit the registers) in the packed wave-pipelined form, fed
  through the LDS ring and therefore limited to arrays at least one strip wide;
  integer programs stop at depth 12 (their wave-pipelined kernels use the ring
  too).  Every fused kernel carries the cost figures the scheduler prices it by."""
  table = kernel.generate(spec_of('jacobi2d', iterate=1000))[1]
  fused = {k['depth']: k for k in table if k['kind'] == 'fused'}
  assert sorted(fused) == [1, 2, 4, 8, 12, 16, 20, 24]
  assert all(k['step_valu'] > 0 and k['step_bytes'] > 0 for k in fused.values())
  assert fused[24]['tile'][0] == 464 and fused[20]['tile'][0] == 472
  # (strips on 64-byte pieces were measured: slower under sustained load; gone)
  assert fused[24]['origin_align'] == 4 and fused[2]['origin_align'] == 32
  # depth 1: seam-free strips (no stage is read across lanes), six rows in flight, at
  # most two workgroups per CU on arrays beyond the caches
  assert fused[1]['exact'] == 1 and fused[1]['halo'] == [0, 0] and fused[1]['tile'][0] == 1024
  assert fused[1]['prefetch'] == 6 and fused[1]['stream_wgs_per_cu'] in (0, 2)
  assert not fused[2].get('exact')
  # ... and the chunk length (with its cap) that tools/calibrate.py measured as the fastest
  # on a streaming array, for the shallow kernels only; the deep ones keep the launcher's
  for depth, k in fused.items():
    assert isinstance(k.get('stream_chunk', 0), int) and k.get('stream_chunk', 0) >= 0
    assert depth <= 4 or not k.get('stream_chunk')
  assert fused[24]['fill_rows'] == 51
  assert fused[12]['groups'] == 4 and fused[12]['pairs'] and not fused[8].get('groups')
  k16 = fused[16]
  # wide strips (512 columns per wavefront), 12-slot ring, four workgroups per CU
  assert k16['groups'] == 4 and k16['pairs'] == 2 and k16['ring'] == 12
  assert k16['min_extent'] == [512, 1] and k16['block'] == [256, 1, 1]
  assert k16['tile'][0] == 480 and fused[12]['ring'] == 6
  table = kernel.generate(spec_of('jacobi2d', iterate=15))[1]
  assert max(k['depth'] for k in table) == 12
  table = kernel.generate(spec_of('seidel2d', iterate=100))[1]
  fused = {k['depth']: k for k in table if k['kind'] == 'fused'}
  assert sorted(fused) == [1, 2, 4, 8, 12, 16] and fused[16]['ring'] == 6
  table = kernel.generate(spec_of('blur', iterate=100))[1]
  fused = {k['depth']: k for k in table if k['kind'] == 'fused'}
  assert sorted(fused) == [1, 2, 4, 8, 12]       # integer program: no depth 16
  assert fused[8]['groups'] == 4 and fused[8]['ring'] == 6 and not fused[8]['pairs']
  assert not fused[4].get('groups')


def test_3d_wave_pipelined_kernel_and_its_constraints(tmp_path):
  """Depth-4 3-D kernels: the wave-pipelined one (one level per wavefront, 64x32
  tiles) and, for programs light on arithmetic, the block form next to it (all
  levels in every wavefront, 8 bands of 128x8 = 128x64 tiles, edge rows through
  LDS); both declare the smallest array they accept; programs they do not cover
  keep depth <= 2."""
  spec = spec_of('jacobi3d', iterate=8)
  text, table = kernel.generate(spec)
  blk = [k for k in table if k['kind'] == 'fused' and k['depth'] == 4 and k.get('stack')]
  assert [k['name'] for k in blk] == ['jacobi3d_fused_k4b']
  # 112 of the 120 columns a tile could keep: row segments of neighbouring tiles
  # meet on 64-byte boundaries (origin_align 16 floats); XCD runs (xcd_tiles -1)
  assert blk[0]['block'] == [512, 1, 1] and blk[0]['tile'][:2] == [112, 56]
  assert blk[0]['origin_align'] == 16
  # input planes through the two-slot LDS ring
  assert blk[0]['min_extent'] == [128, 64]
  assert (blk[0]['prefetch'], blk[0]['ring']) == (0, 2)
  # the trips at a chunk's ends skip the levels nobody needs yet (a guarded copy of the
  # row loop; the steady trips keep the branch-free body)
  assert blk[0]['lean_fill'] == 1 and "// trips at the chunk's start" in text
  assert 'if (n + 0 >= 2 && n + 0 < span + 8) {' in text
  # launches beyond the Infinity Cache store around the caches: an instantiation of
  # its own, chosen by the kernel's entry from the box it is given
  assert blk[0]['nt'] == 4 and '_band<false, true>(' in text
  # ... and in whole 64-byte pieces: the cells between the box and the next boundary
  # are unspecified by contract (include/soda_hip.h) and stored along
  assert blk[0]['wide_stores'] == 2 and 'const bool ST_WIDE = NT;' in text
  assert '* 8 > %dll' % kernel_common.NT_STREAMING_BYTES in text
  in_registers = [k for k in kernel.generate(spec, blk_prefetch=1)[1]
                  if k.get('stack') and k['depth'] == 4]
  assert (in_registers[0]['prefetch'], in_registers[0]['ring']) == (1, 0)
  assert blk[0]['xcd_tiles'] == -1 and blk[0]['fill_rows'] == 8
  # options that were measured null twice are gone: asking for one is an error
  with pytest.raises(TypeError):
    kernel.generate(spec, blk_skip_fill=1)
  with pytest.raises(TypeError):
    kernel.generate(spec, blk_asm_sched=1)
  assert 'edges[' in text and 'soda_lds_barrier' in text
  k4 = [k for k in table if k['kind'] == 'fused' and k['depth'] == 4 and k.get('groups')]
  assert k4 and k4[0]['groups'] == 4 and k4[0]['block'] == [256, 1, 1]
  assert k4[0]['xcd_tiles'] == 4 and k4[0]['buffer_io'] == 1
  assert k4[0]['min_extent'] == [64, 32] and k4[0]['tile'][:2] == [56, 24]
  assert k4[0]['lds_bytes'] <= 64 * 1024
  assert 'v_permlane32_swap' in text or 'rows_across_halves' in text
  # light on arithmetic (memory-bound): scalar form; heat3d: packed pair-rows,
  # seams as scalar pairs
  assert k4[0]['pairs'] == 0
  packed_text, packed = kernel.generate(spec, wp_pairs=1)
  assert [k['pairs'] for k in packed if k['depth'] == 4 and k.get('groups')] == [1]
  assert 'pk2_shifted{' in packed_text
  heat = kernel.generate(spec_of('heat3d', iterate=8))[1]
  assert [k.get('pairs') for k in heat if k['depth'] == 4 and k.get('groups')] == [1]
  # heavy plain-float programs: packed pair-rows in the block form too (the ring freed
  # the registers for them)
  assert [(k['name'], k['ring'], k['pairs']) for k in heat
          if k.get('stack') and k['depth'] == 4] == [('heat3d_fused_k4b', 2, 1)]
  scalar = kernel.generate(spec_of('heat3d', iterate=8), blk_pairs=0)[1]
  assert [k['pairs'] for k in scalar if k.get('stack') and k['depth'] == 4] == [0]
  # the block form serves depths 1 and 2 as well, next to the single-wave kernels
  # (which stay for arrays below its 128 x 64 tile and programs it does not take)
  assert [k['name'] for k in table if k['kind'] == 'fused' and k['depth'] <= 2] == [
      'jacobi3d_fused_k1', 'jacobi3d_fused_k2', 'jacobi3d_fused_k1b', 'jacobi3d_fused_k2b']
  shallow = {k['name']: k for k in table if k.get('stack') and k['depth'] <= 2}
  assert shallow['jacobi3d_fused_k1b']['tile'][:2] == [112, 62]
  assert shallow['jacobi3d_fused_k2b']['fill_rows'] == 4
  only_deep = kernel.generate(spec, deep3d_from=3)[1]
  assert [k['name'] for k in only_deep if k.get('stack')] == ['jacobi3d_fused_k4b']
  out = tmp_path / 'j3d.hsaco'
  kernel.compile_to_code_object(text, str(out))
  assert open(out, 'rb').read(4) == b'\x7fELF'
  # two inputs: no iteration chain, no deep kernel
  table = kernel.generate(spec_of('denoise3d'))[1]
  assert max(k['depth'] for k in table) <= 1
  # iterate below the depth: not generated
  table = kernel.generate(spec_of('jacobi3d', iterate=3))[1]
  assert max(k['depth'] for k in table) == 2


def test_3d_block_form_options_pairs_and_ring(tmp_path):
  """Experimental options of the block form (docs/DESIGN_HISTORY.md 4.1b): packed pair-rows and
  the per-wavefront LDS ring fed by LDS-direct loads.  A ring kernel waits with a
  COUNTED vmcnt, so it must issue the same number of stores every step and must
  not contain a release fence at its barrier."""
  spec = specmod.spec_from_stencil(frontend.load(
      os.path.join(SAMPLES, 'jacobi3d.soda'), iterate=4))
  text, table = kernel.generate(spec, depths=[4], deep3d='blk', blk_prefetch=0,
                                blk_ring=2, blk_pairs=1)
  blk = [k for k in table if k.get('stack')]
  assert len(blk) == 1 and blk[0]['ring'] == 2 and blk[0]['pairs'] == 1
  assert blk[0]['lds_bytes'] == 80 * 1024 + 2 * 32 * 1024
  body = text[text.index('jacobi3d_fused_k4b_band('):]
  # one plane of ring loads (4 x 2 rows) and 2 steps x 8 stores behind the awaited one
  # (the instantiation for ragged tiles stores column by column: 2 x 8 x 2 stores)
  # (the shipped ring loads are the buffer form, masked to the box's reach; the plain
  # global form with blk_mask_loads=0)
  assert '// vmcnt(36 / 20)' in body and '__builtin_amdgcn_raw_ptr_buffer_load_lds' in body
  assert 'ld_dma_byte' in body and '__builtin_amdgcn_global_load_lds' not in body
  plain = kernel.generate(spec, depths=[4], deep3d='blk', blk_mask_loads=0)[0]
  assert '__builtin_amdgcn_global_load_lds' in plain and 'ld_dma_byte' not in plain
  assert 'soda_lds_barrier();' in body and 'soda_block_barrier();' not in body
  assert 'pk2_shifted{' in body and 'pk_from_lane_below(' in body
  assert 'if (z >= z0 && z < z1) {' not in body       # stores are never skipped
  kernel.compile_to_code_object(text, str(tmp_path / 'k.hsaco'))
  with pytest.raises(kernel_stream2d.NotFusable):      # ring and register prefetch
    from soda_hip.codegen import kernel_stream3d_blk
    kernel_stream3d_blk.emit(spec, 4, prefetch=1, ring=2)
  # packing needs a plain float program: heat3d qualifies, an integer one does not
  from soda_hip.codegen import kernel_stream3d_blk
  _, entry = kernel_stream3d_blk.emit(spec_of('heat3d'), 4, stack=4, pairs=1)
  assert entry['pairs'] == 1


def test_stores_around_the_caches_for_boxes_beyond_the_infinity_cache(tmp_path):
  """The memory-bound kernels (2-D depth <= 2, 3-D single-wave, 3-D block form) carry
  a second instantiation of their interior path whose stores are non-temporal, and
  their entry takes it when the launch's box - all inputs and outputs - is larger
  than kernel_common.NT_STREAMING_BYTES.  Deeper 2-D kernels and the 3-D
  wave-pipelined form do not (measured: no gain)."""
  limit = '%dll' % kernel_common.NT_STREAMING_BYTES
  text, table = kernel.generate(spec_of('blur'))
  assert [k.get('nontemporal') for k in table if k['kind'] == 'fused'] == [4]
  assert '* 4 > ' + limit in text and '_strip<true, false, true>(' in text    # u16 in + out
  text, table = kernel.generate(spec_of('denoise2d'))
  assert '* 12 > ' + limit in text                  # two float inputs, one float output
  text, table = kernel.generate(spec_of('jacobi2d', iterate=32))
  by_depth = {k['depth']: k for k in table if k['kind'] == 'fused'}
  assert by_depth[1].get('nontemporal') == 4 and by_depth[2].get('nontemporal') == 4
  assert all('nontemporal' not in by_depth[d] and 'nt' not in by_depth[d]
             for d in by_depth if d > 2)
  text, table = kernel.generate(spec_of('denoise3d'))
  assert [k.get('nt') for k in table if k['kind'] == 'fused'] == [4]
  assert '_tile<true, true>(' in text and '* 12 > ' + limit in text
  text, table = kernel.generate(spec_of('heat3d', iterate=8))
  assert [k.get('nt') for k in table if k['depth'] == 4 and k.get('groups')] == [None]
  # off on request
  text, table = kernel.generate(spec_of('blur'), nontemporal=0)
  assert 'nontemporal' not in text and limit not in text


def test_lane_crossing_operands_leave_short_circuit_expressions():
  """`a(-1,0) < 5 || a(1,0) == 9` in a fused kernel: the x-neighbours come from other
  lanes by DPP, which must run with every lane active - not inside the right-hand
  side of `||`, where a lane whose neighbour went the other way would read 0.  The
  generators evaluate such operands first (kernel_common.cell_assignment); programs
  without && / || keep the operands inside the expression."""
  text = '''
    kernel: shortcut
    burst width: 512
    unroll factor: 1
    iterate: 2
    input int32: a(32, *)
    output int32: b(0, 0) = a(0, 0) + (a(-1, 0) < 5 || a(1, 0) == 9) + (a(0, 1) > 3 && a(-1, -1) != 7)
  '''
  spec = specmod.spec_from_stencil(frontend.loads(text))
  src, table = kernel.generate(spec)
  assert [k['depth'] for k in table if k['kind'] == 'fused'] == [1, 2]
  fused = src[src.index('shortcut_fused_k1_strip'):]
  assignments = [l for l in fused.splitlines() if ' || ' in l]
  assert assignments and all('from_lane' not in l for l in assignments)
  assert 'const auto soda_lane0 = from_lane_' in fused
  plain, _ = kernel.generate(spec_of('jacobi2d'))
  assert 'soda_lane' not in plain and 'from_lane_below(' in plain


def test_pipeline_lags_match_the_reference_reuse_model():
  """jacobi2d: each level trails the previous by one row and keeps three rows
  (the reference's reuse chain for a 3-row window is 2 rows + 1)."""
  spec = spec_of('jacobi2d', iterate=4)
  insts, final = kernel_stream2d.build_pipeline(spec, 4, prefetch=3)
  assert [i.lag for i in insts] == [0, 4, 5, 6, 7]
  assert [i.keep for i in insts] == [6, 3, 3, 3, 0]
  assert final.final and final.lag == 7
  # blur: blur_x reads 3 input rows, blur_y only the current blur_x row
  insts, final = kernel_stream2d.build_pipeline(spec_of('blur'), 1, prefetch=0)
  assert [(i.ident, i.lag, i.keep) for i in insts] == \
      [('in_input', 0, 3), ('k0_blur_x', 2, 1), ('k0_blur_y', 2, 0)]


def test_unfusable_programs_fall_back_to_stage_kernels():
  # denoise3d (lowered to two stages over two inputs) does not fit the register
  # file with two columns per lane; it fuses with one column and 12 rows
  text, table = kernel.generate(spec_of('denoise3d'))
  fused = [k for k in table if k['kind'] == 'fused']
  assert [(k['depth'], k['rows'], k['cols'], k['tile'][1]) for k in fused] == [(1, 12, 1, 8)]
  assert len([k for k in table if k['kind'] == 'stage']) == 2
  # the source program (eight stages, not lowered) fits no fused form at all
  text, table = kernel.generate(spec_of('denoise3d'), inline=False)
  assert all(k['kind'] == 'stage' for k in table) and 'not fused' in text
  with pytest.raises(kernel_stream2d.NotFusable):
    kernel_stream2d.emit(spec_of('jacobi3d'), 1)
  # 3-D single-output programs get the plane-streaming kernels
  _, table = kernel.generate(spec_of('jacobi3d'))
  fused = [k for k in table if k['kind'] == 'fused']
  assert [k['depth'] for k in fused] == [1, 2, 1, 2]      # single-wave, then block form
  assert [bool(k.get('stack')) for k in fused] == [False, False, True, True]
  assert fused[1]['tile'][1] == 12 and fused[1]['fill_rows'] == 4
  # multi-input 2-D programs fuse at depth 1 with padded window lengths
  _, table = kernel.generate(spec_of('denoise2d'))
  assert [k['depth'] for k in table if k['kind'] == 'fused'] == [1]


def test_four_dimensional_programs_get_stage_kernels(tmp_path):
  """Four dimensions is what `buffer_t` and `<app>_test(blob, dims[4])` allow
  (reference header.py:36-48, host.py:992): per-stage kernels only, with the
  fourth index and its stride in the cell address."""
  import os
  from conftest import SAMPLES
  spec = specmod.spec_from_stencil(
      frontend.load(os.path.join(SAMPLES, 'extra', 'hyper4d.soda')))
  assert spec['dim'] == 4
  text, table = kernel.generate(spec)
  assert [k['kind'] for k in table] == ['stage']
  assert 'w * s3' in text and '(-1) * s3' in text and '(1) * s3' in text
  kernel.compile_to_code_object(text, str(tmp_path / 'hyper4d.hsaco'),
                                extra_flags=['-Werror=array-bounds'])


def test_pointwise_locals_are_folded_into_their_readers():
  """denoise: diff_*, r0, r1 are read only at offset 0 -> two stages remain; the
  cast reproduces the rounding of the removed store; windows are unchanged."""
  for app, kept in (('denoise2d', ['g', 'output']), ('denoise3d', ['g', 'output']),
                    ('sobel2d', ['mag']), ('blur', ['blur_x', 'blur_y']),
                    ('jacobi2d', ['t0'])):
    spec = spec_of(app)
    low = specmod.inline_pointwise(spec)
    assert [s['name'] for s in low['stages']] == kept
    assert specmod.iteration_margins(low, 1) == specmod.iteration_margins(spec, 1)
    assert low['inputs'] == spec['inputs'] and low['outputs'] == spec['outputs']
  g = specmod.inline_pointwise(spec_of('denoise2d'))['stages'][0]
  assert 'static_cast<float >(({u:0,0} - {u:0,-1}))' in g['expr']
  assert sorted(map(tuple, (r for _, r in g['loads']))) == \
      [(-1, 0), (0, -1), (0, 0), (0, 1), (1, 0)]
  # the blob is still identified by the SOURCE program
  text, _ = kernel.generate(spec_of('denoise2d'))
  meta = kernel_common.read_meta_from_source(text)
  assert meta['program_hash'] == kernel_common.program_hash(spec_of('denoise2d'))
  assert [s['name'] for s in meta['spec']['stages']] == ['g', 'output']


def test_fill_prologue_start_steps():
  """During the pipeline fill level t of jacobi2d first matters at step
  prefetch + 2t (it trails by t rows and is needed t rows further down)."""
  spec = spec_of('jacobi2d', iterate=4)
  text, entry = kernel_stream2d.emit(spec, 4, prefetch=3)
  assert entry['fill_rows'] == 11
  for t in (1, 2, 3, 4):
    assert 'if (n + 0 >= %d) {' % (3 + 2 * t) in text


def test_dpp_combine_only_for_pure_float32_programs():
  off = ['-mllvm', '-amdgpu-dpp-combine=false']
  # 2-D programs the packed kernels cover also get the ILP-first scheduler
  ilp = ['-mllvm', '-amdgpu-sched-strategy=max-ilp']
  for app in ('jacobi2d', 'seidel2d'):
    assert kernel.extra_flags(spec_of(app)) == ilp
  for app in ('jacobi3d', 'heat3d'):
    assert kernel.extra_flags(spec_of(app)) == []
  for app in ('blur', 'sobel2d', 'denoise2d', 'denoise3d'):   # ints / double math
    assert kernel.extra_flags(spec_of(app)) == off
  head = ('kernel: k\nburst width: 512\nunroll factor: 1\niterate: 2\n'
          'input %s: a(8, *)\noutput %s: o(0, 0) = ')

  def flags(ty, expr):
    return kernel.extra_flags(specmod.spec_from_stencil(
        frontend.loads(head % (ty, ty) + expr + '\n')))
  assert flags('int32', 'a(0, 0) - a(2, 0) + a(-2, 0)') == off
  assert flags('float', 'a(0, 0) * 0.3 + a(1, 0) * 2.f') == off        # double literal
  assert flags('float', 'a(0, 0) * 1e3 + a(1, 0)') == off
  assert flags('float', 'a(0, 0) * .5f + a(1, 0) * 2.f + a(0, 1) * 3') == ilp
  assert flags('double', 'a(0, 0) * 0.5 + a(1, 0)') == off
  text, _ = kernel.generate(spec_of('sobel2d'))
  assert kernel.flags_from_text(text) == off


def test_program_desc_for_the_c_abi():
  desc = host.program_desc(spec_of('sobel2d'))
  assert (desc.dim, desc.n_inputs, desc.n_stages, desc.n_outputs) == (2, 1, 3, 1)
  assert list(desc.elem_size[:4]) == [2, 2, 2, 2]
  assert desc.output_tensor[0] == 3
  wins = {(w.stage, w.parent): (list(w.lo[:2]), list(w.hi[:2]))
          for w in desc.window[:desc.n_windows]}
  assert wins[(1, 0)] == ([-1, -1], [1, 1])
  assert wins[(3, 1)] == ([0, 0], [0, 0])


def test_valid_cell_counts_match_survey():
  """SURVEY.md section 8d work counts."""
  spec = spec_of('jacobi2d')
  def close(a, b):
    return abs(a - b) <= 5e-4 * b
  assert specmod.valid_cells(spec, [8192, 8192], 100) == \
      sum((8192 - 2 * k) ** 2 for k in range(1, 101))
  assert close(specmod.valid_cells(spec, [8192, 8192], 100), 6.547e9)
  assert specmod.valid_cells(spec, [16384, 16384], 1000) == 236970022000
  assert close(specmod.valid_cells(spec, [16384, 16384], 1000), 2.3697e11)
  assert close(specmod.valid_cells(spec_of('jacobi3d'), [512, 512, 512], 200),
               8.504e9)
  assert specmod.valid_cells(spec_of('blur'), [16384, 16384], 1) == 268369924
  assert specmod.algorithmic_bytes_per_update(spec) == 8
  assert specmod.algorithmic_bytes_per_update(spec_of('blur')) == 4


@pytest.mark.parametrize('app', APPS)
def test_generated_cpp_host_is_valid_cpp(app, tmp_path):
  src = tmp_path / (app + '_host.cpp')
  r = run_sodac(os.path.join(SAMPLES, app + '.soda'), '--hip-host-cpp', str(src))
  assert r.returncode == 0, r.stderr
  subprocess.check_call(['g++', '-std=c++11', '-fopenmp', '-fsyntax-only', '-Wall',
                         '-Werror', '-I', os.path.join(ROOT, 'include'), str(src)])
  text = src.read_text()
  assert 'extern "C" int %s_test(const char* blob, const int dims[4])' % app in text


@pytest.mark.parametrize('app', APPS)
def test_compilation_sweep_like_the_reference_script(app, tmp_path):
  """reference tests/test-compilation.sh:5-10 pipes every sample through
  `sodac --xocl-kernel -` and a syntax-only compile, forwarding extra flags such
  as --unroll-factor; the HIP analogue, through the real CLI and hipcc."""
  iterates = ['1', '3'] if app not in ('denoise2d', 'denoise3d') else ['1']
  for unroll, iterate in zip(('1', '4'), iterates + iterates):
    r = run_sodac(os.path.join(SAMPLES, app + '.soda'), '--unroll-factor', unroll,
                  '--iterate', iterate, '--hip-max-depth', '2', '--hip-kernel', '-')
    assert r.returncode == 0, r.stderr
    src = tmp_path / ('%s_u%s_i%s.hip' % (app, unroll, iterate))
    src.write_text(r.stdout)
    flags = [f for f in kernel.HIPCC_FLAGS if f != '--no-gpu-bundle-output']
    subprocess.check_call(['/opt/rocm/bin/hipcc'] + flags +
                          kernel.flags_from_text(r.stdout) +
                          ['-fsyntax-only', str(src)])


@pytest.mark.parametrize('app,shape,iterate', [
    ('jacobi2d', (40, 50), 3), ('blur', (33, 47), 1), ('denoise2d', (30, 40), 1),
    ('heat3d', (12, 14, 16), 2), ('sobel2d', (30, 44), 2)])
def test_selfcheck_of_app_test_counts_mismatches(app, shape, iterate):
  """The CPU half of `<app>_test` (reference host.py:1073-1146): zero mismatches
  on a correct result, exactly the corrupted cells otherwise; corruption outside
  the compared region is ignored, as in the reference."""
  import numpy as np
  from soda_hip.runtime import selfcheck
  from oracle import soda_oracle
  spec = spec_of(app)
  orc = soda_oracle.Oracle(spec)
  rng = np.random.default_rng(1)
  ins = []
  for t in spec['inputs']:
    dt = np.dtype(specmod.NUMPY_NAME[t['c_type']])
    ins.append(rng.random(shape, dtype=np.float32).astype(dt) if dt.kind == 'f'
               else rng.integers(0, 256, size=shape).astype(dt))
  want = orc.run(ins, iterate=iterate)
  outs = [want[o] for o in spec['outputs']]
  assert selfcheck.count_mismatches(spec, ins, outs, iterate) == 0
  sl = orc.valid_slices(tuple(reversed(shape)), iterate)
  bad = [o.copy() for o in outs]
  inside = tuple(s.start + 1 for s in sl)
  bad[0][inside] += 1
  bad[0][tuple(n - 1 for n in shape)] += 1    # outside the compared region
  assert selfcheck.count_mismatches(spec, ins, bad, iterate, max_report=0) == 1


def test_parser_integer_forms():
  from soda_hip.frontend.parser import _Parser
  for text, value in (('0x1F', 31), ('-0b101', -5), ('017', 15), ('42u', 42),
                      ('+7', 7)):
    assert _Parser(text).c_int() == value
  with pytest.raises(Exception):
    _Parser('1.5').c_int()


@pytest.mark.parametrize('body', [
    'o(0, 0) = a(1, 0) + a(2, 0) + a(0, 1)',          # dx range 1..2 excludes 0
    'o(0, 0) = a(-2, 0) + a(-1, 0)',                   # dx range -2..-1
    'o(0, 0) = a(0, 0) + a(3, 1) * 2.0f - a(-1, -1)',
])
def test_one_sided_windows_keep_register_rows_in_bounds(body, tmp_path):
  """Per-stage kernels keep a register row per (tensor, outer offset) whose range
  always covers dx = 0 (the V-wide centre load lands there): a window that lies
  on one side of the store point must not index outside that row.  Compiled with
  -Werror=array-bounds so that a regression is a build failure, not UB."""
  text = ('kernel: onesided\nburst width: 512\nunroll factor: 2\niterate: 1\n'
          'input float: a(32, *)\noutput float: %s\n' % body)
  spec = specmod.spec_from_stencil(frontend.loads(text))
  src, table = kernel.generate(spec)
  assert 'r0[-' not in src
  out = tmp_path / 'onesided.hsaco'
  kernel.compile_to_code_object(src, str(out), extra_flags=['-Werror=array-bounds'])
  assert out.stat().st_size > 1000


def test_generated_cpp_host_multi_gpu_entry_compiles(tmp_path):
  """The generated C++ host with -DSODA_HIP_MULTI_GPU (`<app>_multi_gpu`: one
  thread per GPU, ncclCommInitAll, soda_hip_run_slab) is valid C++ against
  include/soda_hip.h and the RCCL header."""
  src = tmp_path / 'jacobi3d_host.cpp'
  r = run_sodac(os.path.join(SAMPLES, 'jacobi3d.soda'), '--hip-host-cpp', str(src))
  assert r.returncode == 0, r.stderr
  text = src.read_text()
  assert 'extern "C" int jacobi3d_multi_gpu(' in text and 'soda_hip_run_slab(' in text
  subprocess.check_call(['g++', '-std=c++17', '-fsyntax-only', '-fopenmp',
                         '-DSODA_HIP_MAIN', '-DSODA_HIP_MULTI_GPU',
                         '-D__HIP_PLATFORM_AMD__', '-I', '/opt/rocm/include', '-I',
                         os.path.join(ROOT, 'include'), str(src)])
  # programs with several inputs have no slab entry (yet)
  src2 = tmp_path / 'denoise2d_host.cpp'
  assert run_sodac(os.path.join(SAMPLES, 'denoise2d.soda'), '--hip-host-cpp',
                   str(src2)).returncode == 0
  assert '_multi_gpu' not in src2.read_text()


ROUND4_FUZZ_FINDS = {
    # edge rows of an int32 stage through the LDS array of a uint32 input (the block form
    # read them back with a narrowing initialiser: a compile error)
    'ops7222': """kernel: ops7222
burst width: 512
unroll factor: 2
iterate: 2
input uint32: a(32, 32, *)
local int32: m(0, 0, 0) = a(0, 0, 0) - (a(0, 1, 1) ^ a(0, -1, -1)) + (a(0, 0, 1) >= a(0, 1, 1)) + (-a(-1, 1, -1))
output uint32: out(0, 0, 0) = m(0, 0, 0) - (-m(-1, -1, 1)) + (m(1, 0, -1) % 3)
""",
    # two stages that read the INPUT's edge rows ahead of their arithmetic (block form at
    # depth 1, register prefetch): the rows were declared twice in the step's scope
    'st7327': """kernel: st7327
burst width: 512
unroll factor: 4
iterate: 2
input float: in0(64, 64, *)
local float: loc0(0, 0, 0) = (in0(0, 0, 0) + in0(1, 1, 2) * 0.25f - in0(1, 0, -1) + in0(0, 0, -1) - in0(-1, 0, -1)) * 0.125f
local float: loc1(0, 0, 0) = loc0(0, 0, 0) + loc0(2, 1, 1) + in0(-1, -1, 2)
output float: out(0, 0, 0) = in0(0, 0, 0) - loc1(1, 0, -2) * 0.125f - in0(-1, -2, 2) * 0.5f
"""}


@pytest.mark.parametrize('name', sorted(ROUND4_FUZZ_FINDS))
def test_programs_the_round4_fuzzing_found(name, tmp_path):
  """Two 3-D programs of tools/fuzz_gpu.py (fresh seeds, round 4) whose block-form
  kernels did not compile; they build now (and run against the oracle in the GPU suite)."""
  spec = specmod.spec_from_stencil(frontend.loads(ROUND4_FUZZ_FINDS[name]))
  text, table = kernel.generate(spec)
  assert any(k.get('stack') for k in table), [k['name'] for k in table]
  kernel.compile_to_code_object(text, str(tmp_path / 'k.hsaco'))


def test_edge_tiles_cover_every_box_exactly_once():
  """soda_hip_kernel.edge_slack (3-D block form, round 6): the launcher's rule for where
  the tiles of a row start and how many there are, and the kernel's rule for what the first
  and the last of them store, restated here and checked against each other for every box
  start and width: the stored ranges tile [box_lo, box_hi) exactly, every stored column is
  one the tile's (possibly moved) window computes validly, and no box takes more tiles
  than under the old rule (start at box_lo rounded down, every tile stores tile[0])."""
  def launcher(lo, hi, tile, slack, align=16):          # soda_hip.cpp: make_launch
    x0 = (lo + slack) - (lo + slack) % align
    if x0 >= hi:
      x0 = lo - lo % align
    nx = max(1, -(-(hi - x0 - slack) // tile))
    if nx == 1 and hi > x0 + tile:
      x0 = lo - lo % align
      nx = max(1, -(-(hi - x0 - slack) // tile))
    return x0, nx

  def kernel_ranges(lo, hi, dims0, x0, nx, tile, slack, halo, width=128):   # the entry
    out = []
    for bx in range(nx):
      xs = x0 + bx * tile
      if xs >= hi:
        continue
      shifted = bx == 0 and lo < xs
      lo_ext = slack if shifted else 0
      hi_ext = slack if bx + 1 == nx and not shifted else 0
      wx = min(xs - halo - lo_ext, dims0 - width)
      wx = max(wx, 0)
      st = (max(xs - lo_ext, lo), min(xs + tile + hi_ext, hi))
      valid = (wx + halo if wx > 0 else 0, wx + width - halo if wx + width < dims0 else dims0)
      assert valid[0] <= st[0] and st[1] <= valid[1], (lo, hi, dims0, bx, st, valid)
      out.append(st)
    return out
  saved = 0
  for tile, slack, halo in ((112, 8, 4), (112, 12, 2)):      # depth 4; depths 1 and 2
    for lo in range(halo, 70):
      for n in list(range(1, 260)) + [328, 336, 344, 448, 456, 504]:
        hi = lo + n
        for dims0 in (hi + halo, hi + 77):
          if dims0 < 128:
            continue
          x0, nx = launcher(lo, hi, tile, slack)
          at = lo
          for a, b in kernel_ranges(lo, hi, dims0, x0, nx, tile, slack, halo):
            assert a == at and b > a
            at = b
          assert at == hi, (lo, hi, x0, nx)
          old = -(-(n + lo % 16) // tile)
          assert nx <= old
          saved += old - nx
  assert saved > 0
  # cfg5: the boxes whose tile count per row drops
  drops = [n for n in range(112, 505, 8)
           if launcher(256 - n // 2, 256 + n // 2, 112, 8)[1] <
           -(-(n + (256 - n // 2) % 16) // 112)]
  assert drops == [112, 232, 240, 328, 336, 456, 464]
  # ... and the shipped jacobi3d kernels carry the figure the launcher reads
  spec = spec_of('jacobi3d', iterate=200)
  slack = {k['name']: k.get('edge_slack', 0) for k in kernel.generate(spec)[1]}
  assert slack['jacobi3d_fused_k4b'] == 8 and slack['jacobi3d_fused_k1b'] == 12
  assert slack['jacobi3d_fused_k4'] == 0
