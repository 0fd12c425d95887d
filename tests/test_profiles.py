"""bench.py quotes committed rocprofv3 figures (profiles/rNN_traffic.json,
rNN_sq_counters.json) in its roofline block only for the very kernel they were
measured on: these tests hold the matching rule and that the headline kernel's
figures are those of the kernel the generator ships today."""
import glob
import json
import os
import sys

from conftest import ROOT

sys.path.insert(0, ROOT)
import bench                                        # noqa: E402
import __graft_entry__ as entry                     # noqa: E402
from soda_hip import frontend                       # noqa: E402
from soda_hip.codegen import kernel                 # noqa: E402
from soda_hip.codegen import spec as specmod        # noqa: E402


def test_profile_entries_match_only_their_own_kernel_and_schedule():
  e = dict(kernel='a_fused_k4b', dims=[512, 512, 512], iterate=200, kernel_digest='p/k/d',
           launches=42)
  ok = lambda **kw: bench.profile_entry_matches(                      # noqa: E731
      e, **dict(dict(kernel='a_fused_k4b', dims=(512, 512, 512), iterate=200,
                     digest='p/k/d', launches=42), **kw))
  assert ok() and ok(launches=43) and ok(launches=41)
  assert not ok(launches=30)            # another share of the sweep's launches
  assert bench.profile_entry_matches(dict(e, uniform=True), 'a_fused_k4b', (512, 512, 512),
                                     200, 'p/k/d', 30)    # ... of equal-sized launches
  assert not ok(digest='p/k/other')     # the generator changed the kernel's shape
  assert not ok(digest=None)
  assert not ok(dims=(512, 512, 256)) and not ok(iterate=100) and not ok(kernel='a_fused_k4')
  legacy = dict(e)
  del legacy['kernel_digest']           # files of earlier rounds carry no digest
  assert not bench.profile_entry_matches(legacy, 'a_fused_k4b', (512, 512, 512), 200,
                                         'p/k/d', 42)


def test_headline_kernel_has_fresh_traffic_and_counters():
  """The newest traffic and SQ-counter files hold the depth-24 kernel of jacobi2d on the
  BASELINE grid, measured on the kernel shape the generator emits now."""
  st = frontend.load(entry.sample_path('jacobi2d'), iterate=entry.BLOB_ITERATE.get('jacobi2d'))
  spec = specmod.spec_from_stencil(st)
  digests = {e['name']: kernel.calibration_key(e, spec) for e in kernel.generate(spec)[1]}
  for pattern in ('r*_traffic.json', 'r*_sq_counters.json'):
    newest = sorted(glob.glob(os.path.join(ROOT, 'profiles', pattern)))[-1]
    with open(newest) as f:
      entries = json.load(f)['entries']
    mine = [x for x in entries if x['kernel'] == 'jacobi2d_fused_k24' and
            list(x['dims']) == [16384, 16384] and x['iterate'] == 1000]
    assert mine, newest
    assert mine[0]['kernel_digest'] == digests['jacobi2d_fused_k24'], newest


def test_split_argument_reads_back_a_depth_schedule():
  """bench.py --split takes what config.depth_schedule prints (the counter passes of
  tools/collect_on_gpu.sh repeat a plain run's schedule with it)."""
  import pytest
  assert bench.parse_split('41x24+1x16', 1000) == [24] * 41 + [16]
  assert bench.parse_split('5x20', 100) == [20] * 5
  assert bench.parse_split('1x2 + 1x1', 3) == [2, 1]
  for bad, iterate in (('41x24', 1000), ('24', 24), ('ax4', 4), ('0x4+1x4', 4), ('1x4+', 4)):
    with pytest.raises(SystemExit):
      bench.parse_split(bad, iterate)


def _round_of(path):
  import re
  return int(re.match(r'r(\d+)_', os.path.basename(path)).group(1))


def test_profiles_are_stamped_with_a_clean_commit():
  """From round 4 on every entry of the traffic and SQ-counter files names the commit the
  GPU box ran (tools/collect.sh refuses a dirty tree and stamps the snapshot): no
  '+uncommitted', no missing stamp."""
  checked = 0
  for pattern in ('r*_traffic.json', 'r*_sq_counters.json'):
    for path in glob.glob(os.path.join(ROOT, 'profiles', pattern)):
      if _round_of(path) < 4:
        continue
      with open(path) as f:
        for e in json.load(f)['entries']:
          assert e.get('commit') and not e['commit'].endswith('+uncommitted') and \
              e['commit'] != 'unknown', (path, e.get('kernel'), e.get('commit'))
          checked += 1
  newest = max(_round_of(p) for p in glob.glob(os.path.join(ROOT, 'profiles', 'r*_traffic.json')))
  assert newest < 4 or checked > 0


def test_stats_pass_reproduces_the_bench_line():
  """From round 4 on the --kernel-trace --stats pass runs the schedule of the plain run
  (bench.py --split): every kernel is called sweeps x launches-per-sweep times, and the
  summary's average durations add up to the line's own ms_per_step within 2 %."""
  for path in glob.glob(os.path.join(ROOT, 'profiles', 'r*_bench_under_rocprof.json')):
    if _round_of(path) < 4:
      continue
    with open(path) as f:
      record = json.load(f)
    check = record['check']
    assert record['commit'] and not record['commit'].endswith('+uncommitted')
    stats = {k['name']: k for k in record['kernel_stats']}
    for name, calls in check['calls_expected'].items():
      assert stats[name]['calls'] == calls, (name, stats[name]['calls'], calls)
    assert abs(check['ratio'] - 1) < 0.02, check


def test_other_configs_block_has_the_shape_the_driver_line_promises():
  """bench.py at N = 1 prints BASELINE configs 2, 3 and 5 under config.other_configs
  (condensed lines: time, rate, schedule, the dominant kernel's roofline figures); only
  the headline workload carries them."""
  import argparse
  assert [(t, a, d, i) for t, a, d, i in bench.OTHER_CONFIGS] == [
      ('cfg2', 'jacobi2d', [8192, 8192], 100), ('cfg3', 'blur', [16384, 16384], 1),
      ('cfg5', 'jacobi3d', [512, 512, 512], 200)]
  line = dict(ms_per_step=0.95, value=6900.0, steps=30, warmup=10,
              config=dict(workload='jacobi2d.soda float32 8192x8192, iterate 100',
                          launches_per_step=5, depth_schedule='5x20', effective_GBps=55000.0),
              roofline=dict(kernel='jacobi2d_fused_k20', bound='valu', frac=0.46, unit='Tlane-op/s',
                            frac_algorithmic=7.2, traffic=None, hbm_measured_frac=None,
                            hbm_floor_frac=0.36, valu_frac=0.46, kernel_avg_us=188.0,
                            kernel_launches=5, extra='dropped'))
  e = bench.other_config_entry('cfg2', line)
  assert e['config'] == 'cfg2' and e['ms'] == 0.95 and e['gcell_updates_per_s'] == 6900.0
  assert e['depth_schedule'] == '5x20' and e['launches'] == 5
  assert set(e['roofline']) == set(bench.OTHER_ROOFLINE_KEYS)
  assert e['roofline']['bound'] == 'valu' and e['roofline']['shader_clock_ghz'] is None
  ns = argparse.Namespace(app='jacobi2d', size=[16384, 16384], iterate=1000)
  assert bench.is_headline(ns)
  assert not bench.is_headline(argparse.Namespace(app='jacobi2d', size=[8192, 8192], iterate=100))
  assert not bench.is_headline(argparse.Namespace(app='blur', size=[16384, 16384], iterate=1000))


def test_the_keys_the_verdict_reads_come_first():
  """The driver's record of a bench line keeps the first 24 scalar keys of an object, drops
  lists and cuts strings at 120 characters: `roofline` starts with the
  keys the roofline argument rests on, `config` carries scalar copies of the other
  configs' figures, and `cpu_baseline.sample` fits."""
  from soda_hip import frontend
  from soda_hip.codegen import spec as specmod
  st = frontend.load(os.path.join(ROOT, 'tests', 'samples', 'jacobi2d.soda'), iterate=48)
  spec = specmod.spec_from_stencil(st)
  entry = dict(name='jacobi2d_fused_k24', kind='fused', depth=24, pairs=2, block=[256, 1, 1],
               tile=[464, 1, 256, 1], cols=4)
  timing = dict(dominant_name='jacobi2d_fused_k24', dominant_us=1400.0, dominant_launches=2,
                fastest_us=1400.0, launches=2)
  updates = [16000 * 16000] * 48
  block = bench.roofline_block(spec, None, [(entry, 700.0)] * 2, updates, timing,
                               [16384, 16384], 48, step_us=1500.0,
                               clock=dict(ghz=2.0, seconds=0.03))
  head = list(block)[:12]
  assert head == ['kernel', 'bound', 'frac', 'unit', 'achieved', 'peak', 'frac_algorithmic',
                  'hbm_measured_frac', 'valu_frac', 'traffic', 'kernel_avg_us',
                  'kernel_launches']
  scalars = [k for k, v in block.items() if not isinstance(v, (list, dict))]
  assert {'frac', 'bound', 'peak', 'unit', 'achieved', 'shader_clock_ghz'} <= set(scalars[:24])
  assert block['frac'] == block['valu_frac' if block['bound'] == 'valu' else 'hbm_floor_frac']
  config = dict(workload='w', app='jacobi2d', dims=[1, 2], iterate=1000,
                valid_cell_updates=1, nominal_cell_updates=1, nominal_gcell_updates_per_s=1.0,
                launches_per_step=42, depth_schedule='41x24+1x16', depth_split='m',
                stream_chunk_choice='calibrated', effective_GBps=1.0, device='gfx950')
  others = [dict(config='cfg2', ms=0.93, roofline=dict(bound='valu', frac=0.45, traffic=None)),
            dict(config='cfg3', ms=0.19, roofline=dict(bound='hbm', frac=0.79, traffic=1.1e9)),
            dict(config='cfg5', error='RuntimeError: x')]
  bench.add_other_configs(config, others)
  scalars = [k for k, v in config.items() if not isinstance(v, (list, dict))]
  assert len(scalars) <= 24
  assert config['cfg2_ms'] == 0.93 and config['cfg3_bound'] == 'hbm'
  assert config['cfg3_traffic'] == 1.1e9 and config['cfg5_ms'] is None
  assert list(config)[-1] == 'other_configs'


def test_cpu_baseline_is_pinned_and_its_description_fits_the_record():
  """bench.py's cpu_baseline leg on a tiny grid: threads pinned one per core, spread over the mask
  (oracle/pin_threads.c), one discarded sample + three, the description within the 120
  characters the driver's record keeps, the caller's affinity restored."""
  from soda_hip import frontend
  from soda_hip.codegen import spec as specmod
  st = frontend.load(os.path.join(ROOT, 'tests', 'samples', 'jacobi2d.soda'), iterate=4)
  spec = specmod.spec_from_stencil(st)
  before = os.sched_getaffinity(0)
  cpu = bench.cpu_baseline(spec, [96, 64], 0.2)
  assert os.sched_getaffinity(0) == before
  assert cpu['kind'] == 'port' and cpu['cores'] >= 1 and cpu['value'] > 0
  assert len(cpu['samples']) == 3 and cpu['best'] == max(cpu['samples'])
  assert cpu['value'] == sorted(cpu['samples'])[1]
  assert len(cpu['sample']) <= 120 and 'discarded' in cpu['sample']
  assert cpu['pinned'] is True
