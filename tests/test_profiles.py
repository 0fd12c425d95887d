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
