"""The reference's tiled / burst-aligned DRAM layout (soda_hip.runtime.layout)
against fixtures produced by the reference's OWN emitted tiling and copy-back
loops (tests/golden/make_layout_golden.py): overlapping tiles, short last tiles,
several DRAM banks, 2-D and 3-D, the output stream delay."""
import json
import os

import numpy as np
import pytest

from soda_hip import frontend
from soda_hip.runtime import layout

from conftest import ROOT, SAMPLES

GOLDEN = os.path.join(ROOT, 'tests', 'golden')
with open(os.path.join(GOLDEN, 'layout_manifest.json')) as f:
  MANIFEST = json.load(f)
BITS = {'uint16': 16, 'float32': 32}


def test_stencil_constants_match_the_reference_macros():
  """STENCIL_DIM_d and STENCIL_DISTANCE of the generated host (host.py:1183-1197)
  for every sample and iterate 1..4, at the samples' own tile sizes."""
  with open(os.path.join(GOLDEN, 'analysis.json')) as f:
    analysis = json.load(f)
  for key, ref in analysis.items():
    app, it = key.split('.iter')
    st = frontend.load(os.path.join(SAMPLES, app + '.soda'), iterate=int(it))
    c = layout.stencil_constants(st)
    for d in range(st.dim):
      assert c['stencil_dim'][d] == ref['macros']['STENCIL_DIM_%d' % d], key
    assert c['stencil_distance'] == ref['macros']['STENCIL_DISTANCE'], key


@pytest.mark.parametrize('fixture', sorted(MANIFEST))
def test_pack_and_unpack_match_the_reference_loops(fixture):
  meta = MANIFEST[fixture]
  data = np.load(os.path.join(GOLDEN, fixture))
  st = frontend.load(os.path.join(SAMPLES, meta['app'] + '.soda'),
                     iterate=meta['iterate'])
  tile = meta['tile_size']
  c = layout.stencil_constants(st, tile)
  assert c['stencil_distance'] == meta['stencil_distance']
  for name in st.input_names:
    a = data['in_' + name]
    lay = layout.TiledLayout(meta['dims'], tile, c['stencil_dim'],
                             meta['burst_width'], BITS[a.dtype.name],
                             meta['banks_in'])
    banks = lay.pack(a, c['stencil_distance'])
    for b in meta['banks_in']:
      want = data['inbuf_%s_%d' % (name, b)]
      assert banks[b].shape == want.shape, (name, b)
      assert np.array_equal(banks[b], want), (name, b)
    assert np.array_equal(lay.import_input(banks), a)
    assert np.array_equal(lay.import_input(
        {b: data['inbuf_%s_%d' % (name, b)] for b in meta['banks_in']}), a)
  for name in st.output_names:
    want = data['out_' + name]
    lay = layout.TiledLayout(meta['dims'], tile, c['stencil_dim'],
                             meta['burst_width'], BITS[want.dtype.name],
                             meta['banks_out'], input_banks=meta['banks_in'])
    bufs = {b: data['outbuf_%s_%d' % (name, b)] for b in meta['banks_out']}
    got = lay.unpack(bufs, c['copy_back_offset'], c['stencil_offset'][name],
                     window_dim=c['copy_back_dim'])
    assert np.array_equal(got, want), name
    assert want.any()
    if lay.tiles_overlap:     # fewer output banks than input banks: see layout.py
      with pytest.raises(ValueError):
        lay.export_output(want, c['copy_back_offset'], c['stencil_offset'][name],
                          c['stencil_distance'])
      continue
    # and back: the exported buffers hold every copied-back cell where the
    # reference's buffers hold it
    again = lay.export_output(want, c['copy_back_offset'], c['stencil_offset'][name],
                              c['stencil_distance'], window_dim=c['copy_back_dim'])
    assert np.array_equal(lay.unpack(again, c['copy_back_offset'],
                                     c['stencil_offset'][name],
                                     window_dim=c['copy_back_dim']), want)
    if len(st.input_names) == 1:   # (denoise2d's copy-back regions overlap, see
      for b in meta['banks_out']:  # stencil_constants: the last tile wins)
        written = again[b] != 0
        assert np.array_equal(again[b][written], bufs[b][written])


def test_round_trip_through_the_layout():
  """pack -> (identity kernel: output stream = input stream delayed) -> unpack
  returns the interior of every tile."""
  st = frontend.load(os.path.join(SAMPLES, 'jacobi2d.soda'), iterate=2)
  dims, tile = (70, 15), [24]
  c = layout.stencil_constants(st, tile)
  lay = layout.TiledLayout(dims, tile, c['stencil_dim'], 512, 32, (0, 1))
  a = np.random.default_rng(5).random((15, 70), dtype=np.float32)
  banks = lay.pack(a, c['stencil_distance'])
  got = lay.unpack(banks, c['window_offset'], 0)
  lo, ext = c['window_offset'], c['stencil_dim']
  inner = (slice(lo[1], 15 - (ext[1] - 1 - lo[1])), slice(lo[0], 70 - (ext[0] - 1 - lo[0])))
  assert np.array_equal(got[inner], a[inner])
  with pytest.raises(ValueError):
    layout.TiledLayout(dims, [4], c['stencil_dim'], 512, 32)
