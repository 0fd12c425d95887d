"""bench.py's synthetic input: a rank of an N-GPU run materialises only ITS rows of the
seeded grid (the generator is moved past the rows before them), and gets exactly the
values the whole grid holds there."""
import sys

import numpy as np
import pytest

from conftest import ROOT

sys.path.insert(0, ROOT)
import bench                                        # noqa: E402


def spec_of(c_type):
  return dict(inputs=[dict(name='a', c_type=c_type)])


@pytest.mark.parametrize('c_type', ['float', 'uint16_t', 'int32_t', 'uint8_t', 'double',
                                    'int64_t'])
@pytest.mark.parametrize('dims', [(29, 37), (16, 20), (5, 3, 11)])
def test_rows_of_the_seeded_grid_without_the_grid(c_type, dims):
  whole = bench.make_input(spec_of(c_type), list(dims))[0]
  assert whole.shape == tuple(reversed(dims))
  n = dims[-1]
  for first, last in ((0, n), (0, 1), (1, 2), (3, n - 2), (n - 1, n), (n // 2, n // 2 + 3)):
    part = bench.make_input(spec_of(c_type), list(dims), rows=(first, last))[0]
    assert part.dtype == whole.dtype and part.flags['C_CONTIGUOUS']
    assert np.array_equal(part, whole[first:last]), (c_type, dims, first, last)


def test_slabs_of_all_ranks_tile_the_grid():
  from soda_hip.runtime import dist
  dims = [24, 50]
  whole = bench.make_input(spec_of('float'), dims)[0]
  for world in (2, 3, 8):
    parts = [bench.make_input(spec_of('float'), dims, rows=b)[0]
             for b in dist.slab_bounds(dims[-1], world)]
    assert np.array_equal(np.concatenate(parts), whole)
