"""The reference's DRAM layout of inputs and outputs, for callers that hold data
in it (dumps of the FPGA flow's device buffers): import into / export from the
plain row-major arrays the HIP back end works on.

The reference host relays every array before and after the kernel
(reference src/soda/codegen/xilinx/host.py:629-686 in, :823-901 out):
  * all dimensions but the last are cut into OVERLAPPING tiles of
    `TILE_SIZE_DIM_d` cells that advance by `TILE_SIZE_DIM_d - STENCIL_DIM_d + 1`
    (host.py:259-262; the last tile is as short as the array leaves it);
  * a tile is linearised with the FULL tile size as pitch, dimension 0 fastest,
    the last dimension unbounded, and padded to a whole number of bursts
    (`BURST_WIDTH / width * banks` elements, host.py:334-346);
  * tiles follow each other in dimension-0-fastest order of their indices;
  * element k of that stream lives in bank `banks[k % n]` at position `k // n`;
  * an OUTPUT cell sits `stencil_offset` elements further down the stream than
    the input cell of the same coordinates - the pipeline delay of the dataflow
    kernel, the largest serialised offset of the overall stencil window
    (host.py:872-880, core.py:782-785) - and only cells whose whole window lies
    inside their tile are valid (host.py:841-858).
Nothing here touches the GPU: these are O(N) host-side index maps in numpy.
"""
import numpy as np


def serialize(vec, tile_size):
  """Offset of `vec` in a tile linearised with pitch tile_size (dimension 0
  fastest; reference src/soda/util.py:4-7)."""
  out, pitch = 0, 1
  for d, v in enumerate(vec):
    out += v * pitch
    if d < len(tile_size) - 1:
      pitch *= tile_size[d]
  return out


def stencil_constants(stencil, tile_size=None):
  """STENCIL_DIM_d, the window offset, STENCIL_DISTANCE and the per-output
  stream delay of a frontend Stencil, as the reference computes them for its
  generated host (host.py:1183-1197, :872-880; core.py:782-835)."""
  tile = list(stencil.tile_size if tile_size is None else tile_size)
  tile = tile[:stencil.dim - 1] + [0]
  windows = stencil.overall_windows()
  first = windows[stencil.output_names[0]]
  dim = stencil.dim
  offset = tuple(-min(p[d] for p in first) for d in range(dim))
  extent = tuple(max(p[d] for p in first) - min(p[d] for p in first) + 1
                 for d in range(dim))
  distance = max(serialize(p, tile) for p in first) + serialize(offset, tile)
  delays = {}
  for name, pts in windows.items():
    off = tuple(-min(p[d] for p in pts) for d in range(dim))
    dist = max(serialize(p, tile) for p in pts) + serialize(off, tile)
    delays[name] = dist - serialize(off, tile)
  # the copy-back loops are bounded by the window through the FIRST input only
  # (host.py:832-858): for programs with several inputs (denoise2d: `f` is read
  # at the cell itself) they copy back cells the other inputs' windows do not cover
  via_first = stencil.overall_windows(inputs=stencil.input_names[:1])[
      stencil.output_names[0]] or ((0,) * dim,)
  back_offset = tuple(-min(p[d] for p in via_first) for d in range(dim))
  back_extent = tuple(max(p[d] for p in via_first) - min(p[d] for p in via_first) + 1
                      for d in range(dim))
  return dict(stencil_dim=extent, window_offset=offset, stencil_distance=distance,
              stencil_offset=delays, copy_back_offset=back_offset,
              copy_back_dim=back_extent)


class TiledLayout:
  """Geometry of the reference layout for arrays of extents `dims` (dimension 0
  fastest).  tile_size/stencil_dim: per dimension (entries of the last dimension
  are ignored); burst_width in bits; elem_bits the element width; banks the DRAM
  banks of the array in order (`dram` of the statement, host.py:276-281);
  input_banks: the first input's banks, when the array is an output."""

  def __init__(self, dims, tile_size, stencil_dim, burst_width, elem_bits,
               banks=(0,), input_banks=None):
    self.dims = tuple(int(v) for v in dims)
    self.dim = len(self.dims)
    self.tile = tuple(int(v) for v in tile_size[:self.dim - 1])
    self.stencil_dim = tuple(int(v) for v in stencil_dim)
    self.banks = tuple(banks)
    n = len(self.banks)
    self.step = tuple(t - s + 1 for t, s in zip(self.tile, self.stencil_dim))
    if any(v < 1 for v in self.step):
      raise ValueError('tile size must exceed the stencil window')
    # host.py:259-262
    self.tile_num = tuple((self.dims[d] - self.stencil_dim[d] + 1 + self.tile[d] -
                           self.stencil_dim[d]) // self.step[d]
                          for d in range(self.dim - 1))
    self.per_bank_burst = burst_width // elem_bits
    self.burst_elems = self.per_bank_burst * n
    pixels = int(np.prod(self.tile, dtype=np.int64)) * self.dims[-1]
    # the number of bursts per tile is counted with the FIRST INPUT's banks, for
    # the outputs too (host.py:338-346): input_banks when laying out an output
    n_in = len(input_banks) if input_banks is not None else n
    bursts = (pixels - 1) // (self.per_bank_burst * n_in) + 1
    self.tile_stream = bursts * self.burst_elems
    # With fewer output banks than input banks the reference's output tiles are
    # SHORTER than a tile's cells (a defect of host.py:342-346): consecutive tiles
    # then overlap in the stream.  Reading such buffers is still well defined
    # (unpack); writing them is not (export_output refuses).
    self.tiles_overlap = self.tile_stream < pixels

  def actual_tile(self, index):
    """Extents of tile `index` (tuple over the tiled dimensions)."""
    return tuple(self.dims[d] - self.step[d] * index[d]
                 if index[d] == self.tile_num[d] - 1 else self.tile[d]
                 for d in range(self.dim - 1))

  def bank_elems(self, stencil_distance):
    """Elements of one bank's buffer (host.py:372-385)."""
    tiles = int(np.prod(self.tile_num, dtype=np.int64))
    n = len(self.banks)
    return tiles * self.tile_stream // n + \
        ((stencil_distance - 1) // self.burst_elems + 1) * self.per_bank_burst

  def _tile_indices(self):
    return np.ndindex(*reversed(self.tile_num))     # dimension 0 fastest last

  def _map(self, lo, hi_margin, delay):
    """(stream offsets, flat original offsets) of every cell that tile loops
    with per-dimension bounds [lo_d, size_d - hi_margin_d) visit."""
    streams, originals = [], []
    strides = [1]
    for d in range(1, self.dim):
      strides.append(strides[-1] * self.dims[d - 1])
    pitches = [1]
    for d in range(1, self.dim):
      pitches.append(pitches[-1] * self.tile[d - 1])
    for rev in self._tile_indices():
      index = tuple(reversed(rev))
      actual = self.actual_tile(index) + (self.dims[-1],)
      linear, mult = 0, 1
      for d in range(self.dim - 1):
        linear += index[d] * mult
        mult *= self.tile_num[d]
      axes = [np.arange(lo[d], actual[d] - hi_margin[d], dtype=np.int64)
              for d in range(self.dim)]
      if any(a.size == 0 for a in axes):
        continue
      grids = np.meshgrid(*axes, indexing='ij')
      in_tile = sum(g * pitches[d] for d, g in enumerate(grids))
      orig = sum((g + (index[d] * self.step[d] if d < self.dim - 1 else 0)) *
                 strides[d] for d, g in enumerate(grids))
      streams.append((linear * self.tile_stream + in_tile + delay).ravel())
      originals.append(orig.ravel())
    if not streams:
      return np.zeros(0, np.int64), np.zeros(0, np.int64)
    return np.concatenate(streams), np.concatenate(originals)

  def pack(self, array, stencil_distance=0):
    """Row-major array (numpy shape = reversed(dims)) -> {bank: 1-D buffer} in
    the reference's input layout (host.py:629-686).  Cells of the padding are
    zero."""
    flat = np.ascontiguousarray(array).reshape(-1)
    assert flat.size == int(np.prod(self.dims, dtype=np.int64))
    n = len(self.banks)
    out = {b: np.zeros(self.bank_elems(stencil_distance), dtype=flat.dtype)
           for b in self.banks}
    stream, orig = self._map((0,) * self.dim, (0,) * self.dim, 0)
    for k, b in enumerate(self.banks):
      sel = stream % n == k
      out[b][stream[sel] // n] = flat[orig[sel]]
    return out

  def unpack(self, buffers, window_offset, stencil_offset, dtype=None,
             window_dim=None):
    """{bank: 1-D buffer} in the reference's OUTPUT layout -> row-major array;
    only the cells the reference copies back are written (host.py:823-901), the
    rest is zero.  window_offset / window_dim: `copy_back_offset` /
    `copy_back_dim` of stencil_constants (window_dim defaults to STENCIL_DIM)."""
    n = len(self.banks)
    first = np.asarray(buffers[self.banks[0]])
    out = np.zeros(int(np.prod(self.dims, dtype=np.int64)),
                   dtype=dtype or first.dtype)
    extent = self.stencil_dim if window_dim is None else window_dim
    hi = tuple(extent[d] - 1 - window_offset[d] for d in range(self.dim))
    stream, orig = self._map(tuple(window_offset), hi, stencil_offset)
    for k, b in enumerate(self.banks):
      sel = stream % n == k
      out[orig[sel]] = np.asarray(buffers[b])[stream[sel] // n]
    return out.reshape(tuple(reversed(self.dims)))

  # the opposite directions: data that is ALREADY in the reference layout
  def import_input(self, buffers, dtype=None):
    """{bank: 1-D buffer} in the reference's INPUT layout -> row-major array
    (cells that several overlapping tiles hold are taken from the last one)."""
    n = len(self.banks)
    first = np.asarray(buffers[self.banks[0]])
    out = np.zeros(int(np.prod(self.dims, dtype=np.int64)),
                   dtype=dtype or first.dtype)
    stream, orig = self._map((0,) * self.dim, (0,) * self.dim, 0)
    for k, b in enumerate(self.banks):
      sel = stream % n == k
      out[orig[sel]] = np.asarray(buffers[b])[stream[sel] // n]
    return out.reshape(tuple(reversed(self.dims)))

  def export_output(self, array, window_offset, stencil_offset, stencil_distance,
                    window_dim=None):
    """Row-major result -> {bank: 1-D buffer} in the reference's OUTPUT layout:
    every cell the reference would copy back sits where its kernel would have
    written it; the rest of the buffers is zero."""
    if self.tiles_overlap:
      raise ValueError('output tiles overlap in the reference layout when the '
                       'output has fewer DRAM banks than the first input')
    flat = np.ascontiguousarray(array).reshape(-1)
    n = len(self.banks)
    out = {b: np.zeros(self.bank_elems(stencil_distance), dtype=flat.dtype)
           for b in self.banks}
    extent = self.stencil_dim if window_dim is None else window_dim
    hi = tuple(extent[d] - 1 - window_offset[d] for d in range(self.dim))
    stream, orig = self._map(tuple(window_offset), hi, stencil_offset)
    for k, b in enumerate(self.banks):
      sel = stream % n == k
      out[b][stream[sel] // n] = flat[orig[sel]]
    return out


def run_in_reference_layout(program, stencil, in_buffers, dims, tile_size=None,
                            banks_in=(0,), banks_out=(0,), iterate=None):
  """A whole run for a caller that holds its data in the reference's DRAM layout:
  `in_buffers` = {input name: {bank: 1-D buffer}} as the reference's host would
  have handed them to the FPGA; returns {output name: {bank: 1-D buffer}} as its
  kernel would have left them (cells the reference would not copy back: zero).
  `program` is a runtime.host.Program of the same stencil (the HIP path)."""
  tile = list(stencil.tile_size if tile_size is None else tile_size)
  c = stencil_constants(stencil, tile)
  bits = {n: np.asarray(next(iter(b.values()))).dtype.itemsize * 8
          for n, b in in_buffers.items()}
  arrays = []
  for name in stencil.input_names:
    lay = TiledLayout(dims, tile, c['stencil_dim'], stencil.burst_width, bits[name],
                      banks_in)
    arrays.append(lay.import_input(in_buffers[name]))
  results = program.run_numpy(arrays, iterate=iterate or stencil.iterate)
  out = {}
  for name, result in zip(stencil.output_names, results):
    lay = TiledLayout(dims, tile, c['stencil_dim'], stencil.burst_width,
                      result.dtype.itemsize * 8, banks_out, input_banks=banks_in)
    out[name] = lay.export_output(result, c['copy_back_offset'],
                                  c['stencil_offset'][name], c['stencil_distance'],
                                  window_dim=c['copy_back_dim'])
  return out
