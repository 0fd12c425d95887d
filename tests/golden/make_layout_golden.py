#!/opt/conda/bin/python3.9
"""Fixtures for the reference's tiled / burst-aligned DRAM layout
(tests/golden/layout.*.npz), made from the REAL reference like make_golden.py:

    /opt/conda/bin/python3.9 tests/golden/make_layout_golden.py

The reference's `host.print_code` emits its host program for a stencil built with
small tile sizes and chosen DRAM banks; the tile-count, bank-map, burst-alignment
and buffer-size statements, the "tiling" loop nest (host.py:629-686) and the copy
back loop nest (host.py:823-901) are cut out of that text, wrapped in a harness,
compiled with g++ and run.  Stored: the input arrays, the per-bank buffers the
reference's tiling loops produce from them, random per-bank OUTPUT buffers and
the arrays the reference's copy-back loops produce from those.  Only data; none
of the reference's text."""
import json
import os
import re
import subprocess
import sys
import tempfile

import numpy as np

sys.argv = sys.argv[:1]
import make_golden as mg          # noqa: E402  (the reader + reference imports)

HERE = os.path.dirname(os.path.abspath(__file__))

CASES = [   # app, iterate, tile sizes, dims, banks of the inputs / the outputs
    ('blur', 1, [16], (37, 29), [0], [0]),
    ('blur', 1, [12], (50, 9), [0, 2], [1, 3]),
    ('jacobi2d', 1, [16], (40, 21), [0], [1]),
    ('jacobi2d', 3, [24], (64, 20), [1, 0, 3], [2, 0]),
    ('sobel2d', 1, [32], (32, 11), [0], [0]),
    ('jacobi3d', 1, [8, 6], (13, 11, 7), [0], [0]),
    ('jacobi3d', 2, [12, 9], (20, 18, 6), [0, 1], [2, 3]),
    ('denoise2d', 1, [20], (41, 13), [0], [0]),
]


def build(app, iterate, tile, banks_in, banks_out):
  path = os.path.join(mg.REF, 'tests/src', app + '.soda')
  with open(path) as f:
    prog = mg.read_program(f.read())
  mg.core._overall_stencil_window_cache.clear()
  return mg.core.Stencil(
      burst_width=prog['burst_width'], iterate=iterate,
      dram_in='.'.join(map(str, banks_in)), dram_out='.'.join(map(str, banks_out)),
      app_name=prog['app_name'], input_stmts=prog['inputs'],
      param_stmts=[], local_stmts=prog['locals'], output_stmts=prog['outputs'],
      dim=prog['dim'], tile_size=list(tile) + [0],
      unroll_factor=prog['unroll_factor'])


def between(text, start, stop, include_start=True):
  i = text.index(start)
  j = text.index(stop, i)
  return text[i if include_start else i + len(start):j]


def harness(st, text, macros):
  dim = st.dim
  names_in, names_out = list(st.input_names), list(st.output_names)
  tile_nums = between(text, '// allocate buffer for tiled input/output',
                      '// change #bank')
  bank_maps = between(text, 'unordered_map<string, array<bool',
                      'if (const char* env_var_char')
  align = between(text, '// align each linearized tile', '// prepare for opencl')
  sizes = '\n'.join(re.findall(r'(?m)^\s*uint64_t var_\w+_buf_size = .*$', text))
  tiling = between(text, '// tiling', 'err = 0;')
  k = text.index('CL_MAP_READ')
  k2 = text.index('for(int32_t tile_index_dim_', k)
  untiling = text[k2:text.index('read_event_ptr = read_events;', k2)]
  ctype = {n: st.tensors[n].c_type for n in names_in + names_out}
  src = ['#include <cstdint>', '#include <cstdio>', '#include <cstdlib>',
         '#include <cstring>', '#include <string>', '#include <array>',
         '#include <unordered_map>', 'using namespace std;']
  for k_, v in macros.items():
    src.append('#define %s %s' % (k_, v))
  src.append('int main(int argc, char** argv) {')
  src.append('  int32_t dims[4] = {1, 1, 1, 1};')
  src.append('  for (int d = 0; d < %d; ++d) dims[d] = atoi(argv[2 + d]);' % dim)
  src.append('  size_t n = 1; for (int d = 0; d < %d; ++d) n *= dims[d];' % dim)
  for name in names_in + names_out:
    for d in range(dim):
      src.append('  int32_t %s_size_dim_%d = dims[%d]; (void)%s_size_dim_%d;'
                 % (name, d, d, name, d))
      src.append('  int32_t var_%s_stride_%d = %s; (void)var_%s_stride_%d;' % (
          name, d, '*'.join(['1'] + ['dims[%d]' % x for x in range(d)]), name, d))
    src.append('  %s* var_%s = new %s[n]();' % (ctype[name], name, ctype[name]))
  src.append('  char path[4096];')
  src.append('  FILE* fi = fopen(argv[1], "rb");')
  for name in names_in:
    src.append('  if (fread(var_%s, sizeof(%s), n, fi) != n) return 3;'
               % (name, ctype[name]))
  src.append('  fclose(fi);')
  src += [tile_nums, bank_maps, align, sizes]
  for name in names_in + names_out:
    src.append('  %s* var_%s_buf_bank[4] = {};' % (ctype[name], name))
    src.append('  for (int b = 0; b < 4; ++b) if (use_bank["%s"][b]) { '
               'var_%s_buf_bank[b] = (%s*)calloc(var_%s_buf_size, 1); }'
               % (name, name, ctype[name], name))
  src.append(tiling)
  for name in names_in:
    src.append('  for (int b = 0; b < 4; ++b) if (use_bank["%s"][b]) { '
               'snprintf(path, sizeof path, "%%s.in.%s.%%d", argv[1], b); '
               'FILE* fo = fopen(path, "wb"); fwrite(var_%s_buf_bank[b], 1, '
               'var_%s_buf_size, fo); fclose(fo); }' % (name, name, name, name))
  # output side: the harness is handed the per-bank buffers
  for name in names_out:
    src.append('  for (int b = 0; b < 4; ++b) if (use_bank["%s"][b]) { '
               'snprintf(path, sizeof path, "%%s.outbuf.%s.%%d", argv[1], b); '
               'FILE* fb = fopen(path, "rb"); if (!fb || fread(var_%s_buf_bank[b], '
               '1, var_%s_buf_size, fb) != var_%s_buf_size) return 5; fclose(fb); }'
               % (name, name, name, name, name))
  src.append(untiling)
  for name in names_out:
    src.append('  snprintf(path, sizeof path, "%%s.out.%s", argv[1]); { FILE* fo = '
               'fopen(path, "wb"); fwrite(var_%s, sizeof(%s), n, fo); fclose(fo); }'
               % (name, name, ctype[name]))
  # sizes, for the python side to know how much to supply
  src.append('  snprintf(path, sizeof path, "%s.sizes", argv[1]); { FILE* fo = '
             'fopen(path, "w");')
  for name in names_in + names_out:
    src.append('  fprintf(fo, "%s %%llu\\n", (unsigned long long)var_%s_buf_size);'
               % (name, name))
  src.append('  fclose(fo); }')
  src.append('  return 0; }')
  return '\n'.join(src) + '\n'


def main():
  rng = np.random.default_rng(mg.SEED)
  meta = {}
  with tempfile.TemporaryDirectory() as wd:
    for idx, (app, iterate, tile, dims, b_in, b_out) in enumerate(CASES):
      st = build(app, iterate, tile, b_in, b_out)
      ana, text = mg.analysis_of(st)
      macros = dict(ana['macros'])
      for d, t in enumerate(tile):
        macros['TILE_SIZE_DIM_%d' % d] = t
      cpp = os.path.join(wd, 'l%d.cpp' % idx)
      exe = os.path.join(wd, 'l%d' % idx)
      with open(cpp, 'w') as f:
        f.write(harness(st, text, macros))
      subprocess.check_call(['g++', '-std=c++11', '-O1', '-Wno-unused-result',
                             '-Wno-unused-variable', cpp, '-o', exe])
      inputs = mg.make_inputs(st, dims, 'random', rng)
      data = os.path.join(wd, 'd%d.bin' % idx)
      with open(data, 'wb') as f:
        for a in inputs:
          f.write(a.tobytes())
      # dry run with oversized zero output buffers: learns the buffer sizes the
      # reference's own size statements compute
      for name in st.output_names:
        for b in b_out:
          with open('%s.outbuf.%s.%d' % (data, name, b), 'wb') as f:
            f.write(b'\0' * (1 << 24))
      subprocess.check_call([exe, data] + [str(d) for d in dims])
      sizes = {}
      for line in open(data + '.sizes'):
        nm, sz = line.split()
        sizes[nm] = int(sz)
      payload, out_bufs = {}, {}
      for name in st.output_names:
        dt = mg.NP_TYPES[st.tensors[name].c_type]
        for b in b_out:
          n = sizes[name] // np.dtype(dt).itemsize
          if np.issubdtype(dt, np.floating):
            buf = rng.random(n, dtype=np.float32).astype(dt)
          else:
            buf = rng.integers(0, 60000, size=n).astype(dt)
          out_bufs[(name, b)] = buf
          with open('%s.outbuf.%s.%d' % (data, name, b), 'wb') as f:
            f.write(buf.tobytes())
          payload['outbuf_%s_%d' % (name, b)] = buf
      subprocess.check_call([exe, data] + [str(d) for d in dims])
      shape = tuple(reversed(dims))
      for name, a in zip(st.input_names, inputs):
        payload['in_' + name] = a
        dt = a.dtype
        for b in b_in:
          payload['inbuf_%s_%d' % (name, b)] = np.fromfile(
              '%s.in.%s.%d' % (data, name, b), dtype=dt)
      for name in st.output_names:
        dt = mg.NP_TYPES[st.tensors[name].c_type]
        payload['out_' + name] = np.fromfile('%s.out.%s' % (data, name),
                                             dtype=dt).reshape(shape)
      fx = 'layout.%s.iter%d.%s.npz' % (app, iterate, 'x'.join(map(str, dims)))
      np.savez_compressed(os.path.join(HERE, fx), **payload)
      meta[fx] = dict(app=app, iterate=iterate, tile_size=list(tile),
                      dims=list(dims), banks_in=b_in, banks_out=b_out,
                      burst_width=macros['BURST_WIDTH'],
                      stencil_distance=macros['STENCIL_DISTANCE'])
      print('wrote', fx)
  with open(os.path.join(HERE, 'layout_manifest.json'), 'w') as f:
    json.dump(meta, f, indent=1, sort_keys=True)


if __name__ == '__main__':
  main()
