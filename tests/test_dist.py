"""Multi-rank slab decomposition (soda_hip.runtime.dist) under gloo on the CPU:
world sizes 2 and 3, exchange periods that do and do not divide the iteration
count, symmetric and one-sided stencil windows."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from soda_hip import frontend
from soda_hip.codegen import spec as specmod
from soda_hip.runtime import dist as sdist
from oracle import soda_oracle

from conftest import ROOT, SAMPLES


def free_port():
  s = socket.socket()
  s.bind(('127.0.0.1', 0))
  port = s.getsockname()[1]
  s.close()
  return port


def run_world(tmp_path, world, app, dims, iterate, exchange, overlap=False, recut=False):
  env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(free_port()),
             WORLD_SIZE=str(world), OMP_NUM_THREADS='2')
  procs = []
  for rank in range(world):
    procs.append(subprocess.Popen(
        [sys.executable, os.path.join(ROOT, 'tests', 'dist_worker.py'), app,
         'x'.join(map(str, dims)), str(iterate), str(exchange), str(tmp_path)] +
        [('recut' if recut else '') + ('+' if recut and overlap else '') +
         ('' if not overlap else 'overlap' if overlap is True else 'overlap:%d' % overlap)],
        env=dict(env, RANK=str(rank), LOCAL_RANK=str(rank))))
  for p in procs:
    assert p.wait(timeout=300) == 0
  st = frontend.load(os.path.join(SAMPLES, app + '.soda'), iterate=iterate)
  spec = specmod.spec_from_stencil(st)
  dt = np.dtype(specmod.NUMPY_NAME[spec['inputs'][0]['c_type']])
  rng = np.random.default_rng(99)
  shape = tuple(reversed(dims))
  if dt.kind == 'f':
    full = rng.random(shape, dtype=np.float32).astype(dt)
  else:
    full = rng.integers(0, 65536, size=shape).astype(dt)
  orc = soda_oracle.Oracle(spec)
  want = orc.run([full], iterate=iterate)[spec['outputs'][0]]
  sl = orc.valid_slices(tuple(dims), iterate)
  got = np.zeros_like(want)
  meta = []
  for rank in range(world):
    start, stop, ex, n_ex = map(int, open(
        os.path.join(tmp_path, 'rank%d.txt' % rank)).read().split())
    got[start:stop] = np.load(os.path.join(tmp_path, 'rank%d.npy' % rank))
    meta.append((start, stop, ex, n_ex))
  assert want[sl].size > 0
  assert np.array_equal(got[sl], want[sl])
  return meta


@pytest.mark.parametrize('world,app,dims,iterate,exchange,overlap', [
    (2, 'jacobi2d', (64, 50), 7, 3, False),
    (2, 'jacobi2d', (40, 61), 6, 6, False),
    (3, 'jacobi2d', (48, 47), 5, 2, False),
    (2, 'seidel2d', (56, 40), 4, 2, False),
    (2, 'blur', (70, 44), 3, 1, False),   # window reaches only towards higher indices
    (3, 'blur', (70, 45), 4, 3, False),
    # 3-D: ghost PLANES
    (2, 'jacobi3d', (20, 18, 30), 5, 2, False),
    (3, 'heat3d', (16, 14, 40), 4, 3, False),
    # the overlapping schedule's order: boundary bands, exchange, interior
    (2, 'jacobi2d', (64, 70), 9, 3, True),
    (3, 'jacobi2d', (48, 90), 8, 3, True),
    (3, 'blur', (70, 75), 6, 2, True),
    (3, 'jacobi3d', (20, 18, 60), 6, 2, True),
    # ... with super-steps of 3 and 4 LAUNCHES (overlap = launch depth of the CPU
    # engine): the intermediate launches of a piece must not write the shared array
    (2, 'jacobi2d', (40, 120), 13, 6, 2),
    (3, 'jacobi2d', (40, 150), 12, 8, 2),
    (3, 'jacobi3d', (20, 18, 90), 7, 6, 2),
    (3, 'blur', (60, 110), 9, 6, 2),
    # slabs too thin to cut bands from (10 own rows, 3 + 3 to send)
    (3, 'jacobi2d', (48, 30), 8, 3, True),
    # BASELINE cfg5's shape in small: planes cut into slabs THINNER than the iteration
    # count, so the edge ranks run out of valid cells (their boxes are empty for the last
    # super-steps) while the middle ranks still need their rows - nobody may stall
    (4, 'jacobi3d', (20, 18, 24), 8, 2, False),
    (4, 'jacobi3d', (26, 24, 32), 10, 4, True),
])
def test_slabs_match_single_process(tmp_path, world, app, dims, iterate, exchange,
                                    overlap):
  meta = run_world(tmp_path, world, app, dims, iterate, exchange, overlap)
  h = dims[-1]
  assert meta[0][0] == 0 and meta[-1][1] == h
  for (a0, a1, _, _), (b0, b1, _, _) in zip(meta, meta[1:]):
    assert a1 == b0


@pytest.mark.parametrize('world,app,dims,iterate,exchange,overlap', [
    (2, 'jacobi2d', (64, 50), 7, 3, False),
    (3, 'jacobi2d', (48, 47), 5, 2, False),
    (2, 'blur', (70, 44), 3, 1, False),          # one-sided window: the cuts move one way
    (3, 'blur', (70, 75), 6, 2, True),
    (3, 'heat3d', (16, 14, 40), 4, 3, False),
    (4, 'jacobi2d', (40, 96), 12, 4, False),
    (4, 'jacobi2d', (40, 96), 12, 4, True),
    (4, 'jacobi2d', (40, 120), 13, 6, 2),        # pieces of several launches
    # cfg5's shape in small: the valid range ends far thinner than a level-0 slab, rows
    # change owner every super-step and partners are not always the nearest rank
    (4, 'jacobi3d', (20, 18, 24), 8, 2, False),
    (4, 'jacobi3d', (26, 24, 32), 10, 4, True),
    (8, 'jacobi3d', (52, 50, 64), 24, 4, False),
    (8, 'jacobi3d', (52, 50, 64), 24, 4, True),
    (8, 'jacobi2d', (72, 128), 30, 8, False),
    (8, 'jacobi2d', (72, 128), 30, 8, True),
    # more ranks than rows at the end (8 valid rows on 8 ranks, then fewer than ranks)
    (8, 'jacobi2d', (40, 40), 17, 3, False),
])
def test_recut_slabs_match_single_process(tmp_path, world, app, dims, iterate, exchange,
                                          overlap):
  """Slabs cut afresh every super-step: same cells as one process,
  in both orders; the ranks' final rows tile the final valid range."""
  meta = run_world(tmp_path, world, app, dims, iterate, exchange, overlap, recut=True)
  st = frontend.load(os.path.join(SAMPLES, app + '.soda'), iterate=iterate)
  spec = specmod.spec_from_stencil(st)
  lo, hi = specmod.iteration_margins(spec, iterate)[-1]
  assert meta[0][0] == lo[-1] and meta[-1][1] == dims[-1] - hi[-1]
  for (a0, a1, _, _), (b0, b1, _, _) in zip(meta, meta[1:]):
    assert a1 == b0
  sizes = [a1 - a0 for a0, a1, _, _ in meta]
  assert max(sizes) - min(sizes) <= 1


def test_recut_plan_geometry():
  # cfg4 on 8 ranks: every super-step's output rows cut evenly, rows change owner
  plans = [sdist.RecutPlan([16384, 16384], r, 8, 1, 1, 144, 1000) for r in range(8)]
  p = plans[0]
  assert [s for s in p.steps] == [(144 * i, min(144, 1000 - 144 * i)) for i in range(7)]
  assert p.owned[0] == sdist.slab_bounds(16384, 8)
  assert p.cuts[0][0] == 144 and p.cuts[0][-1] == 16384 - 144
  assert p.final[0][0] == 1000 and p.final[-1][1] == 15384
  for s in range(len(p.steps)):
    widths = [b - a for a, b in zip(p.cuts[s], p.cuts[s][1:])]
    assert max(widths) - min(widths) <= 1
    # what one rank sends is what the other receives
    for r in range(8):
      sends, recvs = plans[r].messages(s)
      for q, rows in sends:
        assert (r, rows) in plans[q].messages(s)[1]
      for q, rows in recvs:
        assert (r, rows) in plans[q].messages(s)[0]
      # own rows + received rows cover exactly what the rank reads
      need = plans[r].need[s][r]
      got = sorted([plans[r].owned[s][r]] + [rows for _, rows in recvs])
      covered = need[0]
      for a, b in got:
        if b <= covered or a > covered:
          continue
        covered = max(covered, b)
      assert covered >= need[1]
  # every rank's arrays hold everything it ever reads
  for q in plans:
    for n in q.need:
      a, b = q.local(n[q.rank])
      assert 0 <= a < b <= q.local_extent
    assert q.ghost_lo >= 0 and q.ghost_lo + q.own <= q.local_extent
  # the busiest rank's rows per iteration: 93.9 % of an even share with ghost rows
  # against 87.8 % for the static cut
  ideal = sum(16384 - 2 * k for k in range(1, 1001)) / 8.0
  assert 0.93 < ideal / p.max_rows_per_iteration() < 0.95
  # cfg5 on 8 ranks: the static cut idles the edge ranks after iteration 64
  q = sdist.RecutPlan([512, 512, 512], 0, 8, 1, 1, 4, 200)
  ideal = sum(512 - 2 * k for k in range(1, 201)) / 8.0
  assert ideal / q.max_rows_per_iteration() > 0.93
  assert q.final[0] == (200, 214) and q.final[7] == (298, 312)
  # bands: the rows other ranks read next, the interior the rest
  mid = plans[3]
  bands, interior = mid.pieces(0)
  lo, hi = mid.cuts[0][3], mid.cuts[0][4]
  assert bands[0][0] == lo and bands[-1][1] == hi and interior == (bands[0][1], bands[1][0])
  assert bands[0][1] == mid.need[1][2][1] and bands[1][0] == mid.need[1][4][0]
  assert mid.pieces(len(mid.steps) - 1) is None
  # one rank: no messages, one piece
  solo = sdist.RecutPlan([64, 48], 0, 1, 1, 1, 4, 10)
  assert solo.messages(0) == ([], []) and solo.pieces(0) is None
  assert solo.local_extent == 48 and solo.final_rows == (10, 38)


def test_slab_bounds_and_plan():
  assert sdist.slab_bounds(10, 3) == [(0, 4), (4, 7), (7, 10)]
  assert sdist.slab_bounds(16384, 8)[3] == (6144, 8192)
  p = sdist.SlabPlan([16384, 16384], 3, 8, 1, 1, 48)
  assert (p.ghost_lo, p.ghost_hi, p.own, p.local_extent) == (48, 48, 2048, 2144)
  first = sdist.SlabPlan([16384, 16384], 0, 8, 1, 1, 48)
  assert (first.ghost_lo, first.ghost_hi) == (0, 48)
  # a ghost region can never exceed the neighbour's slab
  tiny = sdist.SlabPlan([64, 20], 1, 4, 1, 1, 100)
  assert tiny.exchange == 5 and tiny.ghost_lo == 5
  # margins: neighbour sides are fully valid, global sides carry the margin
  lo, hi = first.valid_margins(10, lambda k: ((k, k), (k, k)))
  assert (lo, hi) == ([10, 10], [10, 0])
  assert sdist.auto_exchange(2048, 1, 24, 1000) == 144      # 8 ranks of 16384 rows
  assert sdist.auto_exchange(8192, 1, 24, 1000) == 192
  assert sdist.auto_exchange(64, 1, 4, 200) == 4            # 8 ranks of 512 planes
  assert sdist.auto_exchange(64, 1, 12, 1000) == 12
  assert sdist.auto_exchange(2048, 1, 12, 30) == 30
  # a slab thinner than the stencil reach cannot be exchanged correctly
  with pytest.raises(ValueError):
    sdist.SlabPlan([64, 3], 1, 4, 1, 1, 4)
  # the overlapping schedule's pieces tile the own rows
  mid = sdist.SlabPlan([16384, 16384], 3, 8, 1, 1, 48)
  bands, interior = sdist.band_plan(mid, 48)
  assert bands == [(0, 144, True, True), (2000, 2144, True, True)]
  assert interior == (48, 2096, True, True)
  bands, interior = sdist.band_plan(first, 20)
  assert bands == [(2048 - 48 - 20, 2048 + 20, True, True)]
  assert interior == (0, 2048 - 48 + 20, False, True)


def test_exchange_candidates():
  # 1, 2, 4, 8 x the deepest kernel, within the thinnest slab and the iteration count
  assert sdist.exchange_candidates(2048, 1, 24, 1000) == [24, 48, 96, 192]
  assert sdist.exchange_candidates(8192, 1, 24, 100) == [24, 48, 96, 100]
  assert sdist.exchange_candidates(64, 1, 4, 200) == [4, 8, 16, 32]
  assert sdist.exchange_candidates(64, 2, 12, 200) == [12, 24, 32]     # 64 rows / reach 2
  assert sdist.exchange_candidates(5, 1, 24, 1000) == [5]
  assert sdist.exchange_candidates(2048, 1, 1, 1) == [1]


def test_choose_exchange_takes_the_fastest_row_of_the_slowest_ranks():
  pairs = [(24, False), (24, True), (48, False), (48, True)]
  seconds = {(24, False): 0.031, (24, True): 0.034, (48, False): 0.029, (48, True): 0.030}
  # (a second rank that is slower on the locally fastest pair)
  other = {(24, False): 0.030, (24, True): 0.030, (48, False): 0.040, (48, True): 0.0305}
  calls = []

  def time_step(e, o, repeats):
    calls.append(repeats)
    return [seconds[(e, o)] * f for f in (1.3, 1.0, 1.1)[:repeats]]   # the fastest counts
  reduce_max = lambda mine: [max(m, other[p]) for m, p in zip(mine, pairs)]   # noqa: E731
  table, chosen = sdist.choose_exchange(pairs, time_step, reduce_max)
  assert calls == [sdist.EXCHANGE_REPEATS] * 4 and sdist.EXCHANGE_REPEATS >= 3
  assert [(r['exchange'], r['overlapped']) for r in table] == pairs
  assert [round(r['ms'], 3) for r in table] == [31.0, 34.0, 40.0, 30.5]
  assert all(r['repeats'] == sdist.EXCHANGE_REPEATS for r in table)
  assert (chosen['exchange'], chosen['overlapped']) == (48, True)
  assert chosen['ms'] == min(r['ms'] for r in table)
  # the incumbent (default period, serial) is kept unless a candidate beats it by the
  # margin: 30.5 against 31.0 ms is 1.6 %
  _, kept = sdist.choose_exchange(pairs, time_step, reduce_max, incumbent=(24, False))
  assert (kept['exchange'], kept['overlapped']) == (24, False)
  _, moved = sdist.choose_exchange(pairs, time_step, reduce_max, incumbent=(24, True))
  assert (moved['exchange'], moved['overlapped']) == (48, True)       # 34.0 -> 30.5
  _, free = sdist.choose_exchange(pairs, time_step, reduce_max, incumbent=(96, False))
  assert (free['exchange'], free['overlapped']) == (48, True)         # not in the table


def test_all_ranks_choose_the_same_exchange(tmp_path):
  """Two gloo ranks whose own step times rank the candidates differently: both must end
  with the table of the slower rank per candidate and the same choice (a rank that
  picked another exchange period would wait for messages nobody sends)."""
  import json
  world = 2
  # candidates: E = 24, 48, 96, 192, each serial and overlapped
  times = [[9.0, 9.5, 8.0, 8.4, 7.0, 7.6, 7.4, 7.9],      # rank 0 likes (96, serial)
           [9.1, 9.2, 7.9, 7.5, 8.8, 7.7, 7.2, 7.3]]      # rank 1 likes (192, serial)
  env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(free_port()),
             WORLD_SIZE=str(world), OMP_NUM_THREADS='1')
  procs = [subprocess.Popen(
      [sys.executable, os.path.join(ROOT, 'tests', 'choose_worker.py'), json.dumps(times),
       str(tmp_path)], env=dict(env, RANK=str(rank), LOCAL_RANK=str(rank)))
           for rank in range(world)]
  for p in procs:
    assert p.wait(timeout=120) == 0
  got = [json.load(open(os.path.join(tmp_path, 'choice%d.json' % r))) for r in range(world)]
  assert got[0] == got[1]
  slowest = [max(a, b) * 1e3 for a, b in zip(*times)]
  assert [r['ms'] for r in got[0]['table']] == pytest.approx(slowest)
  assert (got[0]['chosen']['exchange'], got[0]['chosen']['overlapped']) == (192, False)
