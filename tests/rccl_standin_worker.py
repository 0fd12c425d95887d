"""Worker of test_run_slab_with_two_and_three_ranks_over_the_rccl_standin: runs
soda_hip_run_slab (the C slab driver: ncclSend / ncclRecv groups + sweeps) with
`world` ranks as host threads of THIS process on the box's one GPU, over the
test-only librccl stand-in (tests/rccl_standin).  No torch here: the stand-in must be
the first object with soname librccl.so in the process, so that libsoda_hip's
dlopen("librccl.so") resolves to it."""
import ctypes
import os
import sys
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'soda-compiler_amd')):
  if p not in sys.path:
    sys.path.insert(0, p)


def main():
  standin_path, app, size, world, iterate, wanted, out_dir = sys.argv[1:8]
  order = int(sys.argv[8]) if len(sys.argv) > 8 else 0      # capi.SLAB_BANDS_FIRST = 1
  options = sys.argv[9:]
  expect_failure = 'expect-failure' in options
  recut = 'cut=recut' in options              # slabs re-cut every super-step
  abort_by_library = 'abort=lib' in options   # soda_hip_slab.abort_on_error
  dims = [int(v) for v in size.split('x')]
  world, iterate, wanted = int(world), int(iterate), int(wanted)
  standin = ctypes.CDLL(standin_path, mode=ctypes.RTLD_GLOBAL)
  from soda_hip import frontend
  from soda_hip.codegen import spec as specmod
  from soda_hip.runtime import capi, host
  hip = ctypes.CDLL('libamdhip64.so')
  sample = os.path.join(ROOT, 'tests', 'samples', app + '.soda')
  if not os.path.exists(sample):
    sample = os.path.join(ROOT, 'tests', 'samples', 'extra', app + '.soda')
  spec = specmod.spec_from_stencil(frontend.load(sample))
  blob = os.path.join(ROOT, 'soda-compiler_amd', 'blobs', app + '.hsaco')
  lib = capi.lib()
  dt = np.dtype(specmod.NUMPY_NAME[spec['inputs'][0]['c_type']])
  rng = np.random.default_rng(99)
  shape = tuple(reversed(dims))
  full = rng.random(shape, dtype=np.float32).astype(dt) if dt.kind == 'f' else \
      rng.integers(0, 65536, size=shape).astype(dt)
  rows = dims[-1]
  r_lo, r_hi = spec['radius']['lo'][-1], spec['radius']['hi'][-1]
  exchange = ctypes.c_int(min(wanted, iterate))
  if not recut:      # (a re-cut run's period is not bounded by the thinnest slab)
    capi.check(lib.soda_hip_slab_exchange(rows, world, r_lo, r_hi, wanted,
                                          ctypes.byref(exchange)))
  comms = (ctypes.c_void_p * world)()
  assert standin.ncclCommInitAll(comms, world, None) == 0
  first_comm = comms[0]        # (aborted communicators are not freed by the stand-in)
  base, extra = divmod(rows, world)
  errors, results = [None] * world, [None] * world
  abort_lock = threading.Lock()

  def abort_all():
    # what a one-process driver does when a rank fails (the generated <app>_multi_gpu):
    # every communicator aborted exactly once - ncclCommAbort is local to its rank, the
    # peers are unblocked by aborting THEIRS
    with abort_lock:
      for r in range(world):
        if comms[r]:
          standin.ncclCommAbort(ctypes.c_void_p(comms[r]))
          comms[r] = None

  def rank_main(rank):
    try:
      prog = host.open_program(blob=blob, spec=spec)   # one plan per host thread
      stream = ctypes.c_void_p()
      assert hip.hipStreamCreate(ctypes.byref(stream)) == 0
      slab = capi.Slab()
      slab.rank, slab.world = rank, world
      slab.reach_lo, slab.reach_hi = r_lo, r_hi
      slab.exchange = exchange.value
      slab.order = order
      slab.cut = capi.SLAB_CUT_RECUT if recut else capi.SLAB_CUT_STATIC
      slab.abort_on_error = 1 if abort_by_library else 0
      for d, n in enumerate(dims):
        slab.dims[d] = n
      slab.own_first = rank * base + min(rank, extra)
      slab.own_last = slab.own_first + base + (1 if rank < extra else 0)
      local = (ctypes.c_int64 * 4)()
      g_lo, res_first, res_last, res_at = (ctypes.c_int64() for _ in range(4))
      capi.check(lib.soda_hip_slab_layout(
          prog.handle, ctypes.byref(slab), iterate, local, ctypes.byref(g_lo),
          ctypes.byref(res_first), ctypes.byref(res_last), ctypes.byref(res_at)))
      if not recut:    # the older query describes the static cut the same way
        local2 = (ctypes.c_int64 * 4)()
        lo2, hi2 = ctypes.c_int64(), ctypes.c_int64()
        capi.check(lib.soda_hip_slab_extent(prog.handle, ctypes.byref(slab), local2,
                                            ctypes.byref(lo2), ctypes.byref(hi2)))
        assert list(local2) == list(local) and lo2.value == g_lo.value == res_at.value
        assert (res_first.value, res_last.value) == (slab.own_first, slab.own_last)
      local_shape = (local[len(dims) - 1],) + shape[1:]
      nbytes = int(np.prod(local_shape)) * dt.itemsize
      arrays = [host.DeviceArray(nbytes) for _ in range(3)]
      slab_in = np.zeros(local_shape, dtype=dt)
      own = slab.own_last - slab.own_first
      slab_in[g_lo.value:g_lo.value + own] = full[slab.own_first:slab.own_last]
      arrays[0].upload(slab_in)
      arrays[1].zero()
      arrays[2].zero()
      capi.check(lib.soda_hip_stream_synchronize(None))
      result, count = ctypes.c_void_p(), ctypes.c_int()
      rc = lib.soda_hip_run_slab(
          prog.handle, ctypes.byref(slab), comms[rank], arrays[0].ptr, arrays[1].ptr,
          arrays[2].ptr, iterate, stream, ctypes.byref(result), ctypes.byref(count))
      if rc:
        message = lib.soda_hip_last_error().decode()
        if abort_by_library:
          # the library has aborted this rank's communicator already
          # (include/soda_hip.h): it must not be aborted or destroyed again
          with abort_lock:
            comms[rank] = None
        lib.soda_hip_stream_synchronize(stream)
        raise RuntimeError('soda_hip_run_slab: %d %s' % (rc, message))
      capi.check(lib.soda_hip_stream_synchronize(stream))
      which = [a for a in arrays if a.ptr == result.value][0]
      out = which.download(local_shape, dt)
      held = res_last.value - res_first.value
      results[rank] = (res_first.value, res_last.value, count.value,
                       out[res_at.value:res_at.value + held].copy())
      prog.close()
    except BaseException as e:   # noqa: BLE001 - reported by the parent
      errors[rank] = e
      # peers blocked in the exchange must not hang: abort every communicator that has
      # not been aborted yet
      abort_all()

  threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
  for t in threads:
    t.start()
  for t in threads:
    t.join(timeout=300)
  if any(t.is_alive() for t in threads):
    print('a rank is still blocked after 300 s', file=sys.stderr)
    os._exit(3)
  for rank, e in enumerate(errors):
    if e is not None:
      print('rank %d: %r' % (rank, e), file=sys.stderr)
  if expect_failure:
    # a rank was made to fail (SODA_HIP_FAIL_RANK): EVERY rank must have come back with
    # an error - nobody hangs, nobody reports success on stale ghost rows
    with open(os.path.join(out_dir, 'errors.txt'), 'w') as f:
      for rank, e in enumerate(errors):
        f.write('%d %s\n' % (rank, 'ok' if e is None else str(e).replace('\n', ' ')))
    with open(os.path.join(out_dir, 'double_abort.txt'), 'w') as f:
      f.write('%d\n' % standin.rccl_standin_double_abort(ctypes.c_void_p(first_comm)))
    for c in comms:
      if c:
        standin.ncclCommDestroy(ctypes.c_void_p(c))
    sys.exit(0)
  if any(e is not None for e in errors):
    sys.exit(2)
  messages, nbytes = ctypes.c_longlong(), ctypes.c_longlong()
  standin.rccl_standin_traffic(ctypes.c_void_p(comms[0]), ctypes.byref(messages),
                               ctypes.byref(nbytes))
  for rank, (first, last, count, own) in enumerate(results):
    np.save(os.path.join(out_dir, 'rank%d.npy' % rank), own)
    with open(os.path.join(out_dir, 'rank%d.txt' % rank), 'w') as f:
      f.write('%d %d %d %d %d %d\n' % (first, last, exchange.value, count,
                                       messages.value, nbytes.value))
  for c in comms:
    if c:
      standin.ncclCommDestroy(ctypes.c_void_p(c))


if __name__ == '__main__':
  main()
