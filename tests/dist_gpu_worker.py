"""Worker of the multi-process GPU rehearsal (tests/test_gpu_parity.py): one rank
of a gloo group; all ranks share the ONE GPU of the box and run the REAL kernels
through soda_hip.runtime.dist (HipEngine, run_slab, the serial and the stream-
overlapped schedule).  Ghost rows travel through the host (gloo) because RCCL
refuses two ranks on one device; everything else is the production path."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'soda-compiler_amd')):
  if p not in sys.path:
    sys.path.insert(0, p)

from soda_hip import frontend                      # noqa: E402
from soda_hip.codegen import spec as specmod       # noqa: E402
from soda_hip.runtime import dist as sdist, host   # noqa: E402


def main():
  app, size, iterate, exchange, out_dir, mode = sys.argv[1:7]
  dims = [int(v) for v in size.split('x')]
  iterate, exchange = int(iterate), int(exchange)
  rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
  dist.init_process_group(backend='gloo')
  torch.cuda.set_device(0)
  st = frontend.load(os.path.join(ROOT, 'tests', 'samples', app + '.soda'),
                     iterate=iterate)
  spec = specmod.spec_from_stencil(st)
  blob = os.path.join(ROOT, 'soda-compiler_amd', 'blobs', app + '.hsaco')
  prog = host.open_program(blob=blob, spec=specmod.spec_from_stencil(
      frontend.load(os.path.join(ROOT, 'tests', 'samples', app + '.soda'))))
  r_lo, r_hi = spec['radius']['lo'][-1], spec['radius']['hi'][-1]
  plan = sdist.SlabPlan(dims, rank, world, r_lo, r_hi, exchange)
  full = np.random.default_rng(99).random(tuple(reversed(dims)), dtype=np.float32)
  dev = torch.device('cuda', 0)
  a = torch.zeros(tuple(reversed(plan.local_dims)), dtype=torch.float32, device=dev)
  a[plan.ghost_lo:plan.ghost_lo + plan.own] = torch.from_numpy(
      full[plan.start:plan.stop]).to(dev)
  b, c = torch.zeros_like(a), torch.zeros_like(a)
  table = specmod.iteration_margins(spec, iterate)

  def margins_of(k):
    return ((0,) * len(dims), (0,) * len(dims)) if k == 0 else table[k - 1]

  order = sdist.StreamSchedule(torch, host_sync=True) if mode == 'overlap' else \
      sdist.TimedSerialSchedule(torch, host_sync=True)
  result, exchanges = sdist.run_slab(sdist.HipEngine(prog, torch), plan, [a, b, c],
                                     iterate, margins_of, dist, schedule=order)
  torch.cuda.synchronize()
  own = result[plan.ghost_lo:plan.ghost_lo + plan.own].cpu().numpy()
  np.save(os.path.join(out_dir, 'rank%d.npy' % rank), own)
  with open(os.path.join(out_dir, 'rank%d.txt' % rank), 'w') as f:
    f.write('%d %d %d %d %.3f\n' % (plan.start, plan.stop, plan.exchange, exchanges,
                                    order.exchange_ms()))
  dist.barrier()
  prog.close()
  dist.destroy_process_group()


if __name__ == '__main__':
  main()
