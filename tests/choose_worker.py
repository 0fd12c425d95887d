"""Worker of tests/test_dist.py::test_all_ranks_choose_the_same_exchange: one rank of a
gloo group running dist.choose_exchange on step times that differ per rank."""
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'soda-compiler_amd')):
  if p not in sys.path:
    sys.path.insert(0, p)

from soda_hip.runtime import dist as sdist         # noqa: E402


def main():
  times = json.loads(sys.argv[1])          # [[seconds per candidate] per rank]
  out_dir = sys.argv[2]
  rank = int(os.environ['RANK'])
  dist.init_process_group(backend='gloo')
  candidates = [(e, o) for e in sdist.exchange_candidates(2048, 1, 24, 1000)
                for o in (False, True)]
  mine = dict(zip(candidates, times[rank]))
  calls = []

  def time_step(exchange, overlapped, repeats):
    calls.append((exchange, overlapped))
    # (noisy repeats around the rank's figure: the fastest one counts)
    return [mine[(exchange, overlapped)] * (1.0 + 0.1 * k) for k in range(repeats)]

  def reduce_max(seconds):
    t = torch.tensor(list(seconds), dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return t.tolist()

  table, chosen = sdist.choose_exchange(candidates, time_step, reduce_max)
  assert calls == candidates               # every rank times every pair, in one order
  with open(os.path.join(out_dir, 'choice%d.json' % rank), 'w') as f:
    json.dump(dict(table=table, chosen=chosen), f)
  dist.barrier()
  dist.destroy_process_group()


if __name__ == '__main__':
  main()
