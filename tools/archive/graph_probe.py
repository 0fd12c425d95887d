#!/usr/bin/env python3
"""Does a HIP graph shorten a sweep?  The launches of one soda_hip_sweep captured from the
stream (torch.cuda.CUDAGraph: plain hipStreamBeginCapture / EndCapture around the same C
call) and replayed, against the same sweep issued launch by launch - alternating, because
whichever is measured first also warms the clocks.  Fastest and median of 20 each.
usage: graph_probe.py app N iterate   (run on the GPU box)
Result (round 4, profiles/r04_graph_probe.txt): nothing - what separates two dependent
launches is the drain and refill of the chip, not the doorbell."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'soda-compiler_amd'), os.path.join(ROOT, 'tests')]
import torch
import gpu_util
from soda_hip.runtime import host

app, n, iterate = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
spec = gpu_util.load_spec(app, iterate=iterate)
prog = host.open_program(blob=os.path.join(gpu_util.BLOBS, app + '.hsaco'), spec=spec)
dims = [n] * spec['dim']
a = torch.rand(tuple(reversed(dims)), dtype=torch.float32, device='cuda')
b = torch.zeros_like(a)
side = torch.cuda.Stream()


def plain():
  prog.sweep([a.data_ptr()], [b.data_ptr()], dims, iterate, stream=side.cuda_stream)


def timed(fn, repeats=20):
  for _ in range(3):
    fn()
  side.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  t = []
  for _ in range(repeats):
    e0.record(side)
    fn()
    e1.record(side)
    side.synchronize()
    t.append(e0.elapsed_time(e1) * 1e3)
  t.sort()
  return t[0], t[len(t) // 2]


with torch.cuda.stream(side):
  plain()
  side.synchronize()
  graph = torch.cuda.CUDAGraph()
  graph.capture_begin()
  plain()
  graph.capture_end()
  out = [(name, timed(fn)) for name, fn in (('launches', plain), ('graph', graph.replay),
                                            ('launches', plain), ('graph', graph.replay))]
print('%s %s x%d, %d launches:' % (app, 'x'.join(map(str, dims)), iterate,
                                   len(prog.schedule(dims, iterate))),
      '   '.join('%s %.1f us (median %.1f)' % (name, t[0], t[1]) for name, t in out))
