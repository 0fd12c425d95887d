#!/usr/bin/env python3
"""The reference README builds the generated host with `g++ -std=c++11 -fopenmp`
and NO optimisation flag (README.md:95-96); SURVEY.md 8(d) asks for that figure
once on BASELINE cfg1 (blur 2000 x 100), next to the -O3 -march=native figure the
bench reports.  Times the CPU oracle (the port of the emitted loop nest) both ways."""
import sys as _sys
if len(_sys.argv) > 1 and _sys.argv[1] in ('-h', '--help'):   # usage = the text above
  print(__doc__)
  _sys.exit(0)
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'soda-compiler_amd')]
import numpy as np  # noqa: E402
from soda_hip import frontend  # noqa: E402
from soda_hip.codegen import spec as specmod  # noqa: E402
from oracle import soda_oracle  # noqa: E402

st = frontend.load(os.path.join(ROOT, 'tests', 'samples', 'blur.soda'))
spec = specmod.spec_from_stencil(st)
dims = (2000, 100)
inputs = soda_oracle.reference_init(spec, dims)
for label, flags in (('README flags (-std=c++11 -fopenmp, no -O)', ()),
                     ('-O3 -march=native', ('-O3', '-march=native'))):
  orc = soda_oracle.Oracle(spec, flags=flags, build_dir=os.environ.get('TMPDIR', '/tmp'))
  quota = orc.cpu_quota()
  best = None
  for threads in sorted({1, max(1, quota // 2), quota}):
    orc.set_threads(threads)
    orc.run(inputs)
    t0 = time.perf_counter()
    n = 200
    for _ in range(n):
      orc.run(inputs)
    dt = (time.perf_counter() - t0) / n
    if best is None or dt < best[0]:
      best = (dt, threads)
  cells = specmod.valid_cells(spec, list(dims), 1)
  print('cfg1 blur 2000x100, %s: %.1f us per run (%d threads) = %.3f Gcell-updates/s' % (
      label, best[0] * 1e6, best[1], cells / best[0] / 1e9))
