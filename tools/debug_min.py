import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'soda-compiler_amd'), os.path.join(ROOT, 'tests')]
import numpy as np
import torch
from soda_hip import frontend
from soda_hip.codegen import kernel, spec as specmod
from soda_hip.runtime import host
from oracle import soda_oracle
HEAD = 'kernel: t\nburst width: 512\nunroll factor: 1\niterate: 2\ninput %s: a(32, *)\n'
CASES = {
 'int dx direct': ('int32', 'local int32: loc0(0, 0) = a(0, 0) + a(1, 0)\noutput int32: o(0, 0) = loc0(0, 0) - a(2, 0) + a(-2, 0)\n'),
 'float dx direct': ('float', 'local float: loc0(0, 0) = a(0, 0) + a(1, 0)\noutput float: o(0, 0) = loc0(0, 0) - a(2, 0) + a(-2, 0)\n'),
 'int add only': ('int32', 'local int32: loc0(0, 0) = a(0, 0) + a(1, 0)\noutput int32: o(0, 0) = loc0(0, 0) + a(2, 0) + a(-2, 0)\n'),
 'int single stage': ('int32', 'output int32: o(0, 0) = a(0, 0) - a(2, 0) + a(-2, 0)\n'),
 'int dx1': ('int32', 'local int32: loc0(0, 0) = a(0, 0) + a(1, 0)\noutput int32: o(0, 0) = loc0(0, 0) - a(1, 0) + a(-1, 0)\n'),
}
rng = np.random.default_rng(5)
ai = rng.integers(0, 200, size=(41, 333)).astype(np.int32)
for name, (ty, body) in CASES.items():
  a = ai if ty == 'int32' else ai.astype(np.float32)
  spec = specmod.spec_from_stencil(frontend.loads(HEAD % ty + body))
  orc = soda_oracle.Oracle(spec)
  want = orc.run([a], iterate=2)['o']
  sl = orc.valid_slices((333, 41), 2)
  src, table = kernel.generate(spec)
  for opts in (('-fno-slp-vectorize', '-fwrapv'), ('-fno-slp-vectorize', '-fwrapv', '-mllvm', '-amdgpu-dpp-combine=false'), ('-fno-slp-vectorize', '-fwrapv', '-O0')):
    blob = host.Blob.from_source(src, options=opts)
    prog = host.Program(blob, spec)
    res = []
    for md in (0, 1, -1):
      prog.set_max_depth(md)
      got = prog.run_numpy([a], iterate=2)[0]
      res.append(int((got[sl] != want[sl]).sum()))
    print('%-20s %-60s bad(depth2, depth1, stage) = %s of %d' % (name, ' '.join(opts[2:]), res, want[sl].size))
    prog.close(); blob.unload()
