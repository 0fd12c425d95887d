import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'soda-compiler_amd'), os.path.join(ROOT, 'tests')]
import numpy as np
import gpu_util
from soda_hip.codegen import kernel
from soda_hip.runtime import host
from oracle import soda_oracle
spec = gpu_util.load_spec('sobel2d', iterate=2)
orc = soda_oracle.Oracle(spec)
for small in (True,):
  inputs = gpu_util.random_inputs(spec, (40, 600), small_ints=small)
  want = orc.run(inputs, iterate=2)['mag']
  sl = orc.valid_slices((600, 40), 2)
  text, _ = kernel.generate(spec)
  for opts in (('-fno-slp-vectorize', '-fwrapv'), ('-fno-slp-vectorize', '-fwrapv', '-mllvm', '-amdgpu-dpp-combine=false'), ('-fno-slp-vectorize', '-fwrapv', '-mllvm', '-amdgpu-sdwa-peephole=false'), ('-fno-slp-vectorize', '-fwrapv', '-mllvm', '-amdgpu-dpp-combine=false', '-mllvm', '-amdgpu-sdwa-peephole=false')):
    blob = host.Blob.from_source(text, options=opts)
    prog = host.Program(blob, spec)
    for md in (0, 1, -1):
      prog.set_max_depth(md)
      got = prog.run_numpy(inputs, iterate=2)[0]
      bad = np.argwhere(got[sl] != want[sl])
      print('small', small, opts, 'max_depth', md, 'bad', len(bad), bad[:3].tolist(), (got[sl][tuple(bad[0])], want[sl][tuple(bad[0])]) if len(bad) else '')
    prog.close()
