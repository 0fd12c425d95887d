import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'soda-compiler_amd'), os.path.join(ROOT, 'tests')]
import numpy as np
import gpu_util
from soda_hip.runtime import host
from oracle import soda_oracle
app = sys.argv[1]; shape = tuple(int(v) for v in sys.argv[2].split('x')); it = int(sys.argv[3]); md = int(sys.argv[4])
spec = gpu_util.load_spec(app)
orc = soda_oracle.Oracle(spec)
prog = gpu_util.open_prebuilt(app)
prog.set_max_depth(md)
for kind in ('ramp', 'random'):
  inputs = host.reference_init(spec, list(reversed(shape))) if kind == 'ramp' else gpu_util.random_inputs(spec, shape)
  got = prog.run_numpy(inputs, iterate=it)[0]
  everything = orc.run(inputs, iterate=it, keep_all=True)
  want = everything[spec['outputs'][0]]
  sl = orc.valid_slices(tuple(reversed(shape)), it)
  bad = np.argwhere(got[sl] != want[sl])
  print(kind, 'bad', len(bad), 'of', got[sl].size, bad[:5].tolist())
  for b in bad[:5]:
    print('   got', repr(got[sl][tuple(b)]), 'want', repr(want[sl][tuple(b)]))
