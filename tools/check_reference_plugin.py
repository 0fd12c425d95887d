#!/opt/conda/bin/python3.9
"""Container-only check (needs /root/reference + python3.9): the HIP back end is
a drop-in plug-in of the REFERENCE driver.  Builds the reference's own
`soda.core.Stencil` for each sample, hands it to
`soda_hip.codegen.backend.print_code(stencil, args)` exactly the way
reference src/sodac:127 calls `xocl.print_code`, and checks that the kernel text
equals what this project's own front end produces for the same program."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'soda-compiler_amd'))
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))

import make_golden                      # brings the reference onto sys.path
from soda_hip.codegen import backend    # noqa: E402
from soda_hip.codegen import spec as specmod  # noqa: E402
from soda_hip import frontend           # noqa: E402

parser = argparse.ArgumentParser()
parser.add_argument('--write', action='store_true')
backend.add_arguments(parser.add_argument_group('HIP backend'))
ok = True
for app in ('blur', 'jacobi2d', 'jacobi3d', 'seidel2d', 'heat3d', 'sobel2d',
            'denoise2d', 'denoise3d'):
  ref_stencil = make_golden.build_stencil(
      os.path.join(make_golden.REF, 'tests/src/%s.soda' % app))
  ref_spec = backend.to_spec(ref_stencil)
  own_spec = specmod.spec_from_stencil(
      frontend.load(os.path.join(ROOT, 'tests', 'samples', app + '.soda')))
  same = specmod.dumps(ref_spec) == specmod.dumps(own_spec)
  out = os.path.join('/tmp', 'plugin_%s_kernel.hip' % app)
  args = parser.parse_args(['--hip-kernel', out])
  backend.print_code(ref_stencil, args)
  print('%-10s spec from reference Stencil == spec from own front end: %s; '
        'kernel text %d bytes' % (app, same, os.path.getsize(out)))
  ok &= same

# ... and for every other program the tests know: the hand-written extras and the
# random programs.  `--write` stores the sha256 of each spec built FROM THE
# REFERENCE'S Stencil in tests/golden/plugin_specs.json; tests/test_frontend.py
# checks the own front end's specs against that file (no reference needed there).
import hashlib   # noqa: E402
import json      # noqa: E402
digests = {}
programs = []
extra = os.path.join(ROOT, 'tests', 'samples', 'extra')
for fname in sorted(os.listdir(extra)):
  programs.append(('extra.' + fname[:-5], os.path.join(extra, fname), None))
with open(os.path.join(ROOT, 'tests', 'golden', 'random_programs.json')) as f:
  for key, entry in sorted(json.load(f).items()):
    programs.append(('random.' + key, None, entry['text']))
mismatches = 0
for key, path, text in programs:
  ref_stencil = make_golden.build_stencil(path, text=text)
  own = frontend.load(path) if path else frontend.loads(text)
  try:
    ref_spec = specmod.dumps(backend.to_spec(ref_stencil))
  except Exception as e:      # both sides must refuse the same programs
    try:
      specmod.spec_from_stencil(own)
      print(key, 'only the plug-in path raises:', e)
      mismatches += 1
    except Exception:
      pass
    continue
  own_spec = specmod.dumps(specmod.spec_from_stencil(own))
  if ref_spec != own_spec:
    print(key, 'spec from reference Stencil != spec from own front end')
    mismatches += 1
  digests[key] = hashlib.sha256(ref_spec.encode()).hexdigest()
print('%d extra / random programs, %d mismatches' % (len(programs), mismatches))
if '--write' in sys.argv:
  with open(os.path.join(ROOT, 'tests', 'golden', 'plugin_specs.json'), 'w') as f:
    json.dump(digests, f, indent=0, sort_keys=True)
sys.exit(0 if ok and not mismatches else 1)
