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
sys.exit(0 if ok else 1)
