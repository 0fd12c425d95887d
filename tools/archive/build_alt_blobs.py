#!/usr/bin/env python3
"""Builds alternative code objects of one sample program (generator options given as
'name:key=value,key=value' ...) under tools/alt_<app>_<name>.hsaco, priced with every
calibration entry the repository's history holds - for bench-protocol A/B runs that swap
the blob between `bench.py` calls on the GPU box (profiles/r03_blk_variants.txt)."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'soda-compiler_amd')]
app = sys.argv[1]
p = os.path.join(ROOT, 'soda-compiler_amd', 'soda_hip', 'codegen', 'calibration.json')
keep = open(p).read()
merged = json.loads(keep)
rel = os.path.relpath(p, ROOT)
for c in subprocess.check_output(['git', 'log', '--format=%h', '-60', '--', rel], cwd=ROOT,
                                 text=True).split():
  try:
    old = json.loads(subprocess.check_output(['git', 'show', c + ':' + rel], cwd=ROOT))
  except (subprocess.CalledProcessError, ValueError):
    continue
  for k, v in old.get('kernels', {}).items():
    merged['kernels'].setdefault(k, v)
try:
  with open(p, 'w') as f:
    json.dump(merged, f)
  from soda_hip import frontend
  from soda_hip.codegen import kernel, spec as specmod
  import __graft_entry__ as e
  st = frontend.load(e.sample_path(app), iterate=e.BLOB_ITERATE.get(app))
  spec = specmod.spec_from_stencil(st)
  for arg in sys.argv[2:]:
    name, _, text = arg.partition(':')
    opts = {}
    flags = []
    for kv in text.split(','):
      if not kv:
        continue
      k, v = kv.split('=', 1)
      if k == 'flags':
        flags = v.split(':')
      else:
        opts[k] = int(v) if v.lstrip('-').isdigit() else v
    src, table = kernel.generate(spec, **opts)
    kernel.compile_to_code_object(src, os.path.join(ROOT, 'tools', 'alt_%s_%s.hsaco' % (app, name)),
                                  extra_flags=flags)
    print(name, [(k['name'].split('_k')[-1], k.get('step_ns_full'), k.get('est_vgprs'))
                 for k in table if k['kind'] == 'fused' and k['depth'] >= 16])
finally:
  with open(p, 'w') as f:
    f.write(keep)
