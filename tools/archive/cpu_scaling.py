import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == 'child':
  sys.path[:0] = [ROOT, os.path.join(ROOT, 'soda-compiler_amd')]
  import numpy as np
  from soda_hip import frontend
  from soda_hip.codegen import spec as S
  from oracle import soda_oracle
  sp = S.spec_from_stencil(frontend.load(os.path.join(ROOT, 'tests/samples/jacobi2d.soda')))
  o = soda_oracle.Oracle(sp, flags=('-O3', '-march=native'))
  a = np.random.default_rng(1).random((16384, 16384), dtype=np.float32)
  t, u = o.time_iterations([a], 20, warmup=3)
  print(json.dumps(dict(threads=os.environ.get('OMP_NUM_THREADS'), bind=os.environ.get('OMP_PROC_BIND'), gcell=u / t / 1e9, gbps=u * 8 / t / 1e9)))
else:
  print(open('/sys/fs/cgroup/cpu.max').read().strip() if os.path.exists('/sys/fs/cgroup/cpu.max') else 'no cpu.max')
  print('affinity', len(os.sched_getaffinity(0)))
  for threads, bind in [(8, ''), (32, ''), (64, ''), (128, ''), (256, ''), (64, 'spread'), (128, 'spread'), (256, 'close')]:
    env = dict(os.environ, OMP_NUM_THREADS=str(threads))
    if bind:
      env.update(OMP_PROC_BIND=bind, OMP_PLACES='cores')
    r = subprocess.run([sys.executable, __file__, 'child'], env=env, capture_output=True, text=True)
    print(r.stdout.strip() or r.stderr[-300:])
