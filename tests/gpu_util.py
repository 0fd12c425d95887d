"""Helpers of the GPU tests: programs are opened through the C ABI from the
code objects that __graft_entry__.build() produced (or JIT-compiled from
freshly generated kernel text when a test wants non-default generator options)."""
import os

import numpy as np

from soda_hip import frontend
from soda_hip.codegen import kernel
from soda_hip.codegen import spec as specmod
from soda_hip.runtime import host

from conftest import ROOT, SAMPLES

SEED = 20240607
BLOBS = os.path.join(ROOT, 'soda-compiler_amd', 'blobs')
# The checker's shared objects of the GPU tests go to tests/_oracle_build (cached,
# git-ignored), not oracle/_build: the driver records which in-tree .so files the
# GPU test processes load, alphabetically and capped - dozens of oracle/_build/*.so
# in front would push soda-compiler_amd/csrc/libsoda_hip.so (the product) off that list.
ORACLE_BUILD = os.path.join(ROOT, 'tests', '_oracle_build')


def make_oracle(spec, **kw):
  from oracle import soda_oracle
  return soda_oracle.Oracle(spec, build_dir=ORACLE_BUILD, **kw)


def sample_path(app):
  path = os.path.join(SAMPLES, app + '.soda')
  return path if os.path.exists(path) else os.path.join(SAMPLES, 'extra', app + '.soda')


def load_spec(app, **overrides):
  st = frontend.load(sample_path(app), **overrides)
  return specmod.spec_from_stencil(st)


def open_prebuilt(app):
  spec = load_spec(app)
  path = os.path.join(BLOBS, app + '.hsaco')
  assert os.path.exists(path), '%s missing: run __graft_entry__.build()' % path
  return host.open_program(blob=path, spec=spec)


def open_jit(app, iterate=None, **gen):
  spec = load_spec(app, iterate=iterate) if iterate else load_spec(app)
  text, _ = kernel.generate(spec, **gen)
  return host.open_program(source=text, spec=spec)


def random_inputs(spec, shape, seed=SEED, small_ints=False):
  rng = np.random.default_rng(seed)
  out = []
  for t in spec['inputs']:
    dt = np.dtype(specmod.NUMPY_NAME[t['c_type']])
    if dt.kind == 'f':
      out.append(rng.random(shape, dtype=np.float32).astype(dt))
    else:
      hi = 256 if small_ints else np.iinfo(dt).max + 1
      out.append(rng.integers(0, hi, size=shape).astype(dt))
  return out
