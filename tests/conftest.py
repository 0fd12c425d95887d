import os
import sys

import pytest

# PyTorch-ROCm ships its own libamdhip64; whichever HIP runtime is loaded first
# owns the process.  Import torch BEFORE libsoda_hip.so is loaded (bench.py and
# the multi-GPU driver do the same) so that both use one runtime; the other
# order leaves torch with "No HIP GPUs are available".
try:
  import torch  # noqa: F401
except ImportError:   # CPU-only environments without torch still run the suite
  torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'soda-compiler_amd')):
  if p not in sys.path:
    sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')
SAMPLES = os.path.join(ROOT, 'tests', 'samples')


def pytest_configure(config):
  config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')


@pytest.fixture(scope='session')
def golden_dir():
  return GOLDEN


@pytest.fixture(scope='session')
def samples_dir():
  return SAMPLES
