import os
import sys

import pytest


def _cpu_quota():
  """CPUs this process may really use: the cgroup quota where there is one (a GPU box
  shows 256 logical CPUs and grants 16), else the affinity mask."""
  n = len(os.sched_getaffinity(0))
  try:
    with open('/sys/fs/cgroup/cpu.max') as f:
      quota, period = f.read().split()
    if quota != 'max':
      n = min(n, max(1, int(round(int(quota) / int(period)))))
  except (OSError, ValueError):
    pass
  return n


# The oracle's loops are OpenMP: a team as large as the machine on a fraction of its CPUs
# spends its time waiting for descheduled threads at every loop's end.  Set before any
# OpenMP runtime is loaded (torch brings one).
os.environ.setdefault('OMP_NUM_THREADS', str(_cpu_quota()))

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
