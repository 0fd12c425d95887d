"""tools/: every script that is kept (not under tools/archive/) at least starts - `--help`
prints its usage and exits 0 without touching a GPU, the reference or the network."""
import glob
import os
import subprocess
import sys

import pytest

from conftest import ROOT

TOOLS = sorted(glob.glob(os.path.join(ROOT, 'tools', '*.py')))
# the reference-plugin check runs under the container's python3.9 with /root/reference
CONTAINER_ONLY = {'check_reference_plugin.py'}


@pytest.mark.parametrize('path', [p for p in TOOLS if os.path.basename(p) not in CONTAINER_ONLY],
                         ids=os.path.basename)
def test_tool_prints_its_usage(path):
  r = subprocess.run([sys.executable, path, '--help'], capture_output=True, text=True,
                     timeout=120)
  assert r.returncode == 0, r.stderr[-1000:]
  assert len(r.stdout.strip()) > 40, r.stdout


def test_archive_is_described():
  """Probes whose question is closed live under tools/archive/ with a line each in its
  README (script, question, where the answer is recorded)."""
  names = sorted(os.path.basename(p) for p in glob.glob(os.path.join(ROOT, 'tools', 'archive',
                                                                     '*.py')))
  text = open(os.path.join(ROOT, 'tools', 'archive', 'README.md')).read()
  assert names and all('`%s`' % n in text for n in names)
  assert len(TOOLS) <= 30
