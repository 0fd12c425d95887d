#!/bin/bash
# In the build container: collects the round's rocprofv3 evidence at a CLEAN commit.
#   bash tools/collect.sh r04
# Refuses a dirty tree (profiles/ and gpurun_out/ aside), stamps the commit into the
# snapshot (the GPU box has no .git), runs tools/collect_on_gpu.sh there, then
# tools/collect_profiles.py here.
set -e
tag=${1:?usage: collect.sh <tag>}
cd "$(dirname "$0")/.."
if [ -n "$(git status --porcelain -- . ':!profiles' ':!gpurun_out')" ]; then
  echo "collect.sh: the tree is dirty - commit first (the profiles name the commit they" >&2
  echo "were measured on, and tests/test_profiles.py refuses '+uncommitted')" >&2
  git status --short -- . ':!profiles' ':!gpurun_out' | head >&2
  exit 1
fi
git rev-parse --short HEAD > tools/.collect_commit
/usr/local/graft/bin/gpurun --timeout ${COLLECT_TIMEOUT:-1200} -- "bash tools/collect_on_gpu.sh $tag"
python3 tools/collect_profiles.py "$tag"
