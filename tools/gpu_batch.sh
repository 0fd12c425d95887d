#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out
( time python3 -m pytest tests -m gpu -x -q --durations=8 ) > $out/gputests_r02c.log 2>&1; tail -16 $out/gputests_r02c.log
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
