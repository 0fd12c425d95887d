#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out
( time python3 -m pytest tests -m gpu -x -q --durations=25 ) > $out/gputests_r02a.log 2>&1; tail -40 $out/gputests_r02a.log
