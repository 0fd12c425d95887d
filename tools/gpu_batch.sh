#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python3 tools/ab_stage.py 2>&1 | grep "new\|old"
python3 -m pytest tests/test_gpu_parity.py -k "beyond_the_grid or multi_output or stage or one_dim or fixture" -x -q 2>&1 | tail -3
