#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -k "3d or slab or denormals_in_the_deep or random_vs_oracle or full_size_cfg5" -x -q 2>&1 | tail -12
