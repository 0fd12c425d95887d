#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out
for n in 256 512; do
TUNE_HIPCC=1 python3 tools/tune3d.py denoise3d $n 1 'rows=16,cols=1' 'rows=12,cols=1' 'rows=16,cols=1,waves_per_eu=3' 'rows=8,cols=2' 'rows=8,cols=1' 'fused=0' 2>&1 | grep -v amdgpu.ids
done
