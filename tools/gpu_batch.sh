#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export SODA_HIP_TUNING=1 TUNE_HIPCC=1 TUNE_WARMUP=20 TUNE_REPEATS=40
for chunk in 0 512 1024 2048; do
  if [ $chunk = 0 ]; then unset SODA_HIP_CHUNK_ROWS; else export SODA_HIP_CHUNK_ROWS=$chunk; fi
  echo "== chunk $chunk"
  python3 tools/tune.py jacobi2d 16384 1 '1,4,256,3' '1,4,256,8' '1,4,256,12' '1,4,256,12,nontemporal=1' '1,4,256,16,nontemporal=1' 2>&1 | grep -v amdgpu | cut -c1-120
done
