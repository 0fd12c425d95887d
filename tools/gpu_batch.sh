#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out
python3 -m pytest tests/test_gpu_parity.py -k "multi_process_slabs" -x -q 2>&1 | tail -15
python3 bench.py --force-dist --steps 3 --warmup 1 --cpu-seconds 0 2>&1 | grep "^{" | cut -c1-2300
SODA_DIST_BACKEND=gloo python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29555 bench.py --gpus 2 --steps 2 --warmup 1 --cpu-seconds 0 --size 16384 4096 --iterate 200 2>&1 | grep "^{" | cut -c1-1500
SODA_DIST_BACKEND=gloo python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29556 bench.py --gpus 2 --steps 2 --warmup 1 --cpu-seconds 0 --size 16384 4096 --iterate 200 --overlap 2>&1 | grep "^{" | cut -c1-900
