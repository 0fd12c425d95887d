#!/bin/bash
# Runs on the GPU box: three rocprofv3 --pmc passes (SQ issue / wait / LDS counters)
# of one bench command; tools/sq_counters.py turns them into per-launch averages.
# usage: bash tools/sq_counters.sh <tag> <bench.py arguments...>
set -e
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/sq_$tag
rm -rf $out; mkdir -p $out
pass1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
pass2="SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAIT_ANY SQ_WAIT_INST_ANY"
pass3="GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_IFETCH SQ_INSTS_BRANCH"
n=0
for counters in "$pass1" "$pass2" "$pass3"; do
  n=$((n + 1))
  rocprofv3 --pmc $counters --output-format csv -d $out/pass$n -- \
      python3 bench.py "$@" --steps 1 --warmup 0 --cpu-seconds 0 > $out/pass$n.log 2>&1
done
python3 tools/sq_counters.py $out > $out/summary.json
cat $out/summary.json | head -c 3000
