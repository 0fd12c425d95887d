#!/bin/bash
# Runs on the GPU box (gpurun): the rocprofv3 passes that tools/collect_profiles.py
# turns into profiles/<tag>_*.  usage: bash tools/collect_on_gpu.sh r01
set -e
tag=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out
rm -rf $out/prof_$tag $out/pmc_fetch* $out/pmc_write*
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$tag -- \
    python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0 > $out/prof_${tag}_bench.log 2>&1
grep "^{" $out/prof_${tag}_bench.log | cut -c1-400
for c in FETCH_SIZE:fetch WRITE_SIZE:write; do
  rocprofv3 --pmc ${c%%:*} --output-format csv -d $out/pmc_${c##*:} -- \
      python3 bench.py --steps 1 --warmup 0 --cpu-seconds 0 > $out/pmc_${c##*:}.log 2>&1
  echo "pmc ${c%%:*} jacobi2d done"
  rocprofv3 --pmc ${c%%:*} --output-format csv -d $out/pmc_${c##*:}_seidel2d -- \
      python3 bench.py --app seidel2d --iterate 100 --steps 1 --warmup 0 --cpu-seconds 0 > $out/pmc_${c##*:}_seidel2d.log 2>&1
  rocprofv3 --pmc ${c%%:*} --output-format csv -d $out/pmc_${c##*:}_blur -- \
      python3 bench.py --app blur --iterate 1 --steps 1 --warmup 0 --cpu-seconds 0 > $out/pmc_${c##*:}_blur.log 2>&1
  rocprofv3 --pmc ${c%%:*} --output-format csv -d $out/pmc_${c##*:}_jacobi3d -- \
      python3 bench.py --app jacobi3d --size 512 512 512 --iterate 200 --steps 1 --warmup 0 --cpu-seconds 0 > $out/pmc_${c##*:}_jacobi3d.log 2>&1
  echo "pmc ${c%%:*} others done"
done
