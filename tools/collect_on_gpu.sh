#!/bin/bash
# Runs on the GPU box (gpurun): the rocprofv3 passes that tools/collect_profiles.py
# turns into profiles/<tag>_*.  usage: bash tools/collect_on_gpu.sh r02
# One --kernel-trace --stats pass of the default bench, then per workload one
# --pmc FETCH_SIZE and one --pmc WRITE_SIZE pass (separate runs: the TCC block
# cannot count both at once, MI355X_MICROARCH.md, rocprofv3 PMC slots).
set -e
tag=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out
rm -rf $out/prof_$tag $out/pmc_${tag}_*
# which tree the kernels were built from (written by tools/collect.sh before the call)
cp tools/.collect_commit $out/pmc_${tag}_commit.txt 2>/dev/null || echo unknown > $out/pmc_${tag}_commit.txt
# The --stats pass repeats the schedule a PLAIN run settles on (bench.py --split): with
# the tuning step under the profiler the summary's AverageNs mixed candidate splits into
# every kernel's average (r03: 574 calls of the depth-24 kernel for 40 per sweep).  With
# the split given: 1 warm-up + 3 timed + 3 event-bracketed sweeps = 7 x launches_per_step
# calls, and sum(launches_i x AverageNs_i) is comparable with the line's ms_per_step.
plain=$(python3 bench.py --steps 10 --warmup 5 --cpu-seconds 0 2>/dev/null | grep "^{" | tail -1)
echo "$plain" > $out/prof_${tag}_plain_bench.json
hsplit=$(echo "$plain" | python3 -c "import sys, json; print(json.loads(sys.stdin.read())['config']['depth_schedule'])")
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$tag -- \
    python3 bench.py --split "$hsplit" --steps 3 --warmup 1 --cpu-seconds 0 > $out/prof_${tag}_bench.log 2>&1
grep "^{" $out/prof_${tag}_bench.log | cut -c1-300
# name|bench arguments   (the workload table tools/collect_profiles.py reads back)
cat > $out/pmc_${tag}_workloads.txt <<'WL'
cfg4|--app jacobi2d --size 16384 16384 --iterate 1000
cfg2|--app jacobi2d --size 8192 8192 --iterate 100
cfg3|--app blur --size 16384 16384 --iterate 1
cfg5|--app jacobi3d --size 512 512 512 --iterate 200
seidel2d|--app seidel2d --size 16384 16384 --iterate 100
sobel2d|--app sobel2d --size 16384 16384 --iterate 1
heat3d|--app heat3d --size 512 512 512 --iterate 20
denoise2d|--app denoise2d --size 8192 8192 --iterate 1
denoise3d|--app denoise3d --size 256 256 256 --iterate 1
cfg2_5x20|--app jacobi2d --size 8192 8192 --iterate 100|5x20
cfg2_mixed|--app jacobi2d --size 8192 8192 --iterate 100|2x24+2x20+1x12
cfg2_4x24|--app jacobi2d --size 8192 8192 --iterate 100|4x24+1x4
WL
# (a third field = a GIVEN split: the tuning step settles on different splits of cfg2 from
# box to box - 5x20 here, 2x24+2x20+1x12 on the driver's box in round 5 - and a traffic
# entry speaks only for the schedule it was measured on (bench.py: profile_entry_matches),
# so each of them gets its own passes)
# The split of `iterate` a plain run settles on (soda_hip_plan_tune) is found once per
# workload, without the profiler, and GIVEN to every counter pass (--split): under
# --pmc the tuning step times its candidates with the counters' overhead on top and
# settled on other splits (cfg2: 3x24+16+12 instead of 5x20), so a pass measured
# launches the timed run does not have.
: > $out/pmc_${tag}_splits.txt
while IFS='|' read -r name wargs given; do
  if [ -n "$given" ]; then
    split=$given
  else
    split=$(python3 bench.py $wargs --no-other-configs --steps 10 --warmup 5 --cpu-seconds 0 2>/dev/null | \
            python3 -c "import sys, json; print(json.loads([l for l in sys.stdin if l.startswith('{')][-1])['config']['depth_schedule'])")
  fi
  case "$split" in *x[0-9]*) : ;; *) split="" ;; esac     # per-stage schedules: nothing to fix
  echo "$name|$split" >> $out/pmc_${tag}_splits.txt
done < $out/pmc_${tag}_workloads.txt
split_of() { grep "^$1|" $out/pmc_${tag}_splits.txt | cut -d'|' -f2; }
while IFS='|' read -r name wargs given; do
  sp=$(split_of $name)
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $out/pmc_${tag}_${name}_$c -- \
        python3 bench.py $wargs ${sp:+--split $sp} --steps 1 --warmup 0 --cpu-seconds 0 \
        > $out/pmc_${tag}_${name}_$c.log 2>&1
  done
  echo "pmc $name done ($sp)"
done < $out/pmc_${tag}_workloads.txt
# SQ counters (VALU issue utilisation, where a wavefront spends its life): three more
# --pmc passes per workload (tools/sq_counters.sh), summarised per kernel
while IFS='|' read -r name wargs given; do
  [ -n "$given" ] && continue          # (the SQ counters of cfg2 come from its own entry)
  sp=$(split_of $name)
  bash tools/sq_counters.sh ${tag}_${name} $wargs ${sp:+--split $sp} > /dev/null 2>&1 || echo "sq $name FAILED"
  echo "sq $name done"
done < $out/pmc_${tag}_workloads.txt

# what soda_hip_plan_tune's streaming step finds on this box (profiles/rNN_stream_chunk.txt)
bash tools/stream_tune_probe.sh $out/${tag}_stream_tune_collect.log || true
