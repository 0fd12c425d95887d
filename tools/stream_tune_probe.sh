#!/bin/bash
# On the GPU box: what soda_hip_plan_tune's streaming step finds on THIS box for the memory-bound
# one-launch sweeps (cfg3 blur, sobel2d, a depth-1 jacobi2d sweep; 16384^2), and the sweep time under
# the bench protocol with the calibrated chunk (--no-tune) and with the tuned one.  Appends to $1.
out=${1:-gpurun_out/stream_tune.log}
echo "# box $(hostname) $(date -u +%H:%M)" >> $out
for app in blur sobel2d jacobi2d; do
  for mode in --no-tune ""; do
    SODA_HIP_TUNING=1 SODA_HIP_DEBUG=1 python3 bench.py --app $app --size 16384 16384 --iterate 1 --steps 30 --warmup 10 \
        --cpu-seconds 0 $mode 2> /tmp/stream_tune.err | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('%-9s %-9s %.4f ms  %6.1f G/s  %s' % ('$app', '${mode:-tuned}', d['ms_per_step'], d['value'], d['roofline']['kernel']))" >> $out
    grep "tune stream" /tmp/stream_tune.err | sed 's/^soda_hip: /    /' >> $out
  done
done
