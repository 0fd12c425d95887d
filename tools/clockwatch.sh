#!/bin/bash
# Polls shader clock / power while a command runs on the GPU box.
# usage: clockwatch.sh <command ...>
"$@" > gpurun_out/clockwatch_cmd.log 2>&1 &
pid=$!
sleep ${CLOCKWATCH_DELAY:-25}
for i in $(seq 1 ${CLOCKWATCH_SAMPLES:-12}); do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power" | tr '\n' ' '
  echo
  sleep 0.5
done
wait $pid
tail -3 gpurun_out/clockwatch_cmd.log
