#!/bin/bash
# Samples rocm-smi power / clocks while a workload loops:  bash tools/power_probe.sh <label> <command...>
label=$1; shift
"$@" > gpurun_out/power_$label.log 2>&1 &
pid=$!
sleep 4
for i in 1 2 3 4 5; do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -i "power\|sclk\|fclk\|mclk" | tr '\n' ';'
  echo
  sleep 1
done
wait $pid
tail -1 gpurun_out/power_$label.log
