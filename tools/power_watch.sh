#!/bin/bash
# Sample power / clocks while a command runs: tools/power_watch.sh out.log -- cmd...
out=$1; shift; shift
"$@" &
pid=$!
while kill -0 $pid 2>/dev/null; do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk|mclk|fclk" >> "$out"
  echo "--" >> "$out"
  sleep 1
done
wait $pid
