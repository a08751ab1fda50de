#!/bin/bash
# usage (GPU box): tools/power_sample.sh <seconds> <command...>: runs the command in the background and samples socket
# power / clocks / throttle state while it runs (rocm-smi every ~0.5 s; one amd-smi metric dump in the middle)
secs=$1; shift
"$@" > /tmp/ps_cmd.log 2>&1 &
pid=$!
for i in $(seq 1 $((2 * secs))); do
  kill -0 $pid 2>/dev/null || break
  rocm-smi -P -g 2>/dev/null | grep -iE "power|sclk" | sed -e 's/=//g' | tr '\n' ' '; echo
  if [ $i -eq 8 ]; then amd-smi metric -g 0 2>&1 | grep -iE "power|clk|throttl|violation|ppt|limit|temp|hot" | head -60; fi
  sleep 0.5
done
wait $pid
cat /tmp/ps_cmd.log
