#!/bin/bash
# sample shader/memory clock and power while a perf_probe configuration loops (diagnostic)
# usage: tools/clock_watch.sh "<perf_probe args>"
python tools/perf_probe.py $1 --iters 120 > /tmp/cw_probe.log 2>&1 &
PID=$!
sleep 2.0
for i in 1 2 3 4; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Socket Power|Average Graphics" | tr -s ' ' | tr '\n' ';'
  echo
  sleep 0.4
done
wait $PID
grep -v amdgpu /tmp/cw_probe.log | tail -1
