#!/bin/bash
# Sample the GPU's shader clock / power while a workload runs (diagnostic).  usage: clock_probe.sh <seconds> -- cmd...
secs=$1; shift; shift
"$@" > gpurun_out/clock_probe_cmd.log 2>&1 &
pid=$!
sleep 6
for i in $(seq 1 $secs); do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|mclk" | tr '\n' ' ' ; echo
  sleep 0.5
done
wait $pid
tail -1 gpurun_out/clock_probe_cmd.log | cut -c1-200
