#!/bin/bash
# Kernel trace of one bench.py workload (no PMC passes): gpurun -- bash scripts/trace_only.sh <tag> [bench.py flags]
# -> gpurun_out/trace_<tag>/{bench_line.json, runc/*_kernel_trace.csv}; scripts/trace_layers.py turns the trace into the per-layer table.
set -e
R=$1
shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/trace_$R
rm -rf $O && mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 bench.py --no-alt --no-cpu-baseline --steps 6 --warmup 2 "$@" > $O/bench.log 2>&1
grep '"metric"' $O/bench.log > $O/bench_line.json
ls $O/*/ | head
