#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/soak_long.txt
: > $O
run() { timeout -k 10 600 python3 bench.py --no-alt --no-cpu-baseline --warmup 5 "$@" 2>/dev/null | grep '"metric"' | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-62s %8.1f %8.2f   %s' % ('$*', d['value'], d['ms_per_step'], d.get('final_loss')))" >> $O; echo "done $*"; }
run --dtype fp32 --steps 2000
run --preset largefov --steps 2500
run --dtype bf16x6 --steps 1500
cat $O
