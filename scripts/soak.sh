#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/soak_r06.txt
: > $O
run() { timeout -k 10 400 python3 bench.py --no-alt --no-cpu-baseline --warmup 5 "$@" 2>/dev/null | grep '"metric"' | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-62s %8.1f %8.2f   %s' % ('$*', d['value'], d['ms_per_step'], d.get('final_loss', d.get('loss'))))" >> $O; }
run --dtype fp32 --steps 400
run --dtype bf16x6 --steps 400
run --preset largefov --steps 400
run --size 513 --classes 41 --batch 10 --steps 100
run --dtype bf16 --steps 400
cat $O
