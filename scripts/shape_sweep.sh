#!/bin/bash
# Smoke over batch sizes / input sizes / presets the launch plans are not tuned for: every line must end with a finite loss.
# usage (GPU box): bash scripts/shape_sweep.sh
cd "$GRAFT_REPO_ROOT"
run() { echo -n "$* -> "; timeout -k 10 200 python3 bench.py --no-alt --steps 2 --warmup 1 --no-cpu-baseline --no-prof "$@" 2>&1 | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('%.1f images/s, loss %.4f' % (d['value'], d['final_loss'])); break
else:
    print('FAILED')
"; }
run --batch 1
run --batch 3
run --batch 7
run --batch 32
run --batch 1 --size 513 --classes 41 --preset largefov
run --batch 5 --size 257 --classes 21
run --batch 16 --width 32
run --batch 2 --size 129 --width 48 --classes 5
run --batch 3 --dtype bf16
run --batch 32 --dtype bf16
run --batch 1 --dtype bf16x6
run --batch 3 --dtype bf16x6
run --batch 7 --dtype bf16x6
run --batch 32 --dtype bf16x6
run --batch 5 --size 257 --classes 21 --dtype bf16x6
run --batch 16 --width 32 --dtype bf16x6
run --batch 2 --size 129 --width 48 --classes 5 --dtype bf16x6
run --batch 1 --size 513 --classes 41 --preset largefov --dtype bf16x6
