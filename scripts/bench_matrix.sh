#!/bin/bash
# The bench lines of every BASELINE configuration that fits one GPU, un-profiled (gpurun -- bash scripts/bench_matrix.sh r02):
# configs[1] fp32 (headline) / bf16 / bf16x6, LargeFOV preset, and configs[4] (513x513, 41 classes, 10 images) in both presets
# and all three arithmetics.  One JSON line per run -> gpurun_out/bench_matrix_<round>.jsonl
set -e
R=${1:-r02}
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/bench_matrix_$R.jsonl
: > $O
run() { python3 bench.py --no-alt --steps 8 --warmup 3 --no-cpu-baseline "$@" 2>/dev/null | grep '"metric"' >> $O; }
run
run --dtype bf16
run --dtype bf16x6
run --preset largefov
run --preset largefov --dtype bf16
run --preset largefov --dtype bf16x6
run --size 513 --classes 41 --batch 10
run --size 513 --classes 41 --batch 10 --dtype bf16
run --size 513 --classes 41 --batch 10 --dtype bf16x6
run --size 513 --classes 41 --batch 10 --preset largefov
run --size 513 --classes 41 --batch 10 --preset largefov --dtype bf16
run --size 513 --classes 41 --batch 10 --preset largefov --dtype bf16x6
python3 - <<PY
import json
for l in open("$O"):
    d = json.loads(l)
    print("%-150s %8.1f images/s  %7.2f ms/step  conv %s TF" % (d["config"]["workload"][:150], d["value"], d["ms_per_step"], d["roofline"]["all_conv_tflops"]))
PY
# round 4: configs[4] "as memory allows" -- 513x513 at batch 32 puts conv1's activations past 2 GiB (batch-sliced launches)
python3 bench.py --no-alt --steps 4 --warmup 2 --no-cpu-baseline --size 513 --classes 41 --batch 32 2>/dev/null | grep '"metric"' | tee -a $O | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-150s %8.1f images/s  %7.2f ms/step' % (d['config']['workload'][:150], d['value'], d['ms_per_step']))"
