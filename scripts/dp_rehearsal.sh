#!/bin/bash
# Data-parallel rehearsals on the one-GPU box (gpurun -- bash scripts/dp_rehearsal.sh r05): the exchange path end to end on a one-rank
# RCCL communicator (no wire bytes: what is measured is the exchange's OWN cost -- staging kernels, extra launches, contention) beside
# the plain step in the same box, and two ranks over gloo sharing the GPU.  One JSON line per run -> gpurun_out/dp_<tag>/.
R=${1:-r05}
O=gpurun_out/dp_$R
mkdir -p $O
run() { name=$1; shift; timeout -k 10 400 python bench.py --no-alt --no-cpu-baseline "$@" > $O/$name.json 2> $O/$name.err || echo "$name failed"; }
run plain_bf16            --dtype bf16 --steps 20 --warmup 5
run rccl_1rank_bf16_direct --dtype bf16 --steps 20 --warmup 5 --force-dist --grad-dtype bf16
run plain_bf16_again      --dtype bf16 --steps 20 --warmup 5
run plain_fp32
run rccl_1rank_fp32_ring  --force-dist
run rccl_1rank_fp32_direct --force-dist --allreduce-algo direct
run gloo_2ranks_bf16_c3   --gpus 2 --backend gloo --dtype bf16 --grad-dtype bf16 --steps 5 --warmup 2 --launch-timeout 380
python - <<PY
import json, glob, os
for f in sorted(glob.glob("$O/*.json")):
    try:
        d = json.loads(open(f).readline())
    except Exception as e:
        print(os.path.basename(f), "unreadable", e); continue
    a = d.get("allreduce") or {}
    print("%-28s %8.1f images/s  %7.3f ms/step  exposed %s  overhead %s  without %s" % (os.path.basename(f)[:-5], d["value"], d["ms_per_step"],
          a.get("exposed_ms_per_step"), a.get("overhead_ms_per_step"), a.get("ms_per_step_without_exchange")))
PY
