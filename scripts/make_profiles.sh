#!/bin/bash
# Regenerate the round's profile artefacts on the GPU box (run through gpurun from the repo root):
#   gpurun -- bash scripts/make_profiles.sh r02 [extra bench.py flags, e.g. --dtype bf16]
# Trace/stats and PMC are separate rocprofv3 runs (never --pmc together with tracing domains other than kernel-trace).
set -e
R=${1:-r02}
shift || true
X="$@"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/profiles_$R
rm -rf $O && mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --no-alt --steps 10 --warmup 3 $X > $O/bench.log 2>&1
grep '"metric"' $O/bench.log > $O/bench_line.json
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/pmc_sq -- python3 bench.py --no-alt --steps 1 --warmup 1 --no-cpu-baseline --no-prof $X > $O/pmc_sq.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 bench.py --no-alt --steps 1 --warmup 1 --no-cpu-baseline --no-prof $X > $O/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_write -- python3 bench.py --no-alt --steps 1 --warmup 1 --no-cpu-baseline --no-prof $X > $O/pmc_write.log 2>&1
echo done; ls $O
