#!/bin/bash
# Extra SQ counters per kernel (one rocprofv3 pass each set; kernel-trace + counters only): where the MFMA waves of the
# 256x256 forward / data-gradient kernel wait compared with the filter-gradient kernel built the same way.
#   gpurun -- bash scripts/pmc_compare.sh
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc_cmp
rm -rf $O && mkdir -p $O
pass() { n=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/$n -- python3 bench.py --no-alt --steps 1 --warmup 1 --no-cpu-baseline --no-prof > $O/$n.log 2>&1; }
pass a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT
pass b SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU
pass c SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_VMEM
python3 - <<PY
import csv, glob, collections, re
def short(n):
    m = re.search(r"((?:igemm|wgrad)(?:_ws|2)?_kernel<[^>]*>)", n)
    return m.group(1) if m else None
for p in "abc":
    f = glob.glob("$O/%s/*/*_counter_collection.csv" % p)
    if not f:
        print("pass", p, "produced no counters (see $O/%s.log)" % p); continue
    agg = collections.OrderedDict()
    for r in csv.DictReader(open(f[0])):
        k = short(r["Kernel_Name"])
        if not k: continue
        agg.setdefault(k, collections.defaultdict(float))[r["Counter_Name"]] += float(r["Counter_Value"])
    print("== pass", p)
    for k, c in agg.items():
        wc = c.get("SQ_WAVE_CYCLES", 0) or 1
        print("%-58s" % k, "  ".join("%s %.3g (%.1f%% of wave cycles)" % (n.replace("SQ_", ""), v, 100 * v / wc) for n, v in c.items() if n != "SQ_WAVE_CYCLES"))
PY
