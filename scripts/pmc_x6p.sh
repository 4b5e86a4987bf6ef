#!/bin/bash
# PMC counters of the plane-image bf16x6 kernel (whole kernel and its DMA stream alone): L2 hits / misses, SQ busy / wait / LDS.
#   gpurun -- bash scripts/pmc_x6p.sh     (separate rocprofv3 passes, kernel-trace + counters only)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc_x6p
rm -rf $O && mkdir -p $O
XP_LAB_MASKS=0,14 timeout -k 10 240 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/a -- python3 scripts/x6p_ablate.py fc7 > $O/a.log 2>&1 || echo "pass a failed"
XP_LAB_MASKS=0,14 timeout -k 10 240 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/b -- python3 scripts/x6p_ablate.py fc7 > $O/b.log 2>&1 || echo "pass b failed"
python3 - <<PY
import csv, glob, collections
for p in "ab":
    f = glob.glob("$O/%s/*/*_counter_collection.csv" % p)
    if not f: print("pass", p, "no counters"); continue
    agg = collections.OrderedDict()
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"]
        if "igemm_x6p" not in k: continue
        k = k[k.index("igemm_x6p"):][:40]
        d = agg.setdefault(k, collections.defaultdict(float))
        d[r["Counter_Name"]] += float(r["Counter_Value"]); d["_n"] += 1
    for k, c in agg.items():
        n = c.pop("_n") / max(1, len(c))
        print(p, k, "launches %d" % n, "  ".join("%s %.4g" % (a, v / n) for a, v in c.items()))
PY
