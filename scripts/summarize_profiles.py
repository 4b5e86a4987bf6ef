#!/usr/bin/env python3
"""Turn gpurun_out/profiles_<round>/ (made by scripts/make_profiles.sh on the GPU box) into the small text/CSV
summaries committed under profiles/.   usage: scripts/summarize_profiles.py r01"""
import collections
import csv
import glob
import os
import re
import shutil
import subprocess
import sys

def newest(pattern):
    """Newest file matching pattern (gpurun merges every run into the same local directory)."""
    return max(glob.glob(pattern), key=os.path.getmtime)


sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # (em_adapt_tensorflow_amd._build.source_hash)
R = sys.argv[1] if len(sys.argv) > 1 else "r01"
SRC = "gpurun_out/profiles_%s" % R
DST = "profiles"
os.makedirs(DST, exist_ok=True)


def short(name):
    m = re.search(r"((?:igemm|wgrad)(?:_ws|2)?_kernel<[^>]*>|conv_d64_kernel<\w+|\w+_kernel\w*)", name)
    return m.group(1) if m else name[:50]


shutil.copy(newest(SRC + "/trace/runc/*_kernel_stats.csv"), "%s/%s_bench_kernel_stats.csv" % (DST, R))
shutil.copy(SRC + "/bench_line.json", "%s/%s_bench_line.json" % (DST, R))
with open("%s/%s_bench_per_layer.txt" % (DST, R), "w") as f:
    f.write(subprocess.check_output([sys.executable, "scripts/trace_layers.py", newest(SRC + "/trace/runc/*_kernel_trace.csv")]).decode())
if os.path.exists(SRC + "/estep_bench.txt"):
    shutil.copy(SRC + "/estep_bench.txt", "%s/%s_estep_bench.txt" % (DST, R))


def load(path):
    by = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        key = int(r["Dispatch_Id"])
        d = by.setdefault(key, {"name": short(r["Kernel_Name"]), "ns": int(r["End_Timestamp"]) - int(r["Start_Timestamp"])})
        d[r["Counter_Name"]] = float(r["Counter_Value"])
    return [by[k] for k in sorted(by)]   # dispatch order (the file is not always in it)


out = ["# rocprofv3 --pmc passes over `bench.py --steps 1 --warmup 1` (kernel-trace + counters only; one run per counter set).",
       "# Aggregated over the LAST step's launches of each conv kernel. FETCH_SIZE/WRITE_SIZE are in KiB as reported;",
       "# gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE under-reports wide coalesced reads by 2x -> hbm_read = 2*FETCH_SIZE*1024.", ""]
agg = collections.OrderedDict()
for sub in ("pmc_sq", "pmc_fetch", "pmc_write"):
    rows = load(newest("%s/%s/runc/*_counter_collection.csv" % (SRC, sub)))
    conv = [r for r in rows if re.search(r"igemm|wgrad|conv_d64|conv_d4|conv_k32", r["name"])]
    conv = conv[len(conv) // 2:]  # second (timed) step
    for r in conv:
        a = agg.setdefault(r["name"], collections.defaultdict(float))
        a["launches_" + sub] += 1
        a["ns_" + sub] += r["ns"]
        for k, v in r.items():
            if k not in ("name", "ns"):
                a[k] += v
for name, a in agg.items():
    n = a["launches_pmc_sq"]
    ms = a["ns_pmc_sq"] / 1e6
    line = "%-44s launches/step %3d  time %7.3f ms" % (name, n, ms)
    if a.get("SQ_VALU_MFMA_BUSY_CYCLES"):
        # busy cycles summed over the 1024 SIMDs; kernel cycles at the measured 2.4 GHz
        line += "  mfma_pipe_busy %.1f%%" % (100 * a["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / (a["ns_pmc_sq"] * 2.4))
        wc = a["SQ_WAVE_CYCLES"]
        line += "  wave: wait_any %.1f%% wait_inst %.1f%% active %.1f%%  lds_bank_conflict %d" % (
            100 * a["SQ_WAIT_ANY"] / wc, 100 * a["SQ_WAIT_INST_ANY"] / wc, 100 * a["SQ_ACTIVE_INST_ANY"] / wc, a["SQ_LDS_BANK_CONFLICT"])
    out.append(line)
    if a.get("FETCH_SIZE") is not None:
        rd = 2 * a["FETCH_SIZE"] * 1024
        wr = a.get("WRITE_SIZE", 0.0) * 1024
        hit = a["TCC_HIT_sum"] / max(a["TCC_HIT_sum"] + a["TCC_MISS_sum"], 1)
        out.append("    HBM traffic/step: read %.2f GB (2 x FETCH_SIZE), write %.2f GB; per launch %.3f GB; L2 hit rate %.1f%%" % (
            rd / 1e9, wr / 1e9, (rd + wr) / 1e9 / max(n, 1), 100 * hit))
open("%s/%s_pmc_conv_kernels.txt" % (DST, R), "w").write("\n".join(out) + "\n")

# The E-step's kernels in the same passes (north_star: "rocprof counters report achieved HBM GB/s on the E-step scan"): per call of
# emadapt_estep (= its launches: the batch-maximum pre-kernel, the scan, and for geometries that overflow the LDS prep / finish)
eout = ["# E-step kernels in the same rocprofv3 --pmc passes (last step): HBM-side bytes = 2 x FETCH_SIZE + WRITE_SIZE (KiB as reported;",
        "# gfx950 correction as above), L2 hit rate, wave-cycle split; algorithmic bytes (SURVEY 8d) = logits once + int64 labels.", ""]
eagg = collections.OrderedDict()
for sub in ("pmc_sq", "pmc_fetch", "pmc_write"):
    rows = load(newest("%s/%s/runc/*_counter_collection.csv" % (SRC, sub)))
    es = [r for r in rows if re.search(r"estep", r["name"])]
    per_step = max(len(es) // 2, 1)
    for r in es[len(es) - per_step:]:
        a = eagg.setdefault(r["name"], collections.defaultdict(float))
        a["launches_" + sub] += 1
        a["ns_" + sub] += r["ns"]
        for k, v in r.items():
            if k not in ("name", "ns"):
                a[k] += v
estep_traffic = {"read_bytes": 0.0, "write_bytes": 0.0, "hit": 0.0, "miss": 0.0, "ns": 0.0}
for name, a in eagg.items():
    rd, wr = 2 * a.get("FETCH_SIZE", 0.0) * 1024, a.get("WRITE_SIZE", 0.0) * 1024
    hit = a["TCC_HIT_sum"] / max(a["TCC_HIT_sum"] + a["TCC_MISS_sum"], 1)
    wc = max(a["SQ_WAVE_CYCLES"], 1)
    eout.append("%-28s launches %d  time %7.1f us  HBM read %8.1f KB  write %7.1f KB  (%.1f GB/s)  L2 hit %.1f%%  wave: wait_any %.1f%% wait_inst %.1f%% active %.1f%%  lds_bank_conflict %d" % (
        name, a["launches_pmc_sq"], a["ns_pmc_sq"] / 1e3, rd / 1e3, wr / 1e3, (rd + wr) / max(a["ns_pmc_fetch"], 1), 100 * hit,
        100 * a["SQ_WAIT_ANY"] / wc, 100 * a["SQ_WAIT_INST_ANY"] / wc, 100 * a["SQ_ACTIVE_INST_ANY"] / wc, a["SQ_LDS_BANK_CONFLICT"]))
    estep_traffic["read_bytes"] += rd
    estep_traffic["write_bytes"] += wr
    estep_traffic["hit"] += a["TCC_HIT_sum"]
    estep_traffic["miss"] += a["TCC_MISS_sum"]
    estep_traffic["ns"] += a["ns_pmc_sq"]
if eagg:
    tot = estep_traffic["read_bytes"] + estep_traffic["write_bytes"]
    eout.append("TOTAL per emadapt_estep call: HBM-side %.1f KB (read %.1f + write %.1f) in %.1f us = %.1f GB/s of 8000 (%.2f %%)" % (
        tot / 1e3, estep_traffic["read_bytes"] / 1e3, estep_traffic["write_bytes"] / 1e3, estep_traffic["ns"] / 1e3,
        tot / max(estep_traffic["ns"], 1), 100 * tot / max(estep_traffic["ns"], 1) / 8000))
    open("%s/%s_pmc_estep_kernels.txt" % (DST, R), "w").write("\n".join(eout) + "\n")
    print("\n".join(eout))

# HBM bytes per step and kernel class (bench.py's roofline.traffic).  The class of a launch comes from its position in the
# step (scripts/kernel_classes.py: forward launches precede the first filter-gradient launch), cross-checked against the BT
# template argument of the fp32 kernels read by position.
import json

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_classes import classes_by_order, conv_kernel  # noqa: E402


def full_rows(path):
    by = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        d = by.setdefault(int(r["Dispatch_Id"]), {"full": r["Kernel_Name"], "name": short(r["Kernel_Name"])})
        d[r["Counter_Name"]] = float(r["Counter_Value"])
    return [by[k] for k in sorted(by)]   # dispatch order = launch order


traffic = {c: {"read_bytes": 0.0, "write_bytes": 0.0, "launches": 0} for c in ("igemm_fwd", "igemm_dgrad", "igemm_wgrad")}
per_kernel = collections.OrderedDict()
for sub, counter, key, mult in (("pmc_fetch", "FETCH_SIZE", "read_bytes", 2 * 1024.0), ("pmc_write", "WRITE_SIZE", "write_bytes", 1024.0)):
    rows = [r for r in full_rows(newest("%s/%s/runc/*_counter_collection.csv" % (SRC, sub))) if conv_kernel(r["full"]) is not None]
    assert len(rows) % 2 == 0, "expected two identical steps (--warmup 1 --steps 1)"
    conv = rows[len(rows) // 2:]   # the second (timed) step
    for r, c in zip(conv, classes_by_order([r["full"] for r in conv])):
        traffic[c][key] += r.get(counter, 0.0) * mult
        traffic[c]["launches"] += key == "read_bytes"
        k = per_kernel.setdefault(r["name"], {"read_bytes": 0.0, "write_bytes": 0.0, "by_class": {}})
        k[key] += r.get(counter, 0.0) * mult
        kc = k["by_class"].setdefault(c, {"read_bytes": 0.0, "write_bytes": 0.0})
        kc[key] += r.get(counter, 0.0) * mult
line = json.load(open(SRC + "/bench_line.json"))
json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of scripts/make_profiles.sh over bench.py --steps 1 --warmup 1; "
                     "read = 2 x FETCH_SIZE KiB (gfx950 correction, MI355X_MICROARCH.md), write = WRITE_SIZE KiB; "
                     "L2-miss (fabric) traffic, Infinity-Cache hits included; class of a launch = its position in the step "
                     "(scripts/kernel_classes.py)",
           "dtype": line["dtype"], "workload": line["config"]["workload"], "key": line["config"].get("workload_key"),
           "source_hash": __import__("importlib").import_module("em_adapt_tensorflow_amd._build").source_hash(),
           "estep": ({"read_bytes": estep_traffic["read_bytes"], "write_bytes": estep_traffic["write_bytes"],
                      "l2_hit_rate": round(estep_traffic["hit"] / max(estep_traffic["hit"] + estep_traffic["miss"], 1), 4)} if eagg else None),
           "per_step": traffic, "per_kernel": per_kernel},
          open("%s/%s_traffic.json" % (DST, R), "w"), indent=1)
print("\n".join(out))
for c, t in traffic.items():
    print("%-12s %3d launches  read %.2f GB  write %.2f GB  per launch %.3f GB" % (
        c, t["launches"], t["read_bytes"] / 1e9, t["write_bytes"] / 1e9, (t["read_bytes"] + t["write_bytes"]) / 1e9 / max(t["launches"], 1)))
