#!/usr/bin/env python3
"""Sum a rocprofv3 --kernel-trace CSV by kernel name.  usage: kernel_times.py <dir or csv> [min_us]"""
import collections, csv, glob, os, sys
path = sys.argv[1]
f = path if path.endswith(".csv") else glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True)[0]
minus = float(sys.argv[2]) if len(sys.argv) > 2 else 50.0
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    d[r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:100]].append(
        (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    if sum(v) > minus:
        print("%10.1f us total  avg %9.1f us  min %9.1f us  x%4d  %s" % (sum(v), sum(v) / len(v), min(v), len(v), k))
