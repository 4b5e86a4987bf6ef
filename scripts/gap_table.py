#!/usr/bin/env python3
"""Idle time between consecutive kernels of a rocprofv3 --kernel-trace csv: total, and grouped by the kernel that FOLLOWS
the gap (what the queue waited to start).  usage: gap_table.py <kernel_trace.csv> [skip_first_n_kernels]"""
import collections
import csv
import re
import sys


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("emadapt::", "")
    n = re.sub(r"^void\s+", "", n)
    n = re.sub(r"\(.*$", "", n)
    return n[:70]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rows = rows[skip:]
    by_next = collections.defaultdict(lambda: [0, 0.0])
    by_prev = collections.defaultdict(lambda: [0, 0.0])
    busy = 0.0
    gaps = 0.0
    big = 0.0
    for a, b in zip(rows, rows[1:]):
        g = (int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3
        busy += (int(a["End_Timestamp"]) - int(a["Start_Timestamp"])) / 1e3
        if g > 200:    # host-side pause (step boundary with a fence, the CPU baseline): not a launch gap
            big += g
            continue
        gaps += g
        by_next[short(b["Kernel_Name"])][0] += 1
        by_next[short(b["Kernel_Name"])][1] += g
        by_prev[short(a["Kernel_Name"])][0] += 1
        by_prev[short(a["Kernel_Name"])][1] += g
    n = len(rows) - 1
    print("%d kernels: busy %.1f us, gaps %.1f us (%.2f %% of busy+gaps), mean gap %.2f us; pauses > 200 us excluded: %.0f us" % (
        n, busy, gaps, 100 * gaps / (busy + gaps), gaps / max(n, 1), big))
    for title, tab in (("gap BEFORE kernel", by_next), ("gap AFTER kernel", by_prev)):
        print("\n%-72s %7s %10s %8s" % (title, "count", "total us", "mean us"))
        for k, (c, t) in sorted(tab.items(), key=lambda kv: -kv[1][1])[:25]:
            print("%-72s %7d %10.1f %8.2f" % (k, c, t, t / c))


if __name__ == "__main__":
    main()
