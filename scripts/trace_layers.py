#!/usr/bin/env python3
"""Per-layer view of a rocprofv3 --kernel-trace CSV of bench.py: maps the conv kernels of the LAST step to
DeepLab layers (launch order is fixed: forward conv1_1..fc8, backward fc8..conv1_1 as wgrad then dgrad) and
prints duration, algorithmic TFLOP/s and fraction of the 157.3 TFLOP/s fp32-matrix peak per launch.

usage: scripts/trace_layers.py <kernel_trace.csv> [batch] [size] [classes] [preset]
"""
import csv
import sys

PEAK = 157.3e12


def layers(size=321, C=21, preset="ref", width=64):
    out = []
    cin, stride, h = 3, 1, size
    for bi, n in enumerate([2, 2, 3, 3, 3], start=1):
        cout = min(width * stride, width * 8)
        for ci in range(1, n + 1):
            out.append(("conv%d_%d" % (bi, ci), h, 3, cin, cout))
            cin = cout
        if bi <= 3:
            h = -(-h // 2)
            stride *= 2
    fcw = 4096 if preset == "ref" else 1024
    out.append(("fc6", h, 4 if preset == "ref" else 3, cin, fcw))
    out.append(("fc7", h, 1, fcw, fcw))
    out.append(("fc8", h, 1, fcw, C))
    return out


def main():
    path = sys.argv[1]
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    size = int(sys.argv[3]) if len(sys.argv) > 3 else 321
    C = int(sys.argv[4]) if len(sys.argv) > 4 else 21
    preset = sys.argv[5] if len(sys.argv) > 5 else "ref"
    L = layers(size, C, preset)
    rows = [r for r in csv.DictReader(open(path)) if "igemm_kernel" in r["Kernel_Name"] or "wgrad_kernel" in r["Kernel_Name"]]
    per_step = len(L) * 3 - 1  # conv1_1 has no dgrad
    last = rows[-per_step:]
    fwd = last[:len(L)]
    bwd = last[len(L):]
    tot = {"fwd": [0, 0], "dgrad": [0, 0], "wgrad": [0, 0]}

    def show(kind, name, r, flops):
        ns = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        tf = flops / ns * 1e9
        tot[kind][0] += ns
        tot[kind][1] += flops
        kn = r["Kernel_Name"]
        short = kn[kn.index("::") + 2:kn.index("(")].replace("(anonymous namespace)::", "")
        print("%-6s %-8s %9.3f ms  %6.1f TF  %5.1f%%  grid %7d  %s" % (kind, name, ns / 1e6, tf / 1e12, 100 * tf / PEAK,
                                                                     int(r["Grid_Size_X"]) // 256, short))

    for (name, h, k, cin, cout), r in zip(L, fwd):
        show("fwd", name, r, 2.0 * B * h * h * k * k * cin * cout)
    i = 0
    for (name, h, k, cin, cout) in reversed(L):
        fl = 2.0 * B * h * h * k * k * cin * cout
        show("wgrad", name, bwd[i], fl)
        i += 1
        if name != "conv1_1":
            show("dgrad", name, bwd[i], fl)
            i += 1
    for k, (ns, fl) in tot.items():
        print("TOTAL %-6s %8.3f ms  %6.1f TF  %5.1f%%" % (k, ns / 1e6, fl / ns * 1e9 / 1e12, 100 * fl / ns * 1e9 / PEAK))
    ns = sum(v[0] for v in tot.values())
    fl = sum(v[1] for v in tot.values())
    print("TOTAL conv   %8.3f ms  %6.1f TF  %5.1f%%   (ideal at peak: %.1f ms)" % (ns / 1e6, fl / ns * 1e9 / 1e12,
                                                                                100 * fl / ns * 1e9 / PEAK, fl / PEAK * 1e3))


if __name__ == "__main__":
    main()
