#!/usr/bin/env python3
"""Per-layer view of a rocprofv3 --kernel-trace CSV of bench.py: maps the conv kernels of the LAST step to DeepLab
layers (launch order is fixed: forward conv1_1..fc8, backward fc8..conv1_1 as wgrad then dgrad; a 256x256 main
launch is followed by its 128x128 tail launch for the rows of the partial last round) and prints duration,
algorithmic TFLOP/s and the fraction of the matrix peak (157.3 TFLOP/s fp32; 2516.6 TFLOP/s dense bf16 when the trace
holds the bf16 kernels, a sixth of that for the bf16x6 kernels).

Two fractions per line: `alg%` = algorithmic FLOPs (2*M*K*N, padding products included: SURVEY 8d) / time / peak -- the
throughput a caller sees, above 100 % where a kernel skips the products with a dilated filter's zero padding -- and `exe%` =
the FLOPs the launch really issued (emadapt_conv_executed_fraction, the library's own host arithmetic) / time / peak: MFMA
utilisation, never above 100 %.

usage: scripts/trace_layers.py <kernel_trace.csv> [batch] [size] [classes] [preset]
"""
import csv
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from em_adapt_tensorflow_amd import ops  # noqa: E402  (host-side plan arithmetic only: works without a GPU)

PEAK = 157.3e12  # replaced by the bf16 peak in main() when the trace holds bf16 kernels


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


def short(name):
    m = re.search(r"(igemm_ws_kernel<\w+>|wgrad_ws_kernel<\d+>|igemm2?_kernel<[^>]*>|wgrad2_kernel(?:<\w+>)?|wgrad_kernel<[^>]*>|igemm_bf16\w*_kernel|igemm_b16t_kernel<\w+|conv_d64_kernel<\w+|conv_d4_kernel|conv_k32_kernel|wgrad_bf16_kernel<\d+>|wgrad_b16t_kernel|wgrad_d64_kernel"
                  r"|igemm_x6p?_kernel<[^>]*>|wgrad_x6p?_kernel<[^>]*>)", name)
    return m.group(1) if m else name[:40]


def main():
    path = sys.argv[1]
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    size = int(sys.argv[3]) if len(sys.argv) > 3 else 321
    C = int(sys.argv[4]) if len(sys.argv) > 4 else 21
    preset = sys.argv[5] if len(sys.argv) > 5 else "ref"
    L = layers(size, C, preset)
    rows = [r for r in csv.DictReader(open(path)) if re.search(r"igemm(_ws|_bf16\w*|_b16t|2|_x6p?)?_kernel|conv_d64_kernel|conv_d4_kernel|conv_k32_kernel|wgrad(_ws|_bf16|_b16t|_d64|2|_x6p?)?_kernel", r["Kernel_Name"])]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))   # (the trace file is not always in launch order)
    global PEAK
    if any("_x6" in r["Kernel_Name"] for r in rows):
        PEAK = 2516.6e12 / 6   # bf16x6: six bf16 MFMAs per product
    elif any("bf16" in r["Kernel_Name"] or "b16t" in r["Kernel_Name"] or "d64" in r["Kernel_Name"] or "conv_d4" in r["Kernel_Name"] or "conv_k32" in r["Kernel_Name"] for r in rows):
        PEAK = 2516.6e12
    # group "256x256 main + igemm_ws tail" pairs into one logical launch: a big-tile launch whose grid is a whole
    # number of rounds of the 256 CUs is followed by the wave-specialised launch over the remaining rows
    groups = []
    i = 0
    while i < len(rows):
        g = [rows[i]]
        n = rows[i]["Kernel_Name"]
        stream_k = re.search(r"igemm2_kernel<[^>]*, true>", n) is not None   # one launch over all rows: never has a tail launch
        big = ("igemm_kernel<256, 256" in n or "igemm2_kernel" in n or "igemm_x6p_kernel<false" in n or "igemm_b16t_kernel<false" in n) and not stream_k
        wgs = int(rows[i]["Grid_Size_X"]) // int(rows[i]["Workgroup_Size_X"])
        nxt = rows[i + 1]["Kernel_Name"] if i + 1 < len(rows) else ""
        if big and wgs % 256 == 0 and ("igemm_ws_kernel" in nxt or re.search(r"igemm2_kernel<[^>]*, true>", nxt) or "igemm_x6p_kernel<true" in nxt or "igemm_b16t_kernel<true" in nxt):   # + its tail launch
            g.append(rows[i + 1])
            i += 1
        groups.append(g)
        i += 1
    per_step = len(L) * 3 - 1  # conv1_1 has no dgrad
    last = groups[-per_step:]
    fwd, bwd = last[:len(L)], last[len(L):]
    tot = {"fwd": [0, 0, 0], "dgrad": [0, 0, 0], "wgrad": [0, 0, 0]}
    arith = "bf16x6" if PEAK == 2516.6e12 / 6 else "bf16" if PEAK == 2516.6e12 else "fp32"
    dil_of = {"fc6": 4 if preset == "ref" else 12, "conv5_1": 2, "conv5_2": 2, "conv5_3": 2}

    def executed(kind, name, g, h, k, cin, cout):
        """Executed / algorithmic FLOPs of this logical launch: the library's plan arithmetic for the kernel family that ran it
        (only the second-generation fp32 kernels and the bf16x6 plane kernels skip anything)."""
        names = " ".join(r["Kernel_Name"] for r in g)
        if arith == "bf16" or (arith == "bf16x6" and "_x6p_kernel" not in names):
            return 1.0
        if arith == "fp32" and not re.search(r"igemm2_kernel|wgrad2_kernel", names):
            return 1.0
        cin_p, cout_p = -(-cin // 4) * 4, -(-cout // 4) * 4   # (conv1_1's 3 channels / fc8's classes are padded in the buffers)
        return ops.conv_executed_fraction(B, h, h, cin_p, cout_p, k, dil_of.get(name, 1), kind, arith)

    def show(kind, name, g, flops, frac):
        ns = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in g)
        tf = flops / ns * 1e9
        tot[kind][0] += ns
        tot[kind][1] += flops
        tot[kind][2] += flops * frac
        kn = " + ".join(short(r["Kernel_Name"]) for r in g)
        wg = "+".join(str(int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"])) for r in g)
        print("%-6s %-8s %9.3f ms  %6.1f TF  alg %5.1f%%  exe %5.1f%%  wgs %-10s %s" % (
            kind, name, ns / 1e6, tf / 1e12, 100 * tf / PEAK, 100 * tf * frac / PEAK, wg, kn))

    for (name, h, k, cin, cout), g in zip(L, fwd):
        show("fwd", name, g, 2.0 * B * h * h * k * k * cin * cout, executed("fwd", name, g, h, k, cin, cout))
    i = 0
    for (name, h, k, cin, cout) in reversed(L):
        fl = 2.0 * B * h * h * k * k * cin * cout
        show("wgrad", name, bwd[i], fl, executed("wgrad", name, bwd[i], h, k, cin, cout))
        i += 1
        if name != "conv1_1":
            show("dgrad", name, bwd[i], fl, executed("dgrad", name, bwd[i], h, k, cin, cout))
            i += 1
    for k, (ns, fl, fe) in tot.items():
        print("TOTAL %-6s %8.3f ms  %6.1f TF  alg %5.1f%%  exe %5.1f%%" % (
            k, ns / 1e6, fl / ns * 1e9 / 1e12, 100 * fl / ns * 1e9 / PEAK, 100 * fe / ns * 1e9 / PEAK))
    ns = sum(v[0] for v in tot.values())
    fl = sum(v[1] for v in tot.values())
    fe = sum(v[2] for v in tot.values())
    print("TOTAL conv   %8.3f ms  %6.1f TF  alg %5.1f%%  exe %5.1f%%   (ideal at peak: %.1f ms algorithmic, %.1f ms executed; "
          "%.1f %% of the products are skipped padding)" % (ns / 1e6, fl / ns * 1e9 / 1e12, 100 * fl / ns * 1e9 / PEAK,
                                                           100 * fe / ns * 1e9 / PEAK, fl / PEAK * 1e3, fe / PEAK * 1e3,
                                                           100 * (1 - fe / fl)))


if __name__ == "__main__":
    main()
