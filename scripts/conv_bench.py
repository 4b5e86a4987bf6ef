#!/usr/bin/env python3
"""Time single conv kernels (fwd / dgrad / wgrad) at DeepLab layer shapes.
usage: conv_bench.py [--bf16|--x6] [--tile=N] [--splitk=S] [layer ...]   (--tile / --splitk: emadapt_tune_set overrides)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
if any(a.startswith("--ablate=") for a in sys.argv):   # diagnostic library (-DEMADAPT_ABLATE_BUILD), bound before first use
    from em_adapt_tensorflow_amd import _build, _lib
    _lib._bind(_build.ABLATE_LIB_PATH)
from em_adapt_tensorflow_amd import ops

SHAPES = {  # name: (B, H, W, Cin, Cout, k, dil)
    "fc6": (16, 41, 41, 512, 4096, 4, 4), "fc7": (16, 41, 41, 4096, 4096, 1, 1), "conv4_2": (16, 41, 41, 512, 512, 3, 1),
    "conv5_2": (16, 41, 41, 512, 512, 3, 2), "conv3_2": (16, 81, 81, 256, 256, 3, 1), "conv2_2": (16, 161, 161, 128, 128, 3, 1),
    "conv1_2": (16, 321, 321, 64, 64, 3, 1), "conv4_2x8": (128, 41, 41, 512, 512, 3, 1), "conv4_1": (16, 41, 41, 256, 512, 3, 1),
    "conv3_1": (16, 81, 81, 128, 256, 3, 1), "conv2_1": (16, 161, 161, 64, 128, 3, 1), "conv1_1": (16, 321, 321, 4, 64, 3, 1),
    "fc8": (16, 41, 41, 4096, 24, 1, 1), "fc6lf": (16, 41, 41, 512, 1024, 3, 12), "fc6c5": (10, 65, 65, 512, 1024, 3, 12),
    "fc7p": (16, 41, 41, 4096 + 64, 4096 + 64, 1, 1), "fc7h": (16, 41, 41, 2048, 2048, 1, 1), "fc7hp": (16, 41, 41, 2048 + 32, 2048 + 32, 1, 1),
}

def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

BF16 = "--bf16" in sys.argv
X6 = "--x6" in sys.argv
names = [a for a in sys.argv[1:] if not a.startswith("--")] or ["fc6", "conv4_2", "conv2_2"]
tiles = [0]
wgens = [0]
ab_key, ab_vals = None, []   # --ab=<tune key>:<v1>,<v2>,..  generic in-process A/B over any launch-plan knob
for a in sys.argv[1:]:
    if a.startswith("--wgens="): wgens = [int(t) for t in a.split("=")[1].split(",")]
    if a.startswith("--ab="):
        ab_key, v = a.split("=", 1)[1].split(":")
        ab_vals = [int(t) for t in v.split(",")]
    if a.startswith("--tile="): ops.tune_set("fp32_tile", int(a.split("=")[1]))
    if a.startswith("--tiles="): tiles = [int(t) for t in a.split("=")[1].split(",")]   # A/B in ONE process (devices differ)
    if a.startswith("--splitk="): ops.tune_set("splitk", int(a.split("=")[1]))
    if a.startswith("--ablate="): ops.tune_set("ablate", int(a.split("=")[1]))
for n in names:
    B, H, W, Cin, Cout, k, dil = SHAPES[n]
    x = torch.randn(B, H, W, Cin, device="cuda"); w = torch.randn(k, k, Cin, Cout, device="cuda") * 0.05
    b = torch.zeros(Cout, device="cuda"); dy = torch.randn(B, H, W, Cout, device="cuda")
    y = torch.empty(B, H, W, Cout, device="cuda"); dx = torch.empty_like(x); dw = torch.empty_like(w)
    fl = 2.0 * B * H * W * k * k * Cin * Cout
    if BF16:
        if "--f32store" not in sys.argv and Cin % 8 == 0 and Cout % 8 == 0:   # bf16-stored activations, as the model keeps them
            x, dy, y, dx = x.bfloat16(), dy.bfloat16(), y.bfloat16(), dx.bfloat16()
        wf, wd = ops.weights_to_bf16(w)
        kinds = (("fwd", lambda: ops.conv_fwd_bf16(x, wf, (k, k), b, dil, out=y)),
                 ("dgrad", lambda: ops.conv_dgrad_bf16(dy, wd, (k, k), dil, out=dx)),
                 ("wgrad", lambda: ops.conv_wgrad_bf16(x, dy, k, k, dil, out=dw)),
                 ("cvt", lambda: ops.weights_to_bf16(w, wf, wd)))
        peak = 2516.6  # dense bf16 MFMA peak, TFLOP/s
    elif X6:
        wk = ops.filter_kmajor(w)
        kinds = (("fwd", lambda: ops.conv_fwd_x6(x, wk, (k, k), b, dil, out=y)),
                 ("dgrad", lambda: ops.conv_dgrad_x6(dy, w, dil, out=dx)),
                 ("wgrad", lambda: ops.conv_wgrad_x6(x, dy, k, k, dil, out=dw)),
                 ("kmaj", lambda: ops.filter_kmajor(w, wk)))
        peak = 2516.6 / 6  # six bf16 MFMAs per fp32 product
    else:
        gate = torch.randn(B, H, W, Cin, device="cuda") if "--gate" in sys.argv else None   # dgrad with the ReLU gate, as in the step
        kinds = (("fwd", lambda: ops.conv_fwd(x, w, b, dil, out=y)), ("dgrad", lambda: ops.conv_dgrad(dy, w, dil, gate=gate, out=dx)),
                 ("wgrad", lambda: ops.conv_wgrad(x, dy, k, k, dil, out=dw)))
        peak = 157.3
    for kind, fn in kinds:
        if len(tiles) > 1 and kind not in ("fwd", "dgrad"):
            continue
        tkey = "bf16_tile" if BF16 else "fp32_tile"
        if ab_key:
            best = {}
            for rnd in range(3):
                for v in ab_vals:
                    ops.tune_set(ab_key, v)
                    best[v] = min(best.get(v, 1e9), timeit(fn))
            ops.tune_set(ab_key, 0)
            print("%-10s %-5s  " % (n, kind) + "   ".join("%s=%d: %7.3f ms %5.1f%%" % (ab_key, v, best[v], fl / best[v] / 1e9 / peak * 100) for v in ab_vals), flush=True)
            continue
        if len(wgens) > 1:
            if kind != "wgrad":
                continue
            best = {}
            for rnd in range(3):
                for g in wgens:
                    ops.tune_set("wgrad_gen", g)
                    best[g] = min(best.get(g, 1e9), timeit(fn))
            ops.tune_set("wgrad_gen", 0)
            print("%-10s wgrad  " % n + "   ".join("gen %d: %7.3f ms %5.1f%%" % (g, best[g], fl / best[g] / 1e9 / peak * 100) for g in wgens), flush=True)
            continue
        best = {}
        for rnd in range(3 if len(tiles) > 1 else 1):      # interleaved rounds, minimum per variant
            for t in tiles:
                if len(tiles) > 1: ops.tune_set(tkey, t)
                ms = timeit(fn)
                best[t] = min(best.get(t, 1e9), ms)
        for t in tiles:
            if len(tiles) > 1: ops.tune_set(tkey, t)
            ms = best[t]
            plan = ""
            if not BF16 and not X6 and kind in ("fwd", "dgrad"):
                plan = " plan %s" % (ops.conv_plan(B, H, W, Cin, Cout, k, dil, dgrad=kind == "dgrad"),)
            print("%-10s %-5s %8.3f ms %7.1f TF  %5.1f%%  %s%s%s" % (n, kind, ms, fl / ms / 1e9, fl / ms / 1e9 / peak * 100,
                                                                   "bf16" if BF16 else "bf16x6" if X6 else "fp32", plan,
                                                                   " tile %d" % t if len(tiles) > 1 else ""), flush=True)
        if len(tiles) > 1: ops.tune_set(tkey, 0)
