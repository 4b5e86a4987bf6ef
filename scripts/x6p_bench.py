#!/usr/bin/env python3
"""Plane-image bf16x6 kernels (csrc/conv_x6p.hip) against the native fp32 kernels, layer by layer at the benchmark shapes:
max |difference| relative to the tensor's maximum, time and TFLOP/s (algorithmic, of the 419.4 TF six-MFMA peak).
  python scripts/x6p_bench.py [--layers fc7,conv4_2] [--batch 16] [--kinds fwd,dgrad,wgrad] [--iters 5]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from em_adapt_tensorflow_amd import ops

LAYERS = {  # name: (H, Cin, Cout, k, rate)
    "conv1_2": (321, 64, 64, 3, 1), "conv2_1": (161, 64, 128, 3, 1), "conv2_2": (161, 128, 128, 3, 1),
    "conv3_1": (81, 128, 256, 3, 1), "conv3_2": (81, 256, 256, 3, 1), "conv4_1": (41, 256, 512, 3, 1),
    "conv4_2": (41, 512, 512, 3, 1), "conv5_2": (41, 512, 512, 3, 2), "fc6": (41, 512, 4096, 4, 4), "fc7": (41, 4096, 4096, 1, 1),
}
ap = argparse.ArgumentParser()
ap.add_argument("--layers", default="fc7,conv4_2,conv5_2,conv3_2,fc6")
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--kinds", default="fwd,dgrad,wgrad")
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--tune", default="")
args = ap.parse_args()
for kv in filter(None, args.tune.split(",")):
    k, v = kv.split("=")
    ops.tune_set(k, int(v))
PEAK = 2516.6 / 6


def timed(fn, iters):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def rel(a, b):
    return float((a - b).abs().max() / b.abs().max())


g = torch.Generator(device="cuda").manual_seed(1)
for name in args.layers.split(","):
    H, Cin, Cout, k, rate = LAYERS[name]
    B = args.batch
    x = torch.randn((B, H, H, Cin), device="cuda", generator=g).relu_()
    w = torch.randn((k, k, Cin, Cout), device="cuda", generator=g) * 0.05
    b = torch.randn((Cout,), device="cuda", generator=g)
    dy = torch.randn((B, H, H, Cout), device="cuda", generator=g)
    d = ops.conv_desc(B, H, H, Cin, Cout, k, k, rate)
    flop = 2.0 * B * H * H * k * k * Cin * Cout
    for kind in args.kinds.split(","):
        if kind == "fwd":
            if not ops.conv_x6p_supported(d, 0): print("%-8s fwd    not taken by the plane kernels" % name); continue
            ref = ops.conv_fwd(x, w, b, rate, relu=True)
            xp = ops.split_planes(x)
            wkp = ops.filter_tiles(w)
            y = ops.conv_fwd_x6p(xp, wkp, d, b, relu=True)
            err = rel(y, ref)
            t = timed(lambda: ops.conv_fwd_x6p(xp, wkp, d, b, relu=True, out=y), args.iters)
            t0 = timed(lambda: ops.conv_fwd(x, w, b, rate, relu=True, out=ref), args.iters)
            ts = timed(lambda: ops.split_planes(x, out=xp), args.iters)
            print("%-8s fwd    %8.3f ms %7.1f TF %5.1f %%   native fp32 %8.3f ms   split(x) %.3f ms   max rel diff %.2e   plan %s"
                  % (name, t, flop / t / 1e9, flop / t / 1e9 / PEAK * 100, t0, ts, err, ops.conv_x6p_plan(d, False)), flush=True)
        elif kind == "dgrad":
            if not ops.conv_x6p_supported(d, 1): print("%-8s dgrad  not taken by the plane kernels" % name); continue
            gate = x
            ref = ops.conv_dgrad(dy, w, rate, gate=gate, gate_scale=1.0)
            dyp = ops.split_planes(dy)
            wp = ops.filter_tiles(w, dgrad=True)
            dx = ops.conv_dgrad_x6p(dyp, wp, d, gate=gate, gate_scale=1.0)
            err = rel(dx, ref)
            t = timed(lambda: ops.conv_dgrad_x6p(dyp, wp, d, gate=gate, gate_scale=1.0, out=dx), args.iters)
            t0 = timed(lambda: ops.conv_dgrad(dy, w, rate, gate=gate, gate_scale=1.0, out=ref), args.iters)
            print("%-8s dgrad  %8.3f ms %7.1f TF %5.1f %%   native fp32 %8.3f ms   max rel diff %.2e   plan %s"
                  % (name, t, flop / t / 1e9, flop / t / 1e9 / PEAK * 100, t0, err, ops.conv_x6p_plan(d, True)), flush=True)
        elif kind == "wgrad" and hasattr(ops, "conv_wgrad_x6p"):
            if not ops.conv_x6p_supported(d, 2): print("%-8s wgrad  not taken by the plane kernels" % name); continue
            ref = ops.conv_wgrad(x, dy, k, k, rate)
            xp = ops.split_planes(x)
            dyp = ops.split_planes(dy)
            dw = ops.conv_wgrad_x6p(xp, dyp, d)
            err = rel(dw, ref)
            t = timed(lambda: ops.conv_wgrad_x6p(xp, dyp, d, out=dw), args.iters)
            t0 = timed(lambda: ops.conv_wgrad(x, dy, k, k, rate, out=ref), args.iters)
            print("%-8s wgrad  %8.3f ms %7.1f TF %5.1f %%   native fp32 %8.3f ms   max rel diff %.2e"
                  % (name, t, flop / t / 1e9, flop / t / 1e9 / PEAK * 100, t0, err), flush=True)
