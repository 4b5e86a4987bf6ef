#!/usr/bin/env python3
"""What a conv kernel costs in the CONTEXT it runs in inside the step (cache state of its operands), as opposed to a hot loop.
Each case times ONLY the kernel under test (events around it), after the preparation named in the case.
usage: ctx_lab.py [layer ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from em_adapt_tensorflow_amd import ops

SHAPES = {"conv4_2": (16, 41, 41, 512, 512, 3, 1), "conv5_2": (16, 41, 41, 512, 512, 3, 2), "conv3_2": (16, 81, 81, 256, 256, 3, 1),
          "fc6": (16, 41, 41, 512, 4096, 4, 4), "fc7": (16, 41, 41, 4096, 4096, 1, 1)}
flush_buf = torch.empty(1 << 28, device="cuda")   # 1 GiB of f32: more than the 256 MiB Infinity Cache


def flush():
    flush_buf.fill_(1.0)


def timed(prep, fn, reps=7):
    ts = []
    for _ in range(reps):
        prep()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[0], ts[len(ts) // 2]


for n in [a for a in sys.argv[1:] if not a.startswith("--")] or ["conv4_2"]:
    B, H, W, Cin, Cout, k, dil = SHAPES[n]
    x = torch.relu(torch.randn(B, H, W, Cin, device="cuda")); w = torch.randn(k, k, Cin, Cout, device="cuda") * 0.05
    b = torch.zeros(Cout, device="cuda"); dy = torch.randn(B, H, W, Cout, device="cuda")
    y = torch.empty(B, H, W, Cout, device="cuda"); dx = torch.empty_like(x); dw = torch.empty_like(w)
    dgrad_g = lambda: ops.conv_dgrad(dy, w, dil, gate=x, out=dx)
    dgrad_n = lambda: ops.conv_dgrad(dy, w, dil, out=dx)
    fwd = lambda: ops.conv_fwd(x, w, b, dil, out=y)
    wgrad = lambda: ops.conv_wgrad(x, dy, k, k, dil, out=dw)
    cases = [
        ("dgrad+gate, hot loop", lambda: None, dgrad_g),
        ("dgrad no gate, hot loop", lambda: None, dgrad_n),
        ("dgrad+gate after flush (all cold)", flush, dgrad_g),
        ("dgrad no gate after flush", flush, dgrad_n),
        ("dgrad+gate after flush + wgrad (x, dy re-read: the step's order)", lambda: (flush(), wgrad()), dgrad_g),
        ("dgrad+gate after flush + touch dy", lambda: (flush(), dy.mul_(1.0)), dgrad_g),
        ("dgrad+gate after flush + touch x", lambda: (flush(), x.mul_(1.0)), dgrad_g),
        ("fwd, hot loop", lambda: None, fwd),
        ("fwd after flush", flush, fwd),
        ("fwd after flush + touch x (the step's order)", lambda: (flush(), x.mul_(1.0)), fwd),
        ("wgrad, hot loop", lambda: None, wgrad),
        ("wgrad after flush", flush, wgrad),
    ]
    for name, prep, fn in cases:
        fn(); torch.cuda.synchronize()
        lo, med = timed(prep, fn)
        print("%-8s %-70s min %7.3f  median %7.3f ms" % (n, name, lo, med), flush=True)
