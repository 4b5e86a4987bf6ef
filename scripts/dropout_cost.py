#!/usr/bin/env python3
"""What the fused dropout costs the fc6 / fc7 forward launches (built-in counter-hash generator vs injected mask vs none)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from em_adapt_tensorflow_amd import ops

def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

for name, (B, H, W, Cin, Cout, k, dil) in {"fc6": (16, 41, 41, 512, 4096, 4, 4), "fc7": (16, 41, 41, 4096, 4096, 1, 1)}.items():
    x = torch.randn(B, H, W, Cin, device="cuda"); w = torch.randn(k, k, Cin, Cout, device="cuda") * 0.02
    b = torch.zeros(Cout, device="cuda"); y = torch.empty(B, H, W, Cout, device="cuda")
    mask = (torch.rand(B, H, W, Cout, device="cuda") < 0.5).to(torch.uint8)
    best = {}
    for rnd in range(3):
        for label, kw in (("relu", dict(relu=True)), ("relu+hash dropout", dict(relu=True, keep_prob=0.5, drop_seed=7)),
                          ("relu+mask dropout", dict(relu=True, keep_prob=0.5, drop_mask=mask))):
            best[label] = min(best.get(label, 1e9), timeit(lambda: ops.conv_fwd(x, w, b, dil, out=y, **kw)))
    print(name, "  ".join("%s %.3f ms" % kv for kv in best.items()), flush=True)
