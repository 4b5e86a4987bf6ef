#!/usr/bin/env python3
"""bf16 forward kernel (256x256x64 tiles) with parts COMPILED out (diagnostic library, one instantiation per mask; results are
garbage, only the time is read):
what bounds it -- operand delivery (global loads), the LDS store path, the barriers, the fragment reads or the MFMAs.
tune "ablate" bits: 1 no global loads, 2 no LDS stores, 4 no barriers, 8 no A loads, 16 no B loads, 32 fragments read in the
first K step only, 64 no MFMAs.   usage: bf16_ablate.py [layer ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from em_adapt_tensorflow_amd import _build, _lib
_lib._bind(_build.ABLATE_LIB_PATH)
from em_adapt_tensorflow_amd import ops
SHAPES = {"fc7": (16, 41, 41, 4096, 4096, 1, 1), "fc6": (16, 41, 41, 512, 4096, 4, 4), "conv4_2": (16, 41, 41, 512, 512, 3, 1)}
CASES = [("everything", 0), ("no global loads", 1), ("no A loads", 8), ("no B loads", 16), ("no LDS stores", 2), ("no loads, no stores", 3),
         ("no loads / stores / barriers", 7), ("no fragment reads", 32), ("no MFMAs", 64), ("no fragment reads, no MFMAs", 96),
         ("no loads / stores, no fragment reads", 35), ("MFMAs only (no loads / stores / reads / barriers)", 39), ("no barriers", 4)]
for n in [a for a in sys.argv[1:]] or ["fc7", "conv4_2"]:
    B, H, W, Cin, Cout, k, dil = SHAPES[n]
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(B, H, W, Cin, device="cuda", generator=g).bfloat16()
    w = torch.randn(k, k, Cin, Cout, device="cuda", generator=g) * 0.05
    b = torch.zeros(Cout, device="cuda"); y = torch.empty(B, H, W, Cout, device="cuda", dtype=torch.bfloat16)
    wf, wd = ops.weights_to_bf16(w)
    fl = 2.0 * B * H * W * k * k * Cin * Cout
    fn = lambda: ops.conv_fwd_bf16(x, wf, (k, k), b, dil, out=y)
    for _ in range(3): fn()
    best = {}
    for rnd in range(3):
        for name, v in CASES:
            ops.tune_set("ablate", v)
            fn(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): fn()
            e1.record(); torch.cuda.synchronize()
            best[name] = min(best.get(name, 1e9), e0.elapsed_time(e1) / 5)
    ops.tune_set("ablate", 0)
    for name, v in CASES:
        print("%-8s %-58s %7.3f ms  %5.1f %% of the bf16 peak (as if all FLOPs were done)" % (n, name, best[name], fl / best[name] / 1e9 / 2516.6 * 100), flush=True)
