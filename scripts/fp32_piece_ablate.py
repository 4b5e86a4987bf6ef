#!/usr/bin/env python3
"""Where does the fp32 256x256 kernel (csrc/igemm_v2.h) lose the gap between its K loop (97.5 % in the loop lab) and a layer's 89-94 %?
Parts of the kernel COMPILED out (diagnostic library, -DEMADAPT_ABLATE_BUILD; results are garbage, only the time is read), on fc7
(plain tiles in whole rounds + stream-K tail), fc6 (stream-K over packed pixel classes) and conv4_2 / conv5_2 (stream-K, short K):
    tune "ablate": 8 | 16 no operand DMAs (the loop runs on stale LDS), 32 no epilogue (nothing stored), 64 no fragment reads
The time is compared with the EXECUTED products' time at the 157.3 TF peak (emadapt_conv_executed_fraction).
usage: fp32_piece_ablate.py [fwd|dgrad] [layer ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from em_adapt_tensorflow_amd import _build, _lib
_lib._bind(_build.ABLATE_LIB_PATH)
from em_adapt_tensorflow_amd import ops
kind = sys.argv[1] if len(sys.argv) > 1 else "fwd"
layers = sys.argv[2:] or ["fc7", "fc6", "conv4_2", "conv5_2", "conv3_2"]
SHAPES = {"fc7": (41, 4096, 4096, 1, 1), "fc6": (41, 512, 4096, 4, 4), "conv4_2": (41, 512, 512, 3, 1), "conv5_2": (41, 512, 512, 3, 2),
          "conv3_2": (81, 256, 256, 3, 1), "fc6_largefov": (41, 512, 1024, 3, 12), "fc7_largefov": (41, 1024, 1024, 1, 1)}
B = 16
for layer in layers:
    H, Cin, Cout, k, rate = SHAPES[layer]
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn((B, H, H, Cin), device="cuda", generator=g).relu_()
    w = torch.randn((k, k, Cin, Cout), device="cuda", generator=g) * 0.05
    b = torch.randn((Cout,), device="cuda", generator=g)
    dy = torch.randn((B, H, H, Cout), device="cuda", generator=g)
    flop = 2.0 * B * H * H * k * k * Cin * Cout * ops.conv_executed_fraction(B, H, H, Cin, Cout, k, rate, kind)
    if kind == "fwd":
        y = torch.empty((B, H, H, Cout), device="cuda")
        call = lambda: ops.conv_fwd(x, w, b, rate, relu=True, out=y)
    else:
        dx = torch.empty((B, H, H, Cin), device="cuda")
        call = lambda: ops.conv_dgrad(dy, w, rate, gate=x, out=dx)
    base = None
    for name, v in (("everything", 0), ("no epilogue", 32), ("no operand DMAs", 24), ("no fragment reads", 64),
                    ("no DMAs, no epilogue", 56), ("no DMAs, no reads, no epilogue (MFMAs + barriers + piece set-up)", 120)):
        ops.tune_set("ablate", v)
        call(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): call()
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 5
        base = base or t
        print("%-13s %-5s %-66s %8.3f ms  %+6.1f %%   executed products at %5.1f %% of 157.3 TF" %
              (layer, kind, name, t, 100 * (t - base) / base, flop / t / 1e9 / 1.573), flush=True)
    ops.tune_set("ablate", 0)
