#!/usr/bin/env python3
"""LDS-DMA bf16 forward kernel (csrc/igemm_b16t.h) with parts of its K loop COMPILED out (diagnostic library,
-DEMADAPT_ABLATE_BUILD; results are garbage, only the time is read): what the DMAs, the fragment reads, the barrier and the MFMAs
cost alone and together -- the rate one CU sustains with the operand stream and the matrix stream both running, measured.
tune "ablate": 1 no DMAs, 2 no fragment reads, 4 no epilogue, 8 no MFMAs.   usage: bf16t_ablate.py [layer] [zeros]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from em_adapt_tensorflow_amd import _build, _lib
_lib._bind(_build.ABLATE_LIB_PATH)
from em_adapt_tensorflow_amd import ops
layer = sys.argv[1] if len(sys.argv) > 1 else "fc7"
zeros = len(sys.argv) > 2 and sys.argv[2] == "zeros"
H, Cin, Cout, k, rate = {"fc7": (41, 4096, 4096, 1, 1), "conv4_2": (41, 512, 512, 3, 1), "fc6": (41, 512, 4096, 4, 4),
                         "conv1_2": (321, 64, 64, 3, 1), "conv2_2": (161, 128, 128, 3, 1)}[layer]
B = 16
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.randn((B, H, H, Cin), device="cuda", generator=g).relu_().bfloat16()
w = torch.randn((k, k, Cin, Cout), device="cuda", generator=g) * 0.05
if zeros:
    x.zero_(); w.zero_()
b = torch.randn((Cout,), device="cuda", generator=g)
wt = ops.filter_tiles_bf16(w)
y = torch.empty((B, H, H, Cout), dtype=torch.bfloat16, device="cuda")
flop = 2.0 * B * H * H * k * k * Cin * Cout
def call(): ops.conv_fwd_bf16t(x, wt, (k, k), Cout, b, rate, relu=True, out=y)
print("%s forward, %s operands, plan %s" % (layer, "all-zero" if zeros else "random", ops.conv_bf16t_plan(B, H, H, Cin, Cout, k, rate)))
for name, v in (("everything", 0), ("everything", 0), ("no epilogue", 4), ("everything", 0), ("no epilogue", 4), ("no DMAs", 1),
                ("no fragment reads", 2), ("no MFMAs", 8), ("no DMAs, no reads", 3), ("no MFMAs, no DMAs (reads + barrier)", 9),
                ("no MFMAs, no reads (DMAs + barrier)", 10)):
    ops.tune_set("ablate", v)
    for _ in range(3): call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): call()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 20
    print("%-8s %-50s %7.3f ms  (%5.1f %% of 2516.6 TF if it were the whole kernel)" % (layer, name, t, flop / t / 1e9 / 25.166), flush=True)
ops.tune_set("ablate", 0)
