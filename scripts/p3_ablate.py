#!/usr/bin/env python3
"""Patch kernel (csrc/conv_p3.h) with parts COMPILED out (diagnostic library, -DEMADAPT_ABLATE_BUILD; results are garbage, only the time
is read).  tune "ablate" bits: 1 no patch DMAs, 2 no filter DMAs, 4 no MFMAs, 8 no epilogue, 16 no A reads, 32 no B reads.
usage: p3_ablate.py [conv1_2|conv2_1|conv2_2] [fwd|dgrad]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from em_adapt_tensorflow_amd import _build, _lib
_lib._bind(_build.ABLATE_LIB_PATH)
from em_adapt_tensorflow_amd import ops
layer = sys.argv[1] if len(sys.argv) > 1 else "conv1_2"
kind = sys.argv[2] if len(sys.argv) > 2 else "fwd"
H, Cin, Cout = {"conv1_2": (321, 64, 64), "conv2_1": (161, 64, 128), "conv2_2": (161, 128, 128)}[layer]
B, k = 16, 3
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.randn((B, H, H, Cin), device="cuda", generator=g).relu_().bfloat16()
dy = torch.randn((B, H, H, Cout), device="cuda", generator=g).bfloat16()
w = torch.randn((k, k, Cin, Cout), device="cuda", generator=g) * 0.05
b = torch.randn((Cout,), device="cuda", generator=g)
wf, wd = ops.weights_to_bf16(w)
y = torch.empty((B, H, H, Cout), dtype=torch.bfloat16, device="cuda")
dx = torch.empty((B, H, H, Cin), dtype=torch.bfloat16, device="cuda")
flop = 2.0 * B * H * H * k * k * Cin * Cout
call = (lambda: ops.conv_fwd_bf16(x, wf, (k, k), b, 1, relu=True, out=y)) if kind == "fwd" else (lambda: ops.conv_dgrad_bf16(dy, wd, (k, k), 1, gate=x, out=dx))
print("%s %s, plan %s" % (layer, kind, ops.conv_bf16_plan(B, H, H, Cin, Cout, k, 1, kind)))
for name, v in (("everything", 0), ("no patch DMAs", 1), ("no filter DMAs", 2), ("no DMAs", 3), ("no MFMAs", 4), ("no epilogue", 8), ("no MFMAs, no epilogue", 12),
                ("no A reads", 16), ("no B reads", 32), ("no fragment reads", 48), ("no reads, no MFMAs", 52),
                ("streams + barrier alone (no reads, MFMAs, epilogue)", 60), ("barriers alone", 63), ("DMA-free MFMA loop + epilogue (no DMAs)", 3)):
    ops.tune_set("ablate", v)
    call(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): call()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 10
    print("%-8s %-52s %7.3f ms  (%5.1f %% of 2516.6 TF)" % (layer, name, t, flop / t / 1e9 / 25.166), flush=True)
ops.tune_set("ablate", 0)
