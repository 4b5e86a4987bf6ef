#!/usr/bin/env python3
"""Direct conv1_2 kernel (csrc/conv_d64.h) with parts COMPILED out (diagnostic library, -DEMADAPT_ABLATE_BUILD; results are garbage, only
the time is read).  tune "ablate" bits: 1 no patch DMAs behind the first, 2 no MFMAs, 4 no output stores, 8 no gate loads, 16 no fragment
reads, 32 no per-tile barrier.   usage: d64_ablate.py [fwd|dgrad]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from em_adapt_tensorflow_amd import _build, _lib
_lib._bind(_build.ABLATE_LIB_PATH)
from em_adapt_tensorflow_amd import ops
kind = sys.argv[1] if len(sys.argv) > 1 else "fwd"
B, H, C, k = 16, 321, 64, 3
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.randn((B, H, H, C), device="cuda", generator=g).relu_().bfloat16()
dy = torch.randn((B, H, H, C), device="cuda", generator=g).bfloat16()
w = torch.randn((k, k, C, C), device="cuda", generator=g) * 0.05
b = torch.randn((C,), device="cuda", generator=g)
wf, wd = ops.weights_to_bf16(w)
y = torch.empty((B, H, H, C), dtype=torch.bfloat16, device="cuda")
dx = torch.empty_like(y)
flop = 2.0 * B * H * H * k * k * C * C
call = (lambda: ops.conv_fwd_bf16(x, wf, (k, k), b, 1, relu=True, out=y)) if kind == "fwd" else (lambda: ops.conv_dgrad_bf16(dy, wd, (k, k), 1, gate=x, out=dx))
print("conv1_2 %s, plan %s" % (kind, ops.conv_bf16_plan(B, H, H, C, C, k, 1, kind)))
for name, v in (("everything", 0), ("stores skipped at run time", 64), ("no DMAs (MFMA loop + reads + epilogue arithmetic)", 65),
                ("no fragment reads", 80), ("no fragment reads, no DMAs (MFMAs + epilogue arithmetic)", 81), ("no MFMAs (DMAs + reads + epilogue arithmetic)", 66),
                ("no MFMAs, no reads (DMAs + epilogue arithmetic)", 82), ("epilogue arithmetic alone", 83), ("no epilogue (DMAs + reads + MFMAs)", 320),
                ("no epilogue, no DMAs (reads + MFMAs)", 321), ("MFMAs alone", 337), ("MFMAs alone, no barrier", 353), ("no gate loads", 8)):
    ops.tune_set("ablate", v)
    call(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): call()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 10
    print("%-60s %7.3f ms  (%5.1f %% of 2516.6 TF)" % (name, t, flop / t / 1e9 / 25.166), flush=True)
ops.tune_set("ablate", 0)
