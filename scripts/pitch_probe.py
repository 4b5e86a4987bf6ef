#!/usr/bin/env python3
"""Does the row pitch of the activation matter to the LDS-DMA bf16 forward kernel?  fc7's rows are 8192 bytes apart (a power of two):
the 256 rows a workgroup fetches per K step differ only in address bits 13 and up.  Same 1 x 1 layer (16 x 41 x 41 pixels, 4096 output
channels) at several input widths around 4096: TF/s per width.   usage: pitch_probe.py [widths...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from em_adapt_tensorflow_amd import ops
widths = [int(a) for a in sys.argv[1:]] or [4096, 4160, 4128, 4224, 4352, 3968, 3840, 4096]
B, H, Cout = 16, 41, 4096
g = torch.Generator(device="cuda").manual_seed(1)
for Cin in widths:
    x = torch.randn((B, H, H, Cin), device="cuda", generator=g).relu_().bfloat16()
    w = torch.randn((1, 1, Cin, Cout), device="cuda", generator=g) * 0.05
    b = torch.randn((Cout,), device="cuda", generator=g)
    wt = ops.filter_tiles_bf16(w)
    y = torch.empty((B, H, H, Cout), dtype=torch.bfloat16, device="cuda")
    def call(): ops.conv_fwd_bf16t(x, wt, (1, 1), Cout, b, 1, relu=True, out=y)
    for _ in range(5): call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): call()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 20
    flop = 2.0 * B * H * H * Cin * Cout
    print("Cin %5d (row pitch %5d B)  %.4f ms  %7.1f TF  %5.1f %% of 2516.6   plan %s" % (Cin, Cin * 2, t, flop / t / 1e9, flop / t / 1e9 / 25.166,
          ops.conv_bf16t_plan(B, H, H, Cin, Cout, 1, 1)), flush=True)
