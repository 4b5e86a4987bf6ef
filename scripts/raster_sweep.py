#!/usr/bin/env python3
"""How much does the L2 hit rate of a 256x256 one-workgroup-per-CU GEMM launch matter for its TIME?  fc7 forward (plain tiles, M 26896,
N 4096, K 4096) with the XCD-aware order's group height swept (tune "raster_gm": the 32 workgroups resident on one XCD cover GM M-tiles x
32 / GM N-tiles, so a K step pulls GM + 32 / GM operand slices into that XCD's L2 instead of 64): GM = 1 and 32 are the worst orders
(33 slices per step), 4-8 the best (12).  Run under `rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum` for the hit rates.
usage: raster_sweep.py [fp32|x6p|bf16t]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from em_adapt_tensorflow_amd import ops
mode = sys.argv[1] if len(sys.argv) > 1 else "fp32"
B, H, Cin, Cout = 16, 41, 4096, 4096
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.randn((B, H, H, Cin), device="cuda", generator=g).relu_()
w = torch.randn((1, 1, Cin, Cout), device="cuda", generator=g) * 0.05
b = torch.randn((Cout,), device="cuda", generator=g)
flop = 2.0 * B * H * H * Cin * Cout
if mode == "fp32":
    y = torch.empty((B, H, H, Cout), device="cuda")
    call, peak = (lambda: ops.conv_fwd(x, w, b, 1, relu=True, out=y)), 157.3
elif mode == "x6p":
    d = ops.conv_desc(B, H, H, Cin, Cout, 1, 1, 1)
    xp, wkp = ops.split_planes(x), ops.filter_tiles(w)
    y = torch.empty((B, H, H, Cout), device="cuda")
    call, peak = (lambda: ops.conv_fwd_x6p(xp, wkp, d, b, relu=True, out=y)), 419.4
else:
    xb, wt = x.bfloat16(), ops.filter_tiles_bf16(w)
    y = torch.empty((B, H, H, Cout), dtype=torch.bfloat16, device="cuda")
    call, peak = (lambda: ops.conv_fwd_bf16t(xb, wt, (1, 1), Cout, b, 1, relu=True, out=y)), 2516.6
for gm in (1, 2, 4, 8, 16, 32, 106):
    ops.tune_set("raster_gm", gm)
    call(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): call()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 5
    print("%s fc7 forward, raster_gm %3d: %7.3f ms  %5.1f %% of %.1f TF" % (mode, gm, t, 100 * flop / t / 1e9 / peak, peak), flush=True)
ops.tune_set("raster_gm", 0)
