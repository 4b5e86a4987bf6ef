#!/usr/bin/env python3
"""Plane-image bf16x6 forward kernel (csrc/igemm_x6p.h) with parts of its K loop COMPILED out (diagnostic library,
-DEMADAPT_ABLATE_BUILD; results are garbage, only the time is read): what the DMAs, the fragment reads, the barrier and the
MFMAs cost alone and together.  tune "ablate": 1 no DMAs, 2 no fragment reads, 4 no waits / barrier, 8 no MFMAs, 16 no barrier, 32 no DMA wait."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from em_adapt_tensorflow_amd import _build, _lib
_lib._bind(os.environ.get("XP_LAB_SO", _build.ABLATE_LIB_PATH))   # (XP_LAB_SO: another diagnostic build; also switches to all-zero operands)
from em_adapt_tensorflow_amd import ops
layer = sys.argv[1] if len(sys.argv) > 1 else "fc7"
H, Cin, Cout, k, rate = {"fc7": (41, 4096, 4096, 1, 1), "conv4_2": (41, 512, 512, 3, 1), "fc6": (41, 512, 4096, 4, 4)}[layer]
B = 16
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.randn((B, H, H, Cin), device="cuda", generator=g).relu_()
w = torch.randn((k, k, Cin, Cout), device="cuda", generator=g) * 0.05
b = torch.randn((Cout,), device="cuda", generator=g)
d = ops.conv_desc(B, H, H, Cin, Cout, k, k, rate)
xp = ops.split_planes(x); wkp = ops.filter_tiles(w)
if os.environ.get("XP_LAB_SO"):   # timing only: room for 8 bytes per element, contents irrelevant
    xp = torch.zeros((x.numel() * 8,), dtype=torch.uint8, device="cuda"); wkp = torch.zeros((w.numel() * 8,), dtype=torch.uint8, device="cuda")
only = [int(v) for v in os.environ.get("XP_LAB_MASKS", "").split(",") if v]
y = torch.empty((B, H, H, Cout), device="cuda")
flop = 2.0 * B * H * H * k * k * Cin * Cout
def call(): ops.conv_fwd_x6p(xp, wkp, d, b, relu=True, out=y)
for name, v in (("everything", 0), ("no DMAs", 1), ("no fragment reads", 2), ("no waits / barrier", 4), ("no MFMAs", 8),
                ("no DMAs, no reads", 3), ("MFMAs alone (no DMAs, reads, barrier)", 7), ("no MFMAs, no DMAs (reads + barrier)", 9),
                ("no MFMAs, no reads (DMAs + barrier)", 10), ("no MFMAs, no barrier", 12), ("DMAs alone (no reads, barrier, MFMAs)", 14),
                ("no barrier (waits stay)", 16), ("no wait for the DMAs (barrier stays)", 32), ("neither barrier nor DMA wait (fragment wait stays)", 48)):
    if only and v not in only: continue
    ops.tune_set("ablate", v)
    call(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): call()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 5
    print("%-8s %-50s %7.3f ms  (%5.1f %% of 419.4 TF if it were the whole kernel)" % (layer, name, t, flop / t / 1e9 / 4.194), flush=True)
ops.tune_set("ablate", 0)
