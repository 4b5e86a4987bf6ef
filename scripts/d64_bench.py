#!/usr/bin/env python3
"""conv1_2 (3 x 3, 64 -> 64 channels, bf16-stored tensors) on the direct kernel (csrc/conv_d64.h) against the implicit-GEMM kernel
(tune bf16_tile = 1: the round-1 kernel's automatic plan), same process, same tensors.
usage (GPU box): python scripts/d64_bench.py [--batch 16] [--size 321] [--reps 20]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from em_adapt_tensorflow_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--size", type=int, default=321)
ap.add_argument("--reps", type=int, default=20)
args = ap.parse_args()
B, H, C, k = args.batch, args.size, 64, 3
PEAK = 2516.6e12


def timed(fn):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / args.reps


g = torch.Generator(device="cuda").manual_seed(1)
x = torch.randn((B, H, H, C), device="cuda", generator=g).clamp_(min=0).bfloat16()
dy = torch.randn((B, H, H, C), device="cuda", generator=g).bfloat16()
w = torch.randn((k, k, C, C), device="cuda", generator=g) * 0.05
b = torch.zeros(C, device="cuda")
wf, wd = ops.weights_to_bf16(w)
y = torch.empty((B, H, H, C), dtype=torch.bfloat16, device="cuda")
dx = torch.empty_like(y)
fl = 2.0 * B * H * H * k * k * C * C
nbytes = {"fwd": 2 * x.numel() * 2, "dgrad": 3 * x.numel() * 2}
dwo = torch.empty((k, k, C, C), device="cuda")
dbo = torch.empty((C,), device="cuda")
nbytes["wgrad"] = 2 * x.numel() * 2
print("%-6s %10s %7s %9s   %10s %7s %9s" % ("kind", "igemm ms", "% peak", "GB/s", "direct ms", "% peak", "GB/s"))
for kind, fn in (("fwd", lambda: ops.conv_fwd_bf16(x, wf, (k, k), b, 1, relu=True, out=y)),
                 ("dgrad", lambda: ops.conv_dgrad_bf16(dy, wd, (k, k), 1, gate=x, out=dx)),
                 ("wgrad", lambda: ops.conv_wgrad_bf16(x, dy, k, k, 1, out=dwo, bias_out=dbo))):   # (with the bias gradient, as the model calls it)
    with ops.tuned(bf16_tile=1):
        t_old = timed(fn)
    assert ops.conv_bf16_plan(B, H, H, C, C, k, 1, kind)[0] == 4
    t_new = timed(fn)
    print("%-6s %10.4f %6.1f%% %9.0f   %10.4f %6.1f%% %9.0f" % (kind, t_old, 100 * fl / t_old / 1e-3 / PEAK, nbytes[kind] / t_old / 1e6,
                                                              t_new, 100 * fl / t_new / 1e-3 / PEAK, nbytes[kind] / t_new / 1e6), flush=True)
