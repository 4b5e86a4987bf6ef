#!/usr/bin/env python3
"""Differential fuzz of the plane-image bf16x6 kernels against the native fp32 kernels (both held to the fp64 oracle elsewhere):
random shapes, filters, dilations and launch layouts (plan_cus), every epilogue; max |difference| / max must stay below 3e-5.
  python scripts/x6p_fuzz.py [cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from em_adapt_tensorflow_amd import ops

def rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))

def main(cases=100, seed=0):
    rng = np.random.default_rng(seed)
    g = torch.Generator(device="cuda").manual_seed(seed)
    worst, ran = 0.0, {"fwd": 0, "dgrad": 0, "wgrad": 0}
    for it in range(cases):
        B, H, W = int(rng.integers(1, 7)), int(rng.integers(3, 48)), int(rng.integers(3, 48))
        Cin = int(rng.choice([16, 32, 48, 64, 128, 256, 512]))
        Cout = int(rng.choice([16, 48, 256, 272, 320, 512, 528, 1040]))
        k = int(rng.choice([1, 3, 4, 5]))
        dil = int(rng.choice([1, 2, 4, 12]))
        if k == 5 and Cin > 128: k = 3
        cus = int(rng.choice([0, 1, 2, 3, 5, 8, 13, 64]))
        dense = bool(rng.integers(0, 4) == 0)
        d = ops.conv_desc(B, H, W, Cin, Cout, k, k, dil)
        x = torch.randn((B, H, W, Cin), device="cuda", generator=g)
        w = torch.randn((k, k, Cin, Cout), device="cuda", generator=g) * 0.1
        b = torch.randn((Cout,), device="cuda", generator=g)
        dy = torch.randn((B, H, W, Cout), device="cuda", generator=g)
        gate = torch.randn((B, H, W, Cin), device="cuda", generator=g)
        ops.tune_set("plan_cus", cus); ops.tune_set("fp32_loader", 2 if dense else 0)
        try:
            errs = []
            if ops.conv_x6p_supported(d, 0):
                relu = bool(rng.integers(0, 2))
                got = ops.conv_fwd_x6p(ops.split_planes(x), ops.filter_tiles(w), d, b, relu=relu)
                errs.append(("fwd", rel(got, ops.conv_fwd(x, w, b, dil, relu=relu)))); ran["fwd"] += 1
            if ops.conv_x6p_supported(d, 1):
                got = ops.conv_dgrad_x6p(ops.split_planes(dy), ops.filter_tiles(w, dgrad=True), d, gate=gate, gate_scale=1.5)
                errs.append(("dgrad", rel(got, ops.conv_dgrad(dy, w, dil, gate=gate, gate_scale=1.5)))); ran["dgrad"] += 1
            if ops.conv_x6p_supported(d, 2):
                got = ops.conv_wgrad_x6p(ops.split_planes(x), ops.split_planes(dy), d)
                errs.append(("wgrad", rel(got, ops.conv_wgrad(x, dy, k, k, dil)))); ran["wgrad"] += 1
        finally:
            ops.tune_set("plan_cus", 0); ops.tune_set("fp32_loader", 0)
        for kind, e in errs:
            worst = max(worst, e)
            if not e < 3e-5:
                print("MISMATCH", kind, (B, H, W, Cin, Cout, k, dil), "plan_cus", cus, "dense", dense, "err %.3e" % e, flush=True)
                return 1
    print("%d cases (%s), worst max|difference|/max %.2e" % (cases, ", ".join("%s %d" % kv for kv in ran.items()), worst))
    return 0

if __name__ == "__main__":
    sys.exit(main(int(sys.argv[1]) if len(sys.argv) > 1 else 100, int(sys.argv[2]) if len(sys.argv) > 2 else 0))
