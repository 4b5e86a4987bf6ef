#!/usr/bin/env python3
"""bf16 forward / data gradient per benchmark layer (configs[2]: 16 x 321x321, ref preset): the round-1 kernel (register-staged
producers) against the LDS-DMA kernel (igemm_b16t.h), same process, same tensors, HIP events around N back-to-back launches on random
operands.  usage: python scripts/bf16t_bench.py [--reps 20] [--preset ref|largefov] [--tune k=v,...]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from em_adapt_tensorflow_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--preset", default="ref")
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--tune", default="")
ap.add_argument("--layers", default="")
ap.add_argument("--kinds", default="", help="fwd,dgrad,wgrad (wgrad: old = the round-1 kernel via tune wgrad_gen = 1, new = csrc/wgrad_b16t.h where it applies)")
args = ap.parse_args()
for kv in filter(None, args.tune.split(",")):
    ops.tune_set(kv.split("=")[0], int(kv.split("=")[1]))
PEAK = 2516.6e12
B = args.batch
fc6 = (41, 512, 4096, 4, 4) if args.preset == "ref" else (41, 512, 1024, 3, 12)
fcw = fc6[2]
LAYERS = [("conv1_2", (321, 64, 64, 3, 1)), ("conv2_1", (161, 64, 128, 3, 1)), ("conv2_2", (161, 128, 128, 3, 1)),
          ("conv3_1", (81, 128, 256, 3, 1)), ("conv3_2", (81, 256, 256, 3, 1)), ("conv4_1", (41, 256, 512, 3, 1)),
          ("conv4_2", (41, 512, 512, 3, 1)), ("conv5_2", (41, 512, 512, 3, 2)), ("fc6", fc6), ("fc7", (41, fcw, fcw, 1, 1))]
if args.layers:
    LAYERS = [l for l in LAYERS if l[0] in args.layers.split(",")]


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


tot = {"old": 0.0, "new": 0.0}
print("%-8s %-6s %9s %7s   %9s %7s   %s" % ("layer", "kind", "old ms", "% peak", "b16t ms", "% peak", "plan (rows, sk, levels, tail)"))
for name, (H, Cin, Cout, k, dil) in LAYERS:
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn((B, H, H, Cin), device="cuda", generator=g).clamp_(min=0).bfloat16()
    dy = torch.randn((B, H, H, Cout), device="cuda", generator=g).bfloat16()
    w = torch.randn((k, k, Cin, Cout), device="cuda", generator=g) * 0.05
    b = torch.zeros(Cout, device="cuda")
    wf, wd = ops.weights_to_bf16(w)
    wt, wtd = ops.filter_tiles_bf16(w), ops.filter_tiles_bf16(w, dgrad=True)
    y = torch.empty((B, H, H, Cout), dtype=torch.bfloat16, device="cuda")
    dx = torch.empty((B, H, H, Cin), dtype=torch.bfloat16, device="cuda")
    fl = 2.0 * B * H * H * k * k * Cin * Cout
    dwo = torch.empty((k, k, Cin, Cout), device="cuda")

    def wgrad_old():
        with ops.tuned(wgrad_gen=1):
            ops.conv_wgrad_bf16(x, dy, k, k, dil, out=dwo)

    for kind, old, new in (("fwd", lambda: ops.conv_fwd_bf16(x, wf, (k, k), b, dil, relu=True, out=y),
                            lambda: ops.conv_fwd_bf16t(x, wt, (k, k), Cout, b, dil, relu=True, out=y)),
                           ("dgrad", lambda: ops.conv_dgrad_bf16(dy, wd, (k, k), dil, gate=x, out=dx),
                            lambda: ops.conv_dgrad_bf16t(dy, wtd, (k, k), Cin, dil, gate=x, out=dx)),
                           ("wgrad", wgrad_old, lambda: ops.conv_wgrad_bf16(x, dy, k, k, dil, out=dwo))):
        if args.kinds and kind not in args.kinds.split(","):
            continue
        t_old, t_new = timed(old), timed(new)
        tot["old"] += t_old
        tot["new"] += t_new
        print("%-8s %-6s %9.4f %6.1f%%   %9.4f %6.1f%%   %s" % (name, kind, t_old, 100 * fl / t_old / 1e-3 / PEAK, t_new,
                                                             100 * fl / t_new / 1e-3 / PEAK, ops.conv_bf16_plan(B, H, H, Cin, Cout, k, dil, kind) if kind == "wgrad" else ops.conv_bf16t_plan(B, H, H, Cin, Cout, k, dil, kind)), flush=True)
print("sum: old %.3f ms, b16t %.3f ms" % (tot["old"], tot["new"]))
