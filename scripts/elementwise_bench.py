#!/usr/bin/env python3
"""The HBM-bound kernels of the step at the batch-16 / 321x321 shapes: time per call and GB/s against the bytes each one has
to move (inputs read once + outputs written once).  usage: elementwise_bench.py [fp32|bf16] [batch]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

POOLS = [("pool1", 321, 64, 2), ("pool2", 161, 128, 2), ("pool3", 81, 256, 2), ("pool4", 41, 512, 1), ("pool5", 41, 512, 1)]
BIAS = [("conv1_x", 321, 64), ("conv2_x", 161, 128), ("conv3_x", 81, 256), ("conv4_x", 41, 512), ("fc6/fc7", 41, 4096),
        ("fc8", 41, 24)]


def timed(fn, n=20):
    import torch
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    import torch
    from em_adapt_tensorflow_amd import ops
    dtype = sys.argv[1] if len(sys.argv) > 1 else "fp32"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    dt = torch.float32 if dtype == "fp32" else torch.bfloat16
    es = 4 if dtype == "fp32" else 2
    tot = {"fwd": 0.0, "bwd": 0.0, "bias": 0.0}
    print("%s storage, batch %d" % (dtype, B))
    for name, S, C, stride in POOLS:
        x = torch.randn((B, S, S, C), device="cuda").to(dt)
        O = ops.pool_out_size(S, stride)
        y = torch.empty((B, O, O, C), dtype=dt, device="cuda")
        idx = torch.empty((B, O, O, C), dtype=torch.uint8, device="cuda")
        dy = torch.randn((B, O, O, C), device="cuda").to(dt)
        dx = torch.empty_like(x)
        x.relu_()
        t_f = timed(lambda: ops.maxpool_fwd(x, stride, out=y, idx=idx))
        t_b = timed(lambda: ops.maxpool_bwd(dy, idx, tuple(x.shape), stride, gate=x, out=dx))
        # the model's form: the ReluGrad gate rides in the index bytes, the gradient reads no gate tensor
        t_fg = timed(lambda: ops.maxpool_fwd(x, stride, out=y, idx=idx, relu_gate=True))
        t_bg = timed(lambda: ops.maxpool_bwd(dy, idx, tuple(x.shape), stride, out=dx))
        by_f = x.numel() * es + y.numel() * (es + 1)
        by_b = 2 * x.numel() * es + y.numel() * (es + 1)
        by_bg = x.numel() * es + y.numel() * (es + 1)
        tot["fwd"] += t_fg
        tot["bwd"] += t_bg
        print("  %-6s %3dx%-3d C=%-4d s%d  fwd %6.1f us %5.0f GB/s  bwd(+gate tensor) %6.1f us %5.0f GB/s | gate in idx: fwd %6.1f us %5.0f GB/s  bwd %6.1f us %5.0f GB/s" % (
            name, S, S, C, stride, t_f, by_f / t_f / 1e3, t_b, by_b / t_b / 1e3, t_fg, by_f / t_fg / 1e3, t_bg, by_bg / t_bg / 1e3))
    for name, S, C in BIAS:
        Cp = (C + 7) // 8 * 8
        dy = torch.randn((B, S, S, Cp), device="cuda").to(dt)
        db = torch.empty((Cp,), dtype=torch.float32, device="cuda")
        t = timed(lambda: ops.bias_grad(dy, out=db))
        tot["bias"] += t
        print("  bias gradient %-8s M=%-8d N=%-5d %7.1f us %6.0f GB/s" % (name, B * S * S, Cp, t, dy.numel() * es / t / 1e3))
    print("  sums (pools: gate in idx): pool fwd %.1f us, pool bwd %.1f us, bias (one layer per shape) %.1f us" % (tot["fwd"], tot["bwd"], tot["bias"]))


if __name__ == "__main__":
    main()
