#!/usr/bin/env python3
"""Error against fp64 of the native fp32, the in-kernel-split bf16x6 and the plane-image bf16x6 kernels on one long reduction,
with operands restricted to fewer planes (bf16-exact x and / or w): which plane pairs carry an error."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from em_adapt_tensorflow_amd import ops
from oracle import deeplab_oracle as do
def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return "max %.2e rms %.2e" % (np.abs(a - b).max() / np.abs(b).max(), np.sqrt(((a - b) ** 2).mean()) / np.sqrt((b ** 2).mean()))
def bf(a): return torch.tensor(a).to(torch.bfloat16).to(torch.float32).numpy()
def bf2(a):   # two planes exact
    p0 = bf(a); return p0 + bf(a - p0)
rng = np.random.default_rng(3)
B, H, W, Cin, Cout, k = 2, 21, 21, 512, 256, 3
x0 = rng.normal(0, 1, (B, H, W, Cin)).astype(np.float32)
w0 = rng.normal(0, 0.05, (k, k, Cin, Cout)).astype(np.float32)
d = ops.conv_desc(B, H, W, Cin, Cout, k, k, 1)
for name, x, w in (("x fp32, w fp32", x0, w0), ("x bf16-exact, w bf16-exact", bf(x0), bf(w0)), ("x fp32, w bf16-exact", x0, bf(w0)),
                   ("x bf16-exact, w fp32", bf(x0), w0), ("x 2 planes, w bf16-exact", bf2(x0), bf(w0)), ("x bf16-exact, w 2 planes", bf(x0), bf2(w0)),
                   ("x 2 planes, w 2 planes", bf2(x0), bf2(w0))):
    y = do.conv_same(torch.tensor(x, dtype=torch.float64), torch.tensor(w, dtype=torch.float64), None, 1).numpy()
    xg, wg = torch.tensor(x, device="cuda"), torch.tensor(w, device="cuda")
    print("%-28s native %s | x6 %s | planes %s" % (name, rel(ops.conv_fwd(xg, wg, None, 1).cpu(), y),
          rel(ops.conv_fwd_x6(xg, ops.filter_kmajor(wg), (k, k), None, 1).cpu(), y),
          rel(ops.conv_fwd_x6p(ops.split_planes(xg), ops.filter_tiles(wg), d, None).cpu(), y)), flush=True)
