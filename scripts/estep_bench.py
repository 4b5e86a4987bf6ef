#!/usr/bin/env python3
"""Time the device E-step at the benchmark geometry (B=16, 41x41x21). EMADAPT_ESTEP_STOP=n stops after phase n."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from em_adapt_tensorflow_amd import ops
from em_adapt_tensorflow_amd.estep import draw_perms
B, S, C, ldl = int(os.environ.get("B", 16)), 41, 21, 24
rng = np.random.default_rng(0)
logits = torch.tensor(rng.normal(0, 2, (B, S, S, ldl)).astype(np.float32), device="cuda")
lab = np.zeros((B, S, S), np.float32)
for b in range(B):
    for c in rng.choice(np.arange(1, C), size=2 + b % 3, replace=False):
        y, x = rng.integers(0, S - 10, 2); lab[b, y:y + 10, x:x + 10] = c
lab = torch.tensor(lab, device="cuda")
perms = torch.tensor(draw_perms(5, C, seed=1), device="cuda")
out = torch.empty((B, S, S), dtype=torch.int64, device="cuda")
def run(): ops.estep_device(logits, lab, perms, C, True, 5, 1e-5, 0.4, 0.2, labels_out=out)
run(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): run()
e1.record(); torch.cuda.synchronize()
print("stop=%s  %.1f us per call" % (os.environ.get("EMADAPT_ESTEP_STOP", "0"), e0.elapsed_time(e1) / 20 * 1e3))
