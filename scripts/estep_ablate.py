#!/usr/bin/env python3
"""E-step kernel with parts of the rank selection left out (diagnostic library, -DEMADAPT_ABLATE_BUILD; results are garbage,
only the time is read): what the LDS atomics, the histogram scans and the barriers of radix_select cost.
tune "estep_stop": 1..4 leave the kernel after that phase; bits 16 no atomics, 32 no scan, 64 no barriers, 128 first pass only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from em_adapt_tensorflow_amd import _build, _lib
_lib._bind(_build.ABLATE_LIB_PATH)
from em_adapt_tensorflow_amd import ops
from em_adapt_tensorflow_amd.estep import draw_perms
B, S, C, ldl = 16, 41, 21, 24
rng = np.random.default_rng(0)
lab = np.zeros((B, S, S), np.float32)
for b in range(B):
    for c in rng.choice(np.arange(1, C), size=2 + b % 3, replace=False):
        y, x = rng.integers(0, S - 10, 2); lab[b, y:y + 10, x:x + 10] = c
lab = torch.tensor(lab, device="cuda"); perms = torch.tensor(draw_perms(5, C, seed=1), device="cuda")
out = torch.empty((B, S, S), dtype=torch.int64, device="cuda")
logits = torch.tensor(rng.normal(0, 2, (B, S, S, ldl)).astype(np.float32), device="cuda")
def call(): ops.estep_device(logits, lab, perms, C, True, 5, 1e-5, 0.4, 0.2, labels_out=out)
for name, v in (("everything", 0), ("stop after phase 3 (no sweeps)", 3), ("no atomics", 16), ("no locate", 32), ("no barriers", 64),
                ("no atomics, no locate", 48), ("no atomics/locate/barriers", 112), ("first pass only", 128), ("first pass only, no atomics/locate/barriers", 240)):
    ops.tune_set("estep_stop", v)
    call(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): call()
    e1.record(); torch.cuda.synchronize()
    print("%-50s %.1f us" % (name, e0.elapsed_time(e1) / 20 * 1e3), flush=True)
ops.tune_set("estep_stop", 0)
