#!/usr/bin/env python3
"""E-step kernel: time per call, achieved GB/s against the algorithmic bytes, and where the time goes.

Runs the shipped library first (total time, both geometries), then the diagnostic library (-DEMADAPT_ABLATE_BUILD, built by
`python -m em_adapt_tensorflow_amd._build --ablate`) with emadapt_tune_set("estep_stop", n) to leave the kernel after phase
n: the differences are the phase times.  usage: estep_bench.py  (prints the table committed as profiles/rNN_estep_bench.txt)
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

GEOMS = [("321x321 input, 21 classes (BASELINE configs[1])", 16, 41, 21, 24),
         ("513x513 input, 41 classes (BASELINE configs[4])", 10, 65, 41, 44)]
PHASES = ["1 label sets, present-class list, staging of the logits into LDS, pairwise-sum program",
          "2 suppress absent classes + per-pixel running max",
          "3 mean0 (NumPy pairwise tree)",
          "4 num_iter x present classes: k-th order statistic by radix select + add (estep.py:62-79)",
          "5 re-centre + argmax + stores (= whole kernel)"]


def run(ablate):
    import numpy as np
    import torch
    if ablate:
        from em_adapt_tensorflow_amd import _build, _lib
        _lib._bind(_build.ABLATE_LIB_PATH)
    from em_adapt_tensorflow_amd import ops
    from em_adapt_tensorflow_amd.estep import draw_perms
    for title, B, S, C, ldl in GEOMS:
        rng = np.random.default_rng(0)
        logits = torch.tensor(rng.normal(0, 2, (B, S, S, ldl)).astype(np.float32), device="cuda")
        lab = np.zeros((B, S, S), np.float32)
        present = []
        for b in range(B):
            cls = rng.choice(np.arange(1, C), size=2 + b % 3, replace=False)
            present.append(len(cls) + 1)
            for c in cls:
                y, x = rng.integers(0, S - 10, 2)
                lab[b, y:y + 10, x:x + 10] = c
        lab = torch.tensor(lab, device="cuda")
        perms = torch.tensor(draw_perms(5, C, seed=1), device="cuda")
        out = torch.empty((B, S, S), dtype=torch.int64, device="cuda")
        fmap = torch.empty((B, S, S, C), dtype=torch.float32, device="cuda")

        def call():
            ops.estep_device(logits, lab, perms, C, True, 5, 1e-5, 0.4, 0.2, labels_out=out)   # labels only, as the training step calls it

        def timed():
            call()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                call()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / 20 * 1e3

        alg = B * S * S * (C * 4 + 8)
        if not ablate:
            us = timed()
            print("%s: B=%d, %dx%dx%d, %.1f present classes per image" % (title, B, S, S, C, sum(present) / B))
            print("  shipped kernel (incl. the global-max pre-kernel): %.1f us per call; algorithmic bytes %.2f MB -> %.1f GB/s "
                  "(%.2f %% of 8 TB/s); %d workgroups on 256 CUs" % (us, alg / 1e6, alg / us / 1e3, alg / us / 1e3 / 80, B))
        else:
            prev = 0.0
            print("%s: phase table (diagnostic build, cumulative -> own time):" % title)
            for n in (1, 2, 3, 4, 0):
                ops.tune_set("estep_stop", n)
                us = timed()
                print("    %8.1f us -> %7.1f us  phase %s" % (us, us - prev, PHASES[(n - 1) % 5]))
                prev = us
            ops.tune_set("estep_stop", 0)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        run(sys.argv[1] == "ablate")
    else:  # two processes: each binds its own library
        for mode in ("shipped", "ablate"):
            sys.stdout.flush()
            subprocess.call([sys.executable, os.path.abspath(__file__), mode])
