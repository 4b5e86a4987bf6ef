#!/usr/bin/env python3
"""Does replaying the training step as one hipGraph shrink the gaps between its ~250 kernels?

Captures micro_step + apply_update (static inputs, one fixed E-step order; the dropout seed is frozen in the
capture, which is fine for a timing probe) with torch.cuda.graph and times eager launches against replays in the
same process.  usage: graph_probe.py [fp32|bf16] [batch]
"""
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    import torch
    from em_adapt_tensorflow_amd.data import SyntheticData
    from em_adapt_tensorflow_amd.deeplab import ADAPT
    from em_adapt_tensorflow_amd.estep import draw_perms
    dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    dev = torch.device("cuda", 0)
    cfg = {"input_size": (321, 321), "category_num": 21, "accum_num": 1, "batch_size": B, "seed": 2, "device": dev,
           "compute_dtype": dtype}
    data = SyntheticData(dict(cfg, seed=0, unique_batches=1))
    a = ADAPT(dict(cfg, data=data))
    a.build()
    a.pre_train(base_lr=1e-3, weight_decay=1e-5, momentum=0.9, batch_size=B)
    x, y = next(data.next_batch("train", B))
    x, y = x.to(dev).float().contiguous(), y.to(dev).float().contiguous()
    random.seed(3)
    perms = torch.as_tensor(np.stack([draw_perms(5, 21, rng=random)])).to(dev)[0].to(torch.int32).contiguous()

    def step():
        a.micro_step(x, y, keep_prob=0.5, perms=perms)
        a.apply_update()

    def timed(fn, n=20):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    eager = timed(step)
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        step()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        step()
    torch.cuda.synchronize()
    replay = timed(g.replay)
    eager2 = timed(step)
    print("%s batch %d: eager %.3f ms/step, graph replay %.3f ms/step, eager again %.3f  (loss %.5f)" % (
        dtype, B, eager, replay, eager2, float(a.loss["norm"].cpu())))


if __name__ == "__main__":
    main()
