#!/usr/bin/env python3
"""Weights after a few full-size training steps with a given build of the library, as a checksum: two builds whose kernels add the same
numbers in the same order must print the same line (the in-launch stream-K combine of round 4 against the second-stage kernel it
replaced: head + next + next, ascending k, in both).
usage: lib_bitwise.py <libemadapt_hip.so> [fp32|bf16|bf16x6] [steps]"""
import hashlib, os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from em_adapt_tensorflow_amd import _lib
_lib._bind(os.path.abspath(sys.argv[1]))
from em_adapt_tensorflow_amd.data import SyntheticData
from em_adapt_tensorflow_amd.deeplab import ADAPT
from em_adapt_tensorflow_amd.estep import draw_perms
dtype = sys.argv[2] if len(sys.argv) > 2 else "fp32"
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
cfg = {"input_size": (321, 321), "category_num": 21, "accum_num": 1, "batch_size": 16, "preset": "ref", "seed": 2, "device": "cuda:0",
       "compute_dtype": dtype}
data = SyntheticData(dict(cfg, seed=0, unique_batches=2))
adapt = ADAPT(dict(cfg, data=data))
adapt.build()
adapt.pre_train(base_lr=1e-3, weight_decay=1e-5, momentum=0.9, batch_size=16)
batches = data.next_batch("train", 16)
pool = [next(batches) for _ in range(2)]
random.seed(3)
adapt._perm_pool = torch.as_tensor(np.stack([draw_perms(5, 21, rng=random) for _ in range(steps)])).to("cuda:0")
adapt._perm_pos = 0
losses = []
for i in range(steps):
    x, y = pool[i % 2]
    losses.append(adapt.micro_step(x, y, keep_prob=0.5))
    adapt.apply_update()
torch.cuda.synchronize()
p = adapt._flat["params"].cpu().numpy()
print("%s %s steps %d  params sha256 %s  losses %s" % (os.path.basename(sys.argv[1]), dtype, steps, hashlib.sha256(p.tobytes()).hexdigest()[:24],
                                                      " ".join("%.9g" % float(l) for l in losses)), flush=True)
