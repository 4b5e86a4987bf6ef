#!/usr/bin/env python3
"""Who is how far from whom at configs[1] (16 x 321x321, 21 classes, ref preset), one micro-step: the GPU step, the fp32 CPU
oracle and the fp64 CPU oracle, pairwise, on logits and on the fc8 / fc7 / fc6 / conv5_3 / conv1_1 gradients (max-norm relative and
relative L2).  Sets the bars of
tests/test_gpu_config_steps.py.  usage: python scripts/b16_grad_margins.py [--no-fp64]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import test_gpu_config_steps as T  # noqa: E402
from em_adapt_tensorflow_amd.deeplab import ADAPT  # noqa: E402
from oracle import deeplab_oracle as do  # noqa: E402
from oracle import estep_oracle as eo  # noqa: E402

torch.set_num_threads(max(1, min(64, len(os.sched_getaffinity(0)))))
images, labels = do.synthetic_batch(T.B, T.SIZE, T.C, seed=0)
images = (images / 50.0).astype(np.float32)
rng = np.random.default_rng(1)
masks = {l: (rng.random((T.B, 41, 41, 4096)) < 0.5).astype(np.uint8) for l in ("fc6", "fc7")}
perms = eo.draw_perms(5, T.C, seed=3)
e = dict(ADAPT=ADAPT, do=do, eo=eo, torch=torch, images=images, labels=labels, masks=masks, perms=perms)
adapt, oracle32, params = T._build(e, "fp32")
logits = adapt.forward(images, keep_prob=0.5, drop_masks=masks).cpu().numpy()
adapt.micro_step(images, labels, keep_prob=0.5, perms=perms, drop_masks=masks)
torch.cuda.synchronize()
want_labels, _ = oracle32.estep_labels(logits, labels, perms)
layers = ("fc8", "fc7", "fc6", "conv5_3", "conv1_1")
res = {"gpu": dict(logits=logits, grads={l: T._grads(adapt, l) for l in layers})}
for name, dt in (("cpu32", torch.float32),) + ((("cpu64", torch.float64),) if "--no-fp64" not in sys.argv else ()):
    o = do.DeepLabOracle(category_num=T.C, preset="ref", width=64, dtype=dt)
    t0 = time.time()
    r = o.micro_step(params, images, labels, perms, weight_decay=1e-5, keep_prob=0.5, masks=masks, labels_override=want_labels)
    print("%s oracle step: %.1f s" % (name, time.time() - t0), flush=True)
    res[name] = dict(logits=r["logits"], grads={l: r["grads"][l] for l in layers})
names = list(res)
for i, a in enumerate(names):
    for b in names[i + 1:]:
        print("%s vs %s: logits max-norm %.2e" % (a, b, T._rel(res[a]["logits"], res[b]["logits"])))
        for l in layers:
            (wa, ba), (wb, bb) = res[a]["grads"][l], res[b]["grads"][l]
            print("   %-8s dw max-norm %.2e  L2 %.2e   db max-norm %.2e  L2 %.2e" % (l, T._rel(wa, wb), T._l2(wa, wb), T._rel(ba, bb), T._l2(ba, bb)))
