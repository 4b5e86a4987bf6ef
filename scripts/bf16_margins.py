#!/usr/bin/env python3
"""Print the measured quantities behind the bf16 model-test bars (tests/test_gpu_bf16.py) so their margins can be read."""
import os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from em_adapt_tensorflow_amd.deeplab import ADAPT
from oracle import deeplab_oracle as do
from oracle import estep_oracle as eo

def rel(g, w):
    g = np.asarray(g, np.float64); w = np.asarray(w, np.float64); return np.abs(g - w).max() / max(np.abs(w).max(), 1e-30)
def l2(g, w):
    g = np.asarray(g, np.float64); w = np.asarray(w, np.float64); return np.linalg.norm(g - w) / max(np.linalg.norm(w), 1e-30)

for act in ("bf16", "fp32"):
    B, width, size, C = 2, 8, 65, 5
    adapt = ADAPT({"input_size": (size, size), "category_num": C, "width": width, "accum_num": 1, "batch_size": B, "compute_dtype": "bf16", "act_dtype": act})
    adapt.build()
    oracle = do.DeepLabOracle(category_num=C, width=width, dtype=torch.float64, compute_dtype="bf16", act_dtype=act)
    params = do.init_params(oracle.specs, seed=11, he=True)
    for layer, (w, b) in params.items():
        adapt.weights[layer][0].copy_(torch.tensor(w).to(adapt.device)); adapt.weights[layer][1].copy_(torch.tensor(b).to(adapt.device))
    adapt.pre_train(base_lr=1e-3, weight_decay=1e-2, momentum=0.9, batch_size=B)
    images, labels = do.synthetic_batch(B, size, C, seed=5); images = (images / 50.0).astype(np.float32)
    S = adapt.out_hw[0]; rng = np.random.default_rng(1)
    masks = {l: (rng.random((B, S, S, adapt.weights[l][1].shape[0])) < 0.5).astype(np.uint8) for l in ("fc6", "fc7")}
    perms = eo.draw_perms(5, C, seed=77)
    logits = adapt.forward(images, keep_prob=0.5, drop_masks=masks).cpu().numpy()
    want = oracle.forward(oracle.tensors(params, requires_grad=False), images, 0.5, masks)["output"].numpy()
    adapt.micro_step(images, labels, keep_prob=0.5, perms=perms, drop_masks=masks)
    wl, _ = oracle.estep_labels(logits, labels, perms)
    ref = oracle.micro_step(params, images, labels, perms, weight_decay=1e-2, keep_prob=0.5, masks=masks, labels_override=wl)
    errs = []
    for layer, (dw, db) in ref["grads"].items():
        seg = adapt._segments[layer]; s, nw = seg["start"], seg["nw"]; kh, kw, cin, cout = seg["shape"]
        aw = adapt._flat["accum"][s:s + nw].view(seg["pshape"])[:, :, :cin, :cout].cpu().numpy()
        ab = adapt._flat["accum"][s + nw:s + nw + cout].cpu().numpy()
        errs.append((l2(aw, dw), l2(ab, db)))
    print("act=%s: logits rel %.2e (bar 1e-2) | loss rel %.2e (bar 5e-3) | grad l2 max w %.3f b %.3f (bar 8e-2) mean w %.3f (bar 5e-2)" % (
        act, rel(logits, want), abs(float(adapt.loss["norm"].cpu()) - ref["loss_norm"]) / abs(ref["loss_norm"]),
        max(e[0] for e in errs), max(e[1] for e in errs), np.mean([e[0] for e in errs])))

B, width, size, C = 4, 16, 65, 5
images, labels = do.synthetic_batch(B, size, C, seed=3); images = (images / 50.0).astype(np.float32)
tr = {}
for mode in ("fp32", "bf16"):
    random.seed(5)
    adapt = ADAPT({"input_size": (size, size), "category_num": C, "width": width, "accum_num": 2, "batch_size": B, "compute_dtype": mode, "seed": 7})
    adapt.build()
    params = do.init_params(do.layer_specs(C, "ref", width), seed=11, he=True)
    for layer, (w, b) in params.items():
        adapt.weights[layer][0].copy_(torch.tensor(w).to(adapt.device)); adapt.weights[layer][1].copy_(torch.tensor(b).to(adapt.device))
    adapt.pre_train(base_lr=2e-3, weight_decay=1e-5, momentum=0.9, batch_size=B)
    t = []
    for it in range(80):
        loss = adapt.micro_step(images, labels, keep_prob=0.5)
        if it % 2 == 1: adapt.apply_update()
        t.append(float(loss.cpu()))
    tr[mode] = np.array(t)
f, h = tr["fp32"], tr["bf16"]
print("training: fp32 start %.4f end %.4f (ratio %.2f, bar 0.7) | bf16 start %.4f end %.4f (ratio %.2f) | start diff %.2e (bar 2e-2) | end diff %.3f (bar 0.25)" % (
    f[:4].mean(), f[-10:].mean(), f[-10:].mean() / f[:4].mean(), h[:4].mean(), h[-10:].mean(), h[-10:].mean() / h[:4].mean(),
    abs(h[0] - f[0]) / f[0], abs(h[-10:].mean() - f[-10:].mean()) / f[-10:].mean()))
