"""End-to-end GPU parity of the ADAPT step (forward + E-step + backward + accumulate + Momentum) against
the CPU oracle (oracle/deeplab_oracle.py), through the package's reference-shaped class surface.

Bars (north_star): fp32 logits within 1e-3 relative (asserted at 1e-4 against an fp64 oracle on narrow nets,
1e-3 against the fp32 oracle at full size); label maps bit-identical to the oracle E-step on the same logits.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch
    assert torch.cuda.is_available()
    from em_adapt_tensorflow_amd.deeplab import ADAPT
    from oracle import deeplab_oracle as do
    from oracle import estep_oracle as eo
    return ADAPT, do, eo, torch


def _rel(got, want):
    want = np.asarray(want, dtype=np.float64)
    got = np.asarray(got, dtype=np.float64)
    return np.abs(got - want).max() / max(np.abs(want).max(), 1e-30)


def _build(ADAPT, do, torch, size, C, width, preset="ref", accum_num=1, B=2, he=True, dtype=None):
    adapt = ADAPT({"input_size": (size, size), "category_num": C, "width": width, "preset": preset,
                   "accum_num": accum_num, "batch_size": B})
    adapt.build()
    oracle = do.DeepLabOracle(category_num=C, preset=preset, width=width, dtype=dtype or torch.float64)
    params = do.init_params(oracle.specs, seed=11, he=he)
    assert list(params.keys()) == list(adapt.weights.keys())
    for layer, (w, b) in params.items():
        gw, gb = adapt.weights[layer]
        assert tuple(gw.shape) == w.shape and tuple(gb.shape) == b.shape
        gw.copy_(torch.tensor(w).to(gw.device))
        gb.copy_(torch.tensor(b).to(gb.device))
    return adapt, oracle, params


@pytest.mark.parametrize("preset,size,C", [("ref", 65, 5), ("largefov", 97, 21)])
def test_micro_step_matches_oracle(env, preset, size, C):
    ADAPT, do, eo, torch = env
    B, width = 2, 8
    adapt, oracle, params = _build(ADAPT, do, torch, size, C, width, preset, B=B)
    adapt.pre_train(base_lr=1e-3, weight_decay=1e-2, momentum=0.9, batch_size=B)
    images, labels = do.synthetic_batch(B, size, C, seed=5)
    images = (images / 50.0).astype(np.float32)
    S = adapt.out_hw[0]
    rng = np.random.default_rng(1)
    masks = {l: (rng.random((B, S, S, adapt.weights[l][1].shape[0])) < 0.5).astype(np.uint8) for l in ("fc6", "fc7")}
    perms = eo.draw_perms(5, C, seed=77)

    # forward
    logits = adapt.forward(images, keep_prob=0.5, drop_masks=masks).cpu().numpy()
    acts = oracle.forward(oracle.tensors(params, requires_grad=False), images, 0.5, masks)
    assert _rel(logits, acts["output"].numpy()) < 1e-4

    # one micro step; the oracle E-step sees the GPU's logits so the label maps must agree exactly
    adapt.micro_step(images, labels, keep_prob=0.5, perms=perms, drop_masks=masks)
    torch.cuda.synchronize()
    got_labels = adapt.net["e_argmax"].cpu().numpy()
    want_labels, _ = oracle.estep_labels(logits, labels, perms)
    assert np.array_equal(got_labels, want_labels)
    assert len(np.unique(got_labels)) >= 2

    ref = oracle.micro_step(params, images, labels, perms, weight_decay=1e-2, keep_prob=0.5, masks=masks,
                            labels_override=want_labels)
    assert abs(float(adapt.loss["norm"].cpu()) - ref["loss_norm"]) / abs(ref["loss_norm"]) < 1e-4
    for layer, (dw, db) in ref["grads"].items():
        seg = adapt._segments[layer]
        s, nw = seg["start"], seg["nw"]
        kh, kw, cin, cout = seg["shape"]
        acc_w = adapt._flat["accum"][s:s + nw].view(seg["pshape"])[:, :, :cin, :cout].cpu().numpy()
        acc_b = adapt._flat["accum"][s + nw:s + nw + cout].cpu().numpy()
        assert _rel(acc_w, dw) < 2e-4, layer
        assert _rel(acc_b, db) < 2e-4, layer
        # padding lanes never receive gradient
        full = adapt._flat["accum"][s:s + nw].view(seg["pshape"]).cpu().numpy()
        assert np.all(full[:, :, cin:, :] == 0) and np.all(full[:, :, :, cout:] == 0)

    # Momentum update + clean (deeplab.py:203-205)
    adapt.apply_update()
    torch.cuda.synchronize()
    for layer, (w, b) in params.items():
        dw, db = ref["grads"][layer]
        want_w = w - np.float32(1e-3) * dw  # v0 = 0 -> v = g
        assert _rel(adapt.weights[layer][0].cpu().numpy(), want_w) < 1e-5, layer
        assert _rel(adapt.weights[layer][1].cpu().numpy(), b - np.float32(1e-3) * db) < 1e-5, layer
    assert float(adapt._flat["accum"].abs().max().cpu()) == 0.0
    total = adapt.eval_loss(images, labels, keep_prob=1.0, perms=perms)
    assert np.isfinite(total) and float(adapt.loss["l2"].cpu()) > 0


def test_accumulation_over_micro_batches(env):
    """accum_num = 2: two micro-steps add g/2 each (deeplab.py:202) before one update."""
    ADAPT, do, eo, torch = env
    B, C, size = 2, 5, 49
    adapt, oracle, params = _build(ADAPT, do, torch, size, C, 8, accum_num=2, B=B)
    adapt.pre_train(1e-2, 0.0, 0.9, B)
    perms = eo.draw_perms(5, C, seed=3)
    want = None
    for mb in range(2):
        images, labels = do.synthetic_batch(B, size, C, seed=20 + mb)
        images = (images / 50.0).astype(np.float32)
        logits = adapt.forward(images, 1.0).cpu().numpy()
        lab, _ = oracle.estep_labels(logits, labels, perms)
        adapt.micro_step(images, labels, keep_prob=1.0, perms=perms)
        r = oracle.micro_step(params, images, labels, perms, weight_decay=0.0, labels_override=lab)
        g = r["grads"]["conv3_2"][0] / 2
        want = g if want is None else want + g
    seg = adapt._segments["conv3_2"]
    got = adapt._flat["accum"][seg["start"]:seg["start"] + seg["nw"]].view(seg["pshape"]).cpu().numpy()
    assert _rel(got, want) < 2e-4


def test_drop_in_surface(env):
    """The reference's attribute / method vocabulary exists and the train loop runs on a synthetic source."""
    ADAPT, do, eo, torch = env
    from em_adapt_tensorflow_amd.data import SyntheticData
    cfg = {"input_size": (65, 65), "category_num": 21, "width": 8, "accum_num": 2, "batch_size": 2, "save_final": False}
    data = SyntheticData(dict(cfg, length=8))
    adapt = ADAPT(dict(cfg, data=data))
    n = adapt.train(base_lr=1e-3, weight_decay=1e-5, momentum=0.9, batch_size=2, epoches=1, log=lambda *a: None)
    assert n == 4 and adapt._step == 2
    for key in ("input", "label", "drop_probe", "output", "relu1_1", "pool5", "drop7", "fc8", "e_argmax", "lr",
                "accum_gradient", "accum_gradient_accum", "accum_gradient_update", "accum_gradient_clean"):
        assert key in adapt.net, key
    assert list(adapt.weights)[0] == "conv1_1" and list(adapt.weights)[-1] == "fc8"
    assert tuple(adapt.weights["conv1_1"][0].shape) == (3, 3, 3, 8)
    assert tuple(adapt.weights["fc6"][0].shape) == (4, 4, 64, 512)
    assert tuple(adapt.weights["fc8"][0].shape) == (1, 1, 512, 21)
    assert len(adapt.trainable_list) == 32
    assert tuple(adapt.net["output"].shape) == (2, 9, 9, 21)
    assert adapt.net["e_argmax"].dtype == torch.int64


def test_full_size_single_image_config1(env):
    """BASELINE config 1: one 321x321 image, 21 classes, the network as written (fc6 4x4 rate 4, 4096)."""
    ADAPT, do, eo, torch = env
    adapt, oracle, params = _build(ADAPT, do, torch, 321, 21, 64, "ref", B=1, dtype=torch.float32)
    assert adapt.parameter_count() == 65_141_397 - 0 or adapt.parameter_count() > 65_000_000
    images, labels = do.synthetic_batch(1, 321, 21, seed=0)
    images = (images / 50.0).astype(np.float32)
    logits = adapt.forward(images, 1.0).cpu().numpy()
    want = oracle.forward(oracle.tensors(params, requires_grad=False), images)["output"].numpy()
    assert logits.shape == (1, 41, 41, 21)
    assert _rel(logits, want) < 1e-3
    perms = eo.draw_perms(5, 21, seed=3)
    adapt.pre_train(1e-3, 1e-5, 0.9, 1)
    adapt.micro_step(images, labels, keep_prob=1.0, perms=perms)
    torch.cuda.synchronize()
    want_labels, _ = oracle.estep_labels(logits, labels, perms)
    assert np.array_equal(adapt.net["e_argmax"].cpu().numpy(), want_labels)


@pytest.mark.parametrize("preset", ["largefov", "ref"])
def test_full_size_config5(env, preset):
    """BASELINE config 5: one 513x513 image, 41 classes -- fc6 3x3 atrous rate 12 -> 1024 ("largefov") and the network
    as written (fc6 4x4 rate 4 -> 4096, "ref").  65x65 output maps: the E-step runs its LDS-overflow path (only the
    present classes stay resident), the rate-12 gather reaches 12 pixels into the zero padding on every side."""
    ADAPT, do, eo, torch = env
    C, size = 41, 513
    adapt, oracle, params = _build(ADAPT, do, torch, size, C, 64, preset, B=1, dtype=torch.float32)
    assert tuple(adapt.weights["fc6"][0].shape) == ((3, 3, 512, 1024) if preset == "largefov" else (4, 4, 512, 4096))
    images, labels = do.synthetic_batch(1, size, C, seed=4)
    images = (images / 50.0).astype(np.float32)
    logits = adapt.forward(images, 1.0).cpu().numpy()
    want = oracle.forward(oracle.tensors(params, requires_grad=False), images)["output"].numpy()
    assert logits.shape == (1, 65, 65, C)
    assert _rel(logits, want) < 1e-3                      # north_star: fp32 logits within 1e-3 relative
    perms = eo.draw_perms(5, C, seed=6)
    adapt.pre_train(1e-3, 1e-5, 0.9, 1)
    adapt.micro_step(images, labels, keep_prob=1.0, perms=perms)
    adapt.apply_update()
    torch.cuda.synchronize()
    want_labels, _ = oracle.estep_labels(logits, labels, perms)
    got = adapt.net["e_argmax"].cpu().numpy()
    assert np.array_equal(got, want_labels)               # bit-identical label map
    assert len(np.unique(got)) >= 2 and np.isfinite(float(adapt.loss["norm"].cpu()))


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_fused_update_leaves_the_same_weights_bit_for_bit(env, dtype):
    """config "fused_update" (what train() switches on for accum_num == 1 without a gradient exchange): backward() skips the
    per-layer accumulate launches and apply_update() runs accumulate + Momentum + clean as one pass per layer
    (emadapt_sgd_accumulate_momentum).  Weights and momentum after three steps must be bit-identical to the three-op path of
    deeplab.py:202-205, and the accumulator must stay zero."""
    ADAPT, do, eo, torch = env
    B, size, C, width = 2, 65, 5, 8
    images, labels = do.synthetic_batch(B, size, C, seed=3)
    images = (images / 50.0).astype(np.float32)
    perms = eo.draw_perms(5, C, seed=7)
    flats = []
    for fused in (False, True):
        adapt = ADAPT({"input_size": (size, size), "category_num": C, "width": width, "accum_num": 1, "batch_size": B,
                       "compute_dtype": dtype, "fused_update": fused, "seed": 5})
        adapt.build()
        adapt.pre_train(base_lr=1e-2, weight_decay=1e-3, momentum=0.9, batch_size=B)
        for _ in range(3):
            adapt.micro_step(images, labels, keep_prob=0.5, perms=perms)
            if fused:
                assert float(adapt._flat["accum"].abs().max()) == 0.0   # bypassed
            adapt.apply_update()
        torch.cuda.synchronize()
        assert float(adapt._flat["accum"].abs().max()) == 0.0
        flats.append((adapt._flat["params"].clone(), adapt._flat["momentum"].clone()))
    assert torch.equal(flats[0][0], flats[1][0]) and torch.equal(flats[0][1], flats[1][1])
    assert float(flats[0][1].abs().max()) > 0


def test_optimizer_pass_writes_the_tiled_images_the_refresh_would(env):
    """config "fused_tiles" (bf16, fused update): the optimizer pass of a layer whose forward and data gradient run on the LDS-DMA kernel
    also writes the layer's two tiled bf16 images and the next forward's refresh skips them.  Full-width net (so that layers with
    Cin % 256 == 0 and Cout % 256 == 0 exist), the LDS-DMA kernel forced onto every layer it supports: after three steps weights,
    momentum, images and the loss of a further forward must be bit-identical to the path whose refresh rebuilds the images."""
    ADAPT, do, eo, torch = env
    B, size, C = 1, 65, 5
    images, labels = do.synthetic_batch(B, size, C, seed=3)
    images = (images / 50.0).astype(np.float32)
    perms = eo.draw_perms(5, C, seed=7)
    outs = []
    for tiles in (False, True):
        adapt = ADAPT({"input_size": (size, size), "category_num": C, "accum_num": 1, "batch_size": B, "compute_dtype": "bf16",
                       "fused_update": True, "fused_tiles": tiles, "bf16t": "always", "seed": 5})
        adapt.build()
        adapt.pre_train(base_lr=1e-2, weight_decay=1e-3, momentum=0.9, batch_size=B)
        took = 0
        for _ in range(3):
            adapt.micro_step(images, labels, keep_prob=0.5, perms=perms)
            adapt.apply_update()
            took = max(took, len(adapt._tiles_fresh))
        assert (took > 0) == tiles
        loss = adapt.eval_loss(images, labels, keep_prob=1.0, perms=perms)   # (its forward refreshes what is not fresh)
        assert len(adapt._tiles_fresh) == 0
        torch.cuda.synchronize()
        imgs = [op.seg[k].clone() for op in adapt._plan if op.kind == "conv" for k in ("wt16", "wtd16") if op.seg.get(k) is not None]
        outs.append((adapt._flat["params"].clone(), adapt._flat["momentum"].clone(), imgs, loss))
        del adapt
        torch.cuda.empty_cache()
    a, b = outs
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and a[3] == b[3] and np.isfinite(a[3])
    assert len(a[2]) == len(b[2]) > 0
    for x, y in zip(a[2], b[2]):
        assert torch.equal(x.view(torch.int16), y.view(torch.int16))
