"""GPU tests of the "next" rows of SURVEY section 8(f) that are built: f4 inference head (network.py:39-41) and
f3 weights I/O (init.npy pickled dict, deeplab.py:126-129,155-167; checkpoints, network.py:100-110)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch
    assert torch.cuda.is_available()
    from em_adapt_tensorflow_amd import ops
    from em_adapt_tensorflow_amd.deeplab import ADAPT
    from oracle import deeplab_oracle as do
    return ops, ADAPT, do, torch


@pytest.mark.parametrize("shape", [(2, 41, 41, 21, 24, 321), (1, 65, 65, 41, 44, 513), (3, 9, 7, 5, 8, 40)])
def test_inference_head_bilinear_argmax(env, shape):
    ops, ADAPT, do, torch = env
    B, IH, IW, C, ldl, out = shape
    rng = np.random.default_rng(sum(shape))
    logits = np.full((B, IH, IW, ldl), 1e30, np.float32)
    logits[..., :C] = rng.normal(0, 2, (B, IH, IW, C)).astype(np.float32)
    want = do.resize_bilinear_tf1(logits[..., :C], (out, out))
    pred, resc = ops.pred_bilinear_argmax(torch.tensor(logits, device="cuda:0"), C, out, out, want_rescale=True)
    assert np.array_equal(resc.cpu().numpy(), want), "interpolated logits must be bit-identical to the restatement"
    assert np.array_equal(pred.cpu().numpy(), np.argmax(want, axis=3))
    pred2, none = ops.pred_bilinear_argmax(torch.tensor(logits, device="cuda:0"), C, out, out)
    assert none is None and np.array_equal(pred2.cpu().numpy(), pred.cpu().numpy())


def test_init_model_and_checkpoint_roundtrip(env, tmp_path):
    ops, ADAPT, do, torch = env
    cfg = {"input_size": (65, 65), "category_num": 21, "width": 8, "batch_size": 2, "accum_num": 1}
    oracle = do.DeepLabOracle(category_num=21, width=8)
    params = do.init_params(oracle.specs, seed=3, he=True)
    # the EM-Adapt init model format: np.save of a dict {layer: {"w","b"}} (deeplab.py:128,160,166)
    init_path = str(tmp_path / "init.npy")
    np.save(init_path, {k: {"w": w, "b": b} for k, (w, b) in params.items()}, allow_pickle=True)
    a = ADAPT(dict(cfg, init_model_path=init_path))
    a.build()
    for layer, (w, b) in params.items():
        gw, gb = a.weights[layer][0].cpu().numpy(), a.weights[layer][1].cpu().numpy()
        if layer == "fc8":  # Xavier-uniform for filter AND bias (deeplab.py:156-164), not the file's values
            lim = (6.0 / (512 + 21)) ** 0.5
            assert np.abs(gw).max() <= lim + 1e-6 and gw.std() > 0.3 * lim
            assert np.abs(gb).max() <= (6.0 / 42) ** 0.5 + 1e-6 and np.abs(gb).max() > 0
        else:
            assert np.array_equal(gw, w) and np.array_equal(gb, b)
    a.pre_train(1e-3, 1e-5, 0.9, 2)
    images, labels = do.synthetic_batch(2, 65, 21, seed=1)
    a.micro_step(images / 50, labels, keep_prob=1.0)
    a.apply_update()
    path = a.saver["norm"].save(None, str(tmp_path / "ckpt" / "norm"), global_step=7)
    assert os.path.exists(path) and path.endswith("norm-7.npz")
    before = a._flat["params"].clone()
    mom = a._flat["momentum"].clone()
    b2 = ADAPT(dict(cfg, model_path=path))
    b2.build()
    b2.pre_train(1e-3, 1e-5, 0.9, 2)
    b2.restore_from_model(b2.saver["norm"], path)
    assert torch.equal(b2._flat["params"], before), "weights must round-trip bit-exactly"
    assert torch.equal(b2._flat["momentum"], mom), "momentum slots are saved too (the reference drops them)"
    pred = b2.predict_labels(b2.forward(images / 50))
    assert tuple(pred.shape) == (2, 65, 65) and pred.dtype == torch.int64


@pytest.mark.parametrize("case", [(375, 500, 1.1337, True), (333, 500, 0.75, False), (281, 413, None, True),
                                  (500, 375, 1.2499, False)])
def test_input_pipeline_matches_restatement(env, case):
    """f1: dataset.image_preprocess (dataset.py:147-185) on the device, randomness as explicit inputs."""
    ops, ADAPT, do, torch = env
    from em_adapt_tensorflow_amd.data import image_preprocess
    h0, w0, scale, flip = case
    rng = np.random.default_rng(h0 + w0)
    img = rng.integers(0, 256, (h0, w0, 3), dtype=np.uint8)
    lab = rng.integers(0, 21, (h0, w0), dtype=np.uint8)
    lab[:5] = 255
    want_img, want_lab = do.image_preprocess_oracle(img, lab, (321, 321), scale, flip)
    got_img, got_lab = image_preprocess(img, lab, (321, 321), scale, flip, device="cuda:0")
    assert tuple(got_img.shape) == (321, 321, 3) and tuple(got_lab.shape) == (321, 321, 1)
    assert np.array_equal(got_lab.cpu().numpy(), want_lab)
    assert np.array_equal(got_img.cpu().numpy(), want_img), np.abs(got_img.cpu().numpy() - want_img).max()


def test_batched_input_pipeline_is_the_op_by_op_one_bit_for_bit(env):
    """emadapt_voc_preprocess: a whole batch -- mixed sizes, with and without the random-scale stage, flipped and not -- from the decoded
    bytes in ONE launch equals dataset.image_preprocess sample by sample (the op list of dataset.py:147-185 as five kernels) bit for
    bit, and the NumPy restatement of the TF-1 semantics."""
    ops, ADAPT, do, torch = env
    from em_adapt_tensorflow_amd.data import image_preprocess
    rng = np.random.default_rng(11)
    cases = [(375, 500, 1.1337, True), (333, 500, 0.75, False), (281, 413, None, True), (500, 375, 1.2499, False), (17, 23, 0.9, True),
             (64, 64, None, False), (321, 321, 1.0, False)]
    for out_hw in ((321, 321), (65, 33)):
        chunks, descs, want = [], [], []
        off = 0
        for h0, w0, scale, flip in cases:
            img = rng.integers(0, 256, (h0, w0, 3), dtype=np.uint8)
            lab = rng.integers(0, 21, (h0, w0), dtype=np.uint8)
            lab[:3] = 255
            pad = int(rng.integers(0, 7))   # (samples need not be packed tightly)
            h1 = int(np.float32(h0) * np.float32(scale)) if scale is not None else 0
            w1 = int(np.float32(w0) * np.float32(scale)) if scale is not None else 0
            descs.append((off, off + h0 * w0 * 3 + pad, h0, w0, h1, w1, 1 if flip else 0))
            chunks += [img.reshape(-1), np.zeros(pad, np.uint8), lab.reshape(-1)]
            off += h0 * w0 * 4 + pad
            want.append((image_preprocess(img, lab, out_hw, scale, flip, device="cuda:0"), do.image_preprocess_oracle(img, lab, out_hw, scale, flip)))
        staging = torch.as_tensor(np.concatenate(chunks)).cuda()
        x, y = ops.voc_preprocess(staging, descs, out_hw[0], out_hw[1])
        assert tuple(x.shape) == (len(cases),) + out_hw + (3,) and tuple(y.shape) == (len(cases),) + out_hw + (1,)
        for b, ((gx, gy), (wx, wy)) in enumerate(want):
            assert torch.equal(x[b], gx) and torch.equal(y[b], gy), (out_hw, cases[b])
            assert np.array_equal(x[b].cpu().numpy(), wx) and np.array_equal(y[b].cpu().numpy(), wy), (out_hw, cases[b])
    # offsets are checked on the host before anything is launched
    from em_adapt_tensorflow_amd import _lib
    with pytest.raises(_lib.EmadaptError):
        ops.voc_preprocess(staging, [(staging.numel() - 10, 0, 8, 8, 0, 0, 0)], 33, 33)
    with pytest.raises(_lib.EmadaptError):
        ops.voc_preprocess(staging, [(0, 0, 8, 8, 9, 0, 0)], 33, 33)


def _make_voc(tmp_path, sizes, n_classes=5, seed=0):
    """A tiny VOC-style tree: JPEGImages/<id>.jpg, SegmentationClassAug/<id>.png (palette PNG), txt/{train,val}.txt."""
    from PIL import Image
    rng = np.random.default_rng(seed)
    main = tmp_path / "VOC"
    (main / "JPEGImages").mkdir(parents=True)
    (main / "SegmentationClassAug").mkdir()
    (tmp_path / "txt").mkdir()
    ids = []
    for i, (h, w) in enumerate(sizes):
        name = "2007_%06d" % i
        ids.append(name)
        Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)).save(main / "JPEGImages" / (name + ".jpg"), quality=95)
        lab = rng.integers(0, n_classes, (h, w), dtype=np.uint8)
        lab[:2] = 255
        im = Image.fromarray(lab, mode="P")
        im.putpalette([c for i in range(256) for c in ((i * 37) % 256, (i * 91) % 256, (i * 53) % 256)])
        im.save(main / "SegmentationClassAug" / (name + ".png"))
    for cat in ("train", "val"):
        (tmp_path / "txt" / (cat + ".txt")).write_text("\n".join(ids) + "\n")
    return main, tmp_path / "txt", ids


@pytest.mark.parametrize("mode", ["process", "thread"])
def test_dataset_class_feeds_the_device_pipeline(tmp_path, mode):
    """f1: the reference's ``dataset`` surface (dataset.py:6-145) over real files: lists, decode (worker processes writing into a
    pinned shared-memory ring, or a thread pool), one preprocessing launch per batch."""
    import torch
    from PIL import Image
    from em_adapt_tensorflow_amd.data import dataset
    from oracle import deeplab_oracle as do
    main, lists, ids = _make_voc(tmp_path, [(37, 53), (64, 48), (50, 50), (41, 77), (33, 35)])
    ds = dataset({"input_size": (33, 33), "categorys": ["train", "val"], "main_path": str(main), "list_path": str(lists),
                  "seed": 4, "decode_workers": 2, "decode_mode": mode})
    assert ds.get_data_len("train") == 5 and ds.get_data_len() == 5 and ds.get_cur_epoch() == 0
    assert ds.data_f["val"]["img"][0].endswith("JPEGImages/%s.jpg" % ids[0])
    # "val": no random scale, no flip -> equals the oracle's preprocessing of the decoded files, sample by sample
    seen = []
    for x, y, names in ds.next_batch("val", batch_size=2, epoches=1):
        assert x.is_cuda and x.dtype == torch.float32 and tuple(x.shape[1:]) == (33, 33, 3) and tuple(y.shape[1:]) == (33, 33, 1)
        for b, name in enumerate(names):
            img = np.asarray(Image.open(main / "JPEGImages" / (name + ".jpg")).convert("RGB"))
            lab = np.asarray(Image.open(main / "SegmentationClassAug" / (name + ".png")))
            wx, wy = do.image_preprocess_oracle(img, lab, (33, 33))
            assert np.abs(x[b].cpu().numpy() - wx).max() < 1e-3
            assert np.array_equal(y[b, :, :, 0].cpu().numpy(), wy.reshape(33, 33))
        seen += names
    assert sorted(seen) == sorted(ids)          # one epoch = every sample once (last batch is short)
    # "train": random scale + flip, endless; shapes and value ranges stay right, labels stay in the class set
    it = ds.next_batch("train", batch_size=3, epoches=-1)
    sizes = []
    for _ in range(4):
        x, y, names = next(it)
        sizes.append(len(names))
        assert tuple(x.shape) == (len(names), 33, 33, 3) and tuple(y.shape) == (len(names), 33, 33, 1)
        assert set(np.unique(y.cpu().numpy()).tolist()) <= set(range(5)) | {255.0}
        assert x.min().item() >= -123.0 and x.max().item() <= 255.0 - 104.0
    assert sizes == [3, 2, 3, 2]                # 5 samples per epoch: a full and a short batch, epoch after epoch
    it.close()
    # the batches a consumer is handed are its own tensors (the staging ring behind them is reused), and the augmentation is a
    # function of the seed alone: a second object draws the same scales and flips whatever decodes the files
    a = dataset({"input_size": (33, 33), "categorys": ["train"], "main_path": str(main), "list_path": str(lists), "seed": 9,
                 "decode_workers": 3, "decode_mode": mode, "prefetch_batches": 1})
    b = dataset({"input_size": (33, 33), "categorys": ["train"], "main_path": str(main), "list_path": str(lists), "seed": 9,
                 "decode_workers": 1, "decode_mode": "thread", "prefetch_batches": 3})
    ia, ib = a.next_batch("train", batch_size=2, epoches=-1), b.next_batch("train", batch_size=2, epoches=-1)
    kept = []
    for _ in range(9):                          # more batches than the ring has slots
        xa, ya, na = next(ia)
        xb, yb, nb_ = next(ib)
        assert na == nb_ and torch.equal(xa, xb) and torch.equal(ya, yb)
        kept.append((xa, xa.clone()))
    assert all(torch.equal(x, c) for x, c in kept), "a batch handed out earlier was overwritten by a later one"
    ia.close(); ib.close()
    # a file that cannot be decoded surfaces as an error naming it, not as a hang
    (main / "JPEGImages" / (ids[0] + ".jpg")).write_bytes(b"not a jpeg")
    bad = dataset({"input_size": (33, 33), "categorys": ["val"], "main_path": str(main), "list_path": str(lists), "seed": 4,
                   "decode_workers": 2, "decode_mode": mode})
    with pytest.raises(Exception) as ei:
        for _ in bad.next_batch("val", batch_size=2, epoches=1):
            pass
    assert ids[0] in str(ei.value)
    # palette helpers
    lab = np.array([[0, 1], [20, 255]], np.uint8)
    rgb = dataset.label2rgb(lab)
    assert tuple(rgb[0, 1]) == (128, 0, 0) and tuple(rgb[1, 1]) == (255, 255, 255)
    assert np.array_equal(dataset.rgb2label(rgb), lab)


def test_train_loop_runs_on_the_dataset_class(tmp_path):
    """ADAPT.train (deeplab.py:210-285) consuming the dataset class end to end."""
    from em_adapt_tensorflow_amd.data import dataset
    from em_adapt_tensorflow_amd.deeplab import ADAPT
    main, lists, ids = _make_voc(tmp_path, [(70, 80)] * 4, n_classes=5)
    ds = dataset({"input_size": (65, 65), "categorys": ["train"], "main_path": str(main), "list_path": str(lists), "seed": 1})
    adapt = ADAPT({"data": ds, "batch_size": 2, "input_size": (65, 65), "category_num": 5, "width": 8, "accum_num": 1,
                   "save_final": False, "saver_path": str(tmp_path / "saver")})
    n = adapt.train(base_lr=1e-3, weight_decay=1e-5, momentum=0.9, batch_size=2, epoches=2, log=lambda *a: None)
    assert n == 4 and np.isfinite(float(adapt.loss["norm"].cpu()))


def test_step_lr_schedule_and_save_cadence(env, tmp_path):
    """f2: the loop bookkeeping of deeplab.py:240-285 across its boundaries.  One iteration per epoch, 31 epochs:
    lr = base for i < 10, 1e-4 for 10 <= i < 20, 1e-5 for 20 <= i < 30, 1e-6 from i = 30 (deeplab.py:245-262); a "lr"
    checkpoint named after the OLD rate is written right before each drop and the final "norm" checkpoint at the end
    (deeplab.py:248,254,260,282); the update runs every accum_num-th iteration (deeplab.py:268-270)."""
    ops, ADAPT, do, torch = env
    from em_adapt_tensorflow_amd.data import SyntheticData
    from em_adapt_tensorflow_amd.network import latest_checkpoint
    seen = []

    class Probe(ADAPT):
        def micro_step(self, images, labels, keep_prob=0.5, perms=None, drop_masks=None):
            seen.append((self._micro, self.net["lr"], keep_prob, self._step))
            return ADAPT.micro_step(self, images, labels, keep_prob, perms, drop_masks)

    cfg = {"input_size": (33, 33), "category_num": 5, "width": 8, "accum_num": 2, "batch_size": 2,
           "saver_path": str(tmp_path / "saver")}
    data = SyntheticData(dict(cfg, length=2))       # data_len // batch_size = 1 iteration per epoch
    lines = []
    adapt = Probe(dict(cfg, data=data))
    n = adapt.train(base_lr=1e-3, weight_decay=1e-5, momentum=0.9, batch_size=2, epoches=31, log=lines.append)
    assert n == 31 and len(seen) == 31
    for i, (micro, lr, keep, step) in enumerate(seen):
        want = 1e-3 if i < 10 else 1e-4 if i < 20 else 1e-5 if i < 30 else 1e-6
        assert micro == i and lr == want and keep == 0.5, (i, lr)
        assert step == i // 2                                       # one Momentum update per accum_num micro-steps
    saver = tmp_path / "saver"
    for name in ("lr-0.001000-10.npz", "lr-0.000100-20.npz", "lr-0.000010-30.npz", "norm-31.npz"):
        assert (saver / name).exists(), sorted(p.name for p in saver.iterdir())
    assert latest_checkpoint(str(saver)).endswith("norm-31.npz")
    assert [l for l in lines if l.startswith("save model before new_lr")] == [
        "save model before new_lr:%f" % v for v in (1e-4, 1e-5, 1e-6)]
    assert any(l.startswith("epoch:10.000000, iteration:10.000000, lr:0.000100") for l in lines)   # i % 500 == 10
    # the lr checkpoint holds the weights as they were BEFORE the drop: restoring it and the final one differ
    a2 = ADAPT(dict(cfg))
    a2.build()
    a2.pre_train(1e-3, 1e-5, 0.9, 2)
    a2.restore_from_model(a2.saver["norm"], str(saver), checkpoint=True)
    assert torch.equal(a2._flat["params"], adapt._flat["params"]) and torch.equal(a2._flat["momentum"], adapt._flat["momentum"])
    a2.restore_from_model(a2.saver["lr"], str(saver / "lr-0.001000-10"))
    assert not torch.equal(a2._flat["params"], adapt._flat["params"])


def test_saver_keeps_the_last_two_and_resumes_from_the_newest(env, tmp_path):
    """network.py:100-110: Saver(max_to_keep=2) for "norm"; get_checkpoint_state picks the last written file even when
    its name sorts first ("norm-11999" < "norm-5999" as strings); a bf16 run restores an fp32 run's checkpoint."""
    ops, ADAPT, do, torch = env
    from em_adapt_tensorflow_amd.network import latest_checkpoint
    cfg = {"input_size": (33, 33), "category_num": 5, "width": 8, "batch_size": 2, "accum_num": 1}
    a = ADAPT(dict(cfg))
    a.build()
    a.pre_train(1e-2, 1e-5, 0.9, 2)
    images, labels = do.synthetic_batch(2, 33, 5, seed=1)
    d = tmp_path / "ck"
    states = {}
    for step in (5999, 11999, 17999):
        a.micro_step(images / 50, labels, keep_prob=1.0)
        a.apply_update()
        a.saver["norm"].save(None, str(d / "norm"), global_step=step)
        states[step] = (a._flat["params"].clone(), a._flat["momentum"].clone())
    names = sorted(p.name for p in d.iterdir())
    assert names == ["checkpoint", "norm-11999.npz", "norm-17999.npz"], names    # the oldest was pruned, no temp files
    assert latest_checkpoint(str(d)).endswith("norm-17999.npz")
    (d / "checkpoint").unlink()
    assert latest_checkpoint(str(d)).endswith("norm-17999.npz")                  # by step NUMBER, not by string order
    b = ADAPT(dict(cfg, compute_dtype="bf16"))      # other channel padding (8 instead of 4) in the flat buffers
    b.build()
    b.pre_train(1e-2, 1e-5, 0.9, 2)
    b.restore_from_model(b.saver["norm"], str(d), checkpoint=True)
    for layer, (w, bias) in a.weights.items():
        assert torch.equal(b.weights[layer][0], w) and torch.equal(b.weights[layer][1], bias)
        va, vb = a.saver["norm"]._momentum_views(layer), b.saver["norm"]._momentum_views(layer)
        assert torch.equal(va[0], vb[0]) and torch.equal(va[1], vb[1]) and float(va[0].abs().max()) > 0


def test_restore_accepts_the_legacy_momentum_array_and_warns_without_one(env, tmp_path):
    """Checkpoints of the first build hold the whole momentum buffer as one "__momentum__" array: restored when it fits the
    run's layout; a file without any momentum (the reference's weights-only Saver, a converted init model) restores the
    weights and says so instead of silently restarting the momentum."""
    ops, ADAPT, do, torch = env
    cfg = {"input_size": (33, 33), "category_num": 5, "width": 8, "batch_size": 2, "accum_num": 1}
    a = ADAPT(dict(cfg))
    a.build()
    a.pre_train(1e-2, 1e-5, 0.9, 2)
    images, labels = do.synthetic_batch(2, 33, 5, seed=1)
    a.micro_step(images / 50, labels, keep_prob=1.0)
    a.apply_update()
    blob = {}
    for layer, (w, b) in a.weights.items():
        blob[layer + "_weights"] = w.cpu().numpy()
        blob[layer + "_bias"] = b.cpu().numpy()
    np.savez(tmp_path / "weights_only.npz", **blob)
    np.savez(tmp_path / "legacy.npz", __momentum__=a._flat["momentum"].cpu().numpy(), **blob)
    np.savez(tmp_path / "legacy_other_layout.npz", __momentum__=np.zeros(7, np.float32), **blob)
    b = ADAPT(dict(cfg))
    b.build()
    b.pre_train(1e-2, 1e-5, 0.9, 2)
    b.restore_from_model(b.saver["norm"], str(tmp_path / "legacy"))
    assert torch.equal(b._flat["momentum"], a._flat["momentum"]) and float(b._flat["momentum"].abs().max()) > 0
    assert torch.equal(b._flat["params"], a._flat["params"])
    for name in ("weights_only", "legacy_other_layout"):
        c = ADAPT(dict(cfg))
        c.build()
        c.pre_train(1e-2, 1e-5, 0.9, 2)
        with pytest.warns(UserWarning, match="momentum left as it is"):
            c.restore_from_model(c.saver["norm"], str(tmp_path / name))
        assert torch.equal(c._flat["params"], a._flat["params"]) and float(c._flat["momentum"].abs().max()) == 0
