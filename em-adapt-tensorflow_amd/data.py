"""Synthetic input source with the call shape the training loop expects from ``dataset``
(reference dataset.py:107-145: ``next_batch(category, batch_size, epoches)`` and ``get_data_len()``).

The reference's real pipeline (VOC file lists, tf.data decode/scale/flip) is outside the hot path and is
not rebuilt here (section 8 f1 lists it as the next row).  This class
produces the benchmark inputs of SURVEY section 8(d) directly in HBM: images ~ N(0, 50^2) (the scale of
BGR-mean-subtracted pixels, dataset.py:15-18,177) and pixel label maps made of background, 1-3 rectangles
of random foreground classes and a 4-pixel void (255) border.
"""
import numpy as np
import torch


def synthetic_labels(rng, B, h, w, category_num, max_objects=3):
    labels = np.zeros((B, h, w, 1), np.float32)
    lo_h, lo_w = max(h // 5, 2), max(w // 5, 2)
    for b in range(B):
        for _ in range(int(rng.integers(1, max_objects + 1))):
            c = int(rng.integers(1, category_num))
            hh = int(rng.integers(lo_h, max(h // 2, lo_h + 1)))
            ww = int(rng.integers(lo_w, max(w // 2, lo_w + 1)))
            y = int(rng.integers(0, h - hh + 1))
            x = int(rng.integers(0, w - ww + 1))
            labels[b, y:y + hh, x:x + ww, 0] = c
        brd = min(4, min(h, w) // 16)
        if brd:
            labels[b, :brd] = 255
            labels[b, -brd:] = 255
            labels[b, :, :brd] = 255
            labels[b, :, -brd:] = 255
    return labels


class SyntheticData:
    def __init__(self, config):
        self.config = config
        self.h, self.w = config.get("input_size", (321, 321))
        self.category_num = config.get("category_num", 21)
        self.length = int(config.get("length", 10582))  # size of VOC12 train_aug
        self.seed = int(config.get("seed", 0))
        self.device = torch.device(config.get("device", "cuda"))
        self.unique_batches = int(config.get("unique_batches", 2))

    def get_data_len(self, category="train"):
        return self.length

    def make_batch(self, batch_size, index=0):
        rng = np.random.default_rng(self.seed * 7919 + index + 1)
        gen = torch.Generator(device=self.device)
        gen.manual_seed(self.seed * 104729 + index)
        images = torch.randn((batch_size, self.h, self.w, 3), generator=gen, device=self.device) * 50.0
        labels = torch.as_tensor(synthetic_labels(rng, batch_size, self.h, self.w, self.category_num)).to(self.device)
        return images, labels

    def next_batch(self, category="train", batch_size=None, epoches=-1):
        """Endless iterator over (images f32 [B,H,W,3], labels f32 [B,H,W,1]), resident on the device."""
        batch_size = batch_size or self.config.get("batch_size", 1)
        pool = [self.make_batch(batch_size, i) for i in range(self.unique_batches)]
        i = 0
        while True:
            yield pool[i % len(pool)]
            i += 1


def image_preprocess(img_u8, label_u8, out_hw, scale=None, flip=False, device="cuda"):
    """dataset.image_preprocess (dataset.py:147-185) for ONE decoded sample, on the device.

    img_u8: uint8 [h0,w0,3] RGB, label_u8: uint8 [h0,w0] (or [h0,w0,1]); ``scale`` is the random factor the
    reference draws from U[0.75,1.25) (None = no random scaling, the "val" branch), ``flip`` its coin toss.
    Returns (image f32 [H,W,3] BGR minus mean, label f32 [H,W,1]) like the reference's tensors.  File listing and
    JPEG/PNG decoding stay on the host (out of the hot path)."""
    from . import ops
    dev = torch.device(device)
    img = torch.as_tensor(np.ascontiguousarray(img_u8)).to(dev, torch.float32).unsqueeze(0)
    lab = torch.as_tensor(np.ascontiguousarray(label_u8)).to(dev, torch.float32).reshape(1, img.shape[1], img.shape[2])
    if scale is not None:
        # tf.to_int32(tf.to_float(h) * scale): fp32 product, truncation (dataset.py:154-155)
        h_new = int(np.float32(img.shape[1]) * np.float32(scale))
        w_new = int(np.float32(img.shape[2]) * np.float32(scale))
        img = ops.resize_bilinear(img, h_new, w_new)
        lab = ops.resize_nearest(lab, h_new, w_new)
    img = ops.resize_bilinear(img.contiguous(), out_hw[0], out_hw[1])
    lab = ops.resize_nearest(lab.contiguous(), out_hw[0], out_hw[1])
    fl = torch.tensor([1 if flip else 0], dtype=torch.uint8, device=dev)
    img, lab = ops.bgr_mean_flip(img, lab, fl)
    return img[0], lab[0].unsqueeze(-1)
