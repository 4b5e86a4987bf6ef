"""Input sources with the call shape the training loop expects from the reference's ``dataset``
(dataset.py:107-145: ``next_batch(category, batch_size, epoches)`` and ``get_data_len()``).

``dataset`` (below) is the reference's class surface over real VOC-style files: host decode, device
scale/resize/BGR-mean/flip.  ``SyntheticData``
produces the benchmark inputs of SURVEY section 8(d) directly in HBM: images ~ N(0, 50^2) (the scale of
BGR-mean-subtracted pixels, dataset.py:15-18,177) and pixel label maps made of background, 1-3 rectangles
of random foreground classes and a 4-pixel void (255) border.
"""
import numpy as np
import torch


def dist_rank_world(config=None):
    """(rank, world_size) of this process: config["rank"] / config["world_size"] when given, else torch.distributed's
    (0, 1 for a single process).  Data parallelism maps micro-batches to ranks (dist.py), so every source below hands
    rank r the r-th slice of each global batch and seeds its augmentation / synthetic draws per rank."""
    config = config or {}
    if "rank" in config or "world_size" in config:
        return int(config.get("rank", 0)), int(config.get("world_size", 1))
    d = torch.distributed
    if d.is_available() and d.is_initialized():
        return d.get_rank(), d.get_world_size()
    return 0, 1


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
        self.rank, self.world = dist_rank_world(config)
        self.seed = int(config.get("seed", 0)) * self.world + self.rank   # every rank draws its own batches
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

    def next_batch(self, category="train", batch_size=None, epoches=-1, start_epoch=None):
        """Endless iterator over (images f32 [B,H,W,3], labels f32 [B,H,W,1]), resident on the device (start_epoch: accepted for
        the dataset class's signature; a synthetic source has no epochs)."""
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
    img = torch.as_tensor(np.array(img_u8, copy=True)).to(dev, torch.float32).unsqueeze(0)
    lab = torch.as_tensor(np.array(label_u8, copy=True)).to(dev, torch.float32).reshape(1, img.shape[1], img.shape[2])
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


# VOC palette of the reference's label2rgb / rgb2label (dataset.py:78-84, 91-97)
VOC_COLORS = [(0, 0, 0), (128, 0, 0), (0, 128, 0), (128, 128, 0), (0, 0, 128), (128, 0, 128), (0, 128, 128),
              (128, 128, 128), (64, 0, 0), (192, 0, 0), (64, 128, 0), (192, 128, 0), (64, 0, 128), (192, 0, 128),
              (64, 128, 128), (192, 128, 128), (0, 64, 0), (128, 64, 0), (0, 192, 0), (128, 192, 0), (0, 64, 128)]


class dataset:
    """The reference's ``dataset`` class (dataset.py:6-145) with the same config keys, attributes and methods, feeding
    the device pipeline instead of tf.data:

      * file lists: ``<list_path>/<category>.txt`` (default ``pascal/txt``, dataset.py:32) holding one id per line;
        images ``<main_path>/JPEGImages/<id>.jpg``, labels ``<main_path>/SegmentationClassAug/<id>.png``
        (dataset.py:35-37); ``config["length"]`` truncates every list (dataset.py:38-42);
      * ``next_batch(category, batch_size, epoches)`` (dataset.py:107-145) is an iterator over
        ``(images f32 [B,H,W,3] BGR-mean, labels f32 [B,H,W,1], ids)`` device tensors: shuffled every epoch, ``epoches``
        passes (-1 = endless), "train" samples get the random scale U[0.75,1.25) and the coin-toss flip
        (dataset.py:125-126), other categories neither; a short final batch is yielded as is (tf.data ``batch``);
      * JPEG/PNG decoding runs in a small host thread pool one batch ahead of the consumer; scaling, resizing, channel
        swap, mean subtraction and flip are the HIP kernels behind ``image_preprocess``.
    ``ADAPT.train`` accepts it as ``config["data"]`` (it ignores the third element)."""

    def __init__(self, config={}):
        import os
        self.config = config
        self.w, self.h = self.config.get("input_size", (240, 240))   # (w, h) naming as in dataset.py:9; square inputs
        self.categorys = self.config.get("categorys", ["train", "val"])
        assert len(self.categorys) > 0, "no enough categorys in dataset"
        self.main_path = self.config.get("main_path", os.path.join("pascal", "VOCdevkit", "VOC2012"))
        self.list_path = self.config.get("list_path", os.path.join("pascal", "txt"))
        self.ignore_label = self.config.get("ignore_label", 255)
        self.default_category = self.config.get("default_category", self.categorys[0])
        self.device = self.config.get("device", "cuda")
        self.workers = int(self.config.get("decode_workers", 4))
        # two generators: the epoch permutations must be the SAME on every rank of a data-parallel job (each rank then
        # takes its own slice of every global batch), the augmentation draws must differ
        self.rank, self.world = dist_rank_world(self.config)
        seed = self.config.get("seed", None)
        if seed is None and self.world > 1:
            box = [int(np.random.SeedSequence().entropy % (1 << 31))]
            torch.distributed.broadcast_object_list(box, src=0)
            seed = box[0]
        if seed is None:
            seed = int(np.random.SeedSequence().entropy % (1 << 31))
        # the permutation of (category, epoch) is a pure function of the seed, so no call order on one rank (a rank-0-only
        # pass over "val", say) can shift the shards of the ranks against each other
        self.perm_seed = int(seed)
        self.rng = np.random.default_rng([int(seed), self.rank])
        self.img_mean = np.ones((self.w, self.h, 3))
        self.img_mean[:, :, 0] *= 104.00698793
        self.img_mean[:, :, 1] *= 116.66876762
        self.img_mean[:, :, 2] *= 122.67891434
        self.init()

    def init(self):
        self.data_f, self.data_len = self.get_data_f()
        self.reset_info()

    def get_data_f(self):
        import os
        data_f, data_len = {}, {}
        for one in self.categorys:
            data_f[one] = {"img": [], "label": [], "id": []}
            with open(os.path.join(self.list_path, "%s.txt" % one), "r") as f:
                for line in f.readlines():
                    line = line.strip("\n")
                    if not line:
                        continue
                    data_f[one]["id"].append(line)
                    data_f[one]["img"].append(os.path.join(self.main_path, "JPEGImages", "%s.jpg" % line))
                    data_f[one]["label"].append(os.path.join(self.main_path, "SegmentationClassAug", "%s.png" % line))
            if "length" in self.config:
                n = self.config["length"]
                for k in ("id", "img", "label"):
                    data_f[one][k] = data_f[one][k][:n]
            data_len[one] = len(data_f[one]["label"])
        print("len:%s" % str(data_len))
        return data_f, data_len

    def reset_info(self):
        self.info = {"epoch": {}, "index": {}, "perm": {}, "next_epoch": {}}
        for category in self.categorys:
            self.info["epoch"][category] = 0
            self.info["index"][category] = 0
            self.info["perm"][category] = self.epoch_perm(category, 0)
        return self.info

    def epoch_perm(self, category, epoch):
        """The sample order of `epoch` over `category`: identical on every rank, independent of what was drawn before."""
        rng = np.random.default_rng([self.perm_seed, self.categorys.index(category), int(epoch)])
        return rng.permutation(self.data_len[category])

    def get_info(self, key="epoch", category=None):
        return self.info[key][category or self.default_category]

    def get_data_len(self, category=None):
        return self.data_len[category or self.default_category]

    def get_cur_epoch(self, category=None):
        return self.info["epoch"][category or self.default_category]

    @staticmethod
    def label2rgb(label, colors=(), ignore_label=255, ignore_color=(255, 255, 255)):
        colors = list(colors) or VOC_COLORS
        label = np.asarray(label)
        out = np.empty(label.shape[:2] + (3,), np.uint8)
        out[...] = ignore_color
        for i, c in enumerate(colors):
            out[label[..., 0] == i if label.ndim == 3 else label == i] = c
        return out

    @staticmethod
    def rgb2label(label, colors=(), ignore_color=255):
        colors = list(colors) or VOC_COLORS
        label = np.asarray(label)
        out = ignore_color * np.ones(label.shape[0:2], dtype=np.uint8)
        for i, c in enumerate(colors):
            out[np.all(label[:, :, 0:3] == np.asarray(c, label.dtype), axis=2)] = i
        return out

    # -- decoding (host) and preprocessing (device) ----------------------------------------------------------------
    @staticmethod
    def _decode(img_f, label_f):
        from PIL import Image
        img = np.asarray(Image.open(img_f).convert("RGB"), dtype=np.uint8)
        lab = Image.open(label_f)
        lab = np.asarray(lab if lab.mode in ("L", "P") else lab.convert("L"), dtype=np.uint8)  # palette index = class id
        return img, lab

    def image_preprocess(self, img, label, random_scale=False, flip=False):
        """dataset.image_preprocess (dataset.py:147-185) for one decoded sample, on the device."""
        scale = float(self.rng.uniform(0.75, 1.25)) if random_scale else None
        do_flip = bool(flip) and bool(self.rng.uniform() < 0.5)
        return image_preprocess(img, label, (self.h, self.w), scale=scale, flip=do_flip, device=self.device)

    def sample_order(self, category=None, batch_size=None, epoches=-1, start_epoch=0):
        """Iterator over (epoch, position, sample indices) of this rank's batches, `epoches` epochs from `start_epoch` on (the
        order of an epoch is a function of (seed, category, epoch) alone, so a resumed run that passes the epoch it stopped in --
        config "start_epoch", or a second train() call on the same object, which continues behind the epochs already handed out
        -- does not replay epoch 0, 1, ...: advisor finding, round 3).  One permutation per epoch, shared
        by all ranks (``epoch_perm``: a function of seed, category and epoch); a global batch is ``batch_size * world`` consecutive entries of it and rank r
        takes entries [r*batch_size, (r+1)*batch_size) -- ranks never see the same sample within an epoch.  With one
        rank a short final batch is yielded as is (tf.data ``batch``); with several ranks the ragged remainder of an
        epoch is dropped so that every rank always holds the same number of samples (the gradient all-reduce averages
        equal-sized shards, dist.py)."""
        category = category or self.default_category
        batch_size = batch_size or self.config.get("batch_size", 1)
        n = self.data_len[category]
        gb = batch_size * self.world
        if n == 0 or (self.world > 1 and n < gb):
            raise ValueError("category %r holds %d samples: fewer than one global batch of %d x %d ranks" % (
                category, n, batch_size, self.world))
        ep = int(start_epoch)
        last = ep + epoches
        while epoches < 0 or ep < last:
            perm = self.epoch_perm(category, ep)
            self.info["perm"][category] = perm
            stop = n if self.world == 1 else (n // gb) * gb
            for s in range(0, stop, gb):
                yield ep, s, perm[s + self.rank * batch_size:s + (self.rank + 1) * batch_size]
            ep += 1

    def next_batch(self, category=None, batch_size=None, epoches=-1, start_epoch=None):
        """dataset.next_batch.  start_epoch = None continues behind the epochs this object has already handed out for the
        category (0 for a fresh object); an integer starts there (resuming a run: ADAPT.train passes config "start_epoch")."""
        import concurrent.futures
        category = category or self.default_category
        if start_epoch is None:
            start_epoch = self.info["next_epoch"].get(category, 0)
        batch_size = batch_size or self.config.get("batch_size", 1)
        files = self.data_f[category]
        train = category == "train"

        with concurrent.futures.ThreadPoolExecutor(max(self.workers, 1)) as pool:
            def submit(idx):
                return [pool.submit(self._decode, files["img"][i], files["label"][i]) for i in idx]

            it = self.sample_order(category, batch_size, epoches, start_epoch)
            nxt = next(it, None)
            pending = submit(nxt[2]) if nxt else None
            while nxt is not None:
                ep, s, idx = nxt
                futs = pending
                nxt = next(it, None)
                pending = submit(nxt[2]) if nxt else None   # decode the next batch while this one is consumed
                imgs, labs = [], []
                for f in futs:
                    img, lab = f.result()
                    x, y = self.image_preprocess(img, lab, random_scale=train, flip=train)
                    imgs.append(x)
                    labs.append(y)
                self.info["epoch"][category] = ep
                self.info["index"][category] = s + len(idx) * self.world
                # a later iterator over this category starts behind the last epoch this one has drawn from
                self.info["next_epoch"][category] = ep + 1
                yield torch.stack(imgs), torch.stack(labs), [files["id"][i] for i in idx]
