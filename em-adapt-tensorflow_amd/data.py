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
      * JPEG/PNG decoding runs in ``decode_workers`` worker PROCESSES (config ``decode_mode``: "process" | "thread"),
        ``prefetch_batches`` (2) batches ahead of the consumer, straight into a pinned staging ring; scaling, resizing, channel
        swap, mean subtraction and flip of a whole batch are ONE kernel on a side stream (``ops.voc_preprocess``:
        bit for bit the op-by-op ``image_preprocess`` below, which stays as the single-sample entry).
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
        category (0 for a fresh object); an integer starts there (resuming a run: ADAPT.train passes config "start_epoch").

        Per batch the consumer's thread does a handful of things -- collect the decoders' replies, draw the augmentation, queue ONE
        host-to-device copy and ONE kernel (ops.voc_preprocess) on a side stream -- while decode runs `prefetch_batches` batches
        ahead in worker processes (``_Pipeline``)."""
        category = category or self.default_category
        if start_epoch is None:
            start_epoch = self.info["next_epoch"].get(category, 0)
        batch_size = batch_size or self.config.get("batch_size", 1)
        files = self.data_f[category]
        train = category == "train"
        depth = max(int(self.config.get("prefetch_batches", 2)), 1)
        pipe = self._make_pipeline(batch_size, depth + 2)
        try:
            it = self.sample_order(category, batch_size, epoches, start_epoch)
            queue = []   # (epoch, position, indices, ticket) of the batches whose decode is under way, oldest first
            k = 0
            while True:
                while len(queue) <= depth:
                    nxt = next(it, None)
                    if nxt is None:
                        break
                    queue.append(nxt + (pipe.submit(k, [(files["img"][i], files["label"][i]) for i in nxt[2]]),))
                    k += 1
                if not queue:
                    break
                ep, s, idx, ticket = queue.pop(0)
                sizes = pipe.collect(ticket)
                samples = []
                for (h0, w0) in sizes:
                    # the draws of image_preprocess(), in its order: the scale, then the coin
                    scale = float(self.rng.uniform(0.75, 1.25)) if train else None
                    flip = bool(train) and bool(self.rng.uniform() < 0.5)
                    # tf.to_int32(tf.to_float(h) * scale): fp32 product, truncation (dataset.py:154-155)
                    h1 = int(np.float32(h0) * np.float32(scale)) if scale is not None else 0
                    w1 = int(np.float32(w0) * np.float32(scale)) if scale is not None else 0
                    samples.append((h0, w0, h1, w1, flip))
                x, y = pipe.preprocess(ticket, samples)
                self.info["epoch"][category] = ep
                self.info["index"][category] = s + len(idx) * self.world
                # a later iterator over this category starts behind the last epoch this one has drawn from
                self.info["next_epoch"][category] = ep + 1
                yield x, y, [files["id"][i] for i in idx]
        finally:
            pipe.close()


    def _make_pipeline(self, batch_size, slots):
        """The staging ring, decode workers and device preprocessing of one iterator (a seam for the host-logic tests)."""
        return _Pipeline(self.device, batch_size, (self.h, self.w), slots, max(self.workers, 1),
                         str(self.config.get("decode_mode", "process")), int(self.config.get("max_sample_pixels", 640 * 640)))


class _Pipeline:
    """Staging ring + decode workers + device preprocessing behind ``dataset.next_batch``.

    * a ring of `slots` batch slots in ONE host block; sample j of a slot owns 4 * max_pixels bytes (uint8 RGB, then the uint8 class
      map).  decode_mode "process" (default): the block is POSIX shared memory, pinned with emadapt_host_register, and the decoders are
      `workers` processes running ``_decode`` (each with its own interpreter lock: PIL's per-sample Python work neither serialises
      them nor delays the training step's launch code); "thread": a pinned torch tensor and a thread pool (no child processes).
    * per batch, on a side stream: one copy of the slot to its device twin, one ``ops.voc_preprocess`` launch into the slot's output
      buffers; the consumer's stream waits for that, takes a copy (the tensors it is handed are its own) and marks the slot free.
    Nothing on the consumer's stream waits for a decode or a copy that is not its own batch's."""

    def __init__(self, device, batch, out_hw, slots, workers, mode, max_pixels):
        import os
        import torch
        from . import _lib
        self.device = torch.device(device)
        self.batch, self.out_hw, self.slots, self.mode = int(batch), out_hw, int(slots), mode
        self.max_pixels = int(max_pixels)
        self.cap = 4 * self.max_pixels
        self.slot_bytes = self.batch * self.cap
        total = self.slots * self.slot_bytes
        self._lib = _lib
        self._registered = None
        self._shm = None
        self._procs = []
        self._pool = None
        self._closed = False
        if mode == "process":
            import subprocess
            import sys
            from multiprocessing import shared_memory
            self._shm = shared_memory.SharedMemory(create=True, size=total)
            self._host = np.ndarray((total,), dtype=np.uint8, buffer=self._shm.buf)
            self._host_ptr = self._host.ctypes.data
            try:   # pinned: the batch's copy is then one asynchronous DMA (unpinned it is staged by the runtime: slower, still right)
                _lib.call("emadapt_host_register", ctypes_void_p(self._host_ptr), total)
                self._registered = self._host_ptr
            except _lib.EmadaptError:
                self._registered = None
            root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
            env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""), OMP_NUM_THREADS="1")
            for _ in range(workers):
                self._procs.append(subprocess.Popen([sys.executable, "-m", "em_adapt_tensorflow_amd._decode", self._shm.name], env=env,
                                                    stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True, bufsize=1))
        elif mode == "thread":
            import concurrent.futures
            self._pinned = torch.empty(total, dtype=torch.uint8).pin_memory()
            self._host = self._pinned.numpy()
            self._host_ptr = self._pinned.data_ptr()
            self._pool = concurrent.futures.ThreadPoolExecutor(workers)
        else:
            raise ValueError("decode_mode must be 'process' or 'thread', got %r" % (mode,))
        self._dev = torch.empty((self.slots, self.slot_bytes), dtype=torch.uint8, device=self.device)
        H, W = out_hw
        self._img = torch.empty((self.slots, self.batch, H, W, 3), dtype=torch.float32, device=self.device)
        self._lab = torch.empty((self.slots, self.batch, H, W, 1), dtype=torch.float32, device=self.device)
        self._side = torch.cuda.Stream(device=self.device)
        self._copied = [None] * self.slots     # event on the side stream: the slot's host bytes have left for the device
        self._consumed = [None] * self.slots   # event on the consumer's stream: its copy of the slot's outputs has been taken

    def submit(self, k, paths):
        """Start decoding batch k (a list of (image path, label path)) into slot k % slots.  Returns the ticket collect() takes."""
        slot = k % self.slots
        if self._copied[slot] is not None:
            self._copied[slot].synchronize()   # (the copy of what the slot held `slots` batches ago: long done)
        base = slot * self.slot_bytes
        jobs = []
        for j, (img_f, lab_f) in enumerate(paths):
            off = base + j * self.cap
            if self._pool is not None:
                from . import _decode
                jobs.append(self._pool.submit(_decode.decode_into, self._host, off, off + 3 * self.max_pixels, self.cap, img_f, lab_f))
            else:
                p = self._procs[j % len(self._procs)]
                if "\t" in img_f or "\n" in img_f or "\t" in lab_f or "\n" in lab_f:
                    raise ValueError("file names with tabs or newlines are not supported: %r" % (img_f,))
                p.stdin.write("%d\t%d\t%d\t%s\t%s\n" % (off, off + 3 * self.max_pixels, self.cap, img_f, lab_f))
                jobs.append((p, img_f))
        if self._pool is None:
            for p in self._procs:
                p.stdin.flush()
        return (slot, jobs)

    def collect(self, ticket):
        """Wait for the batch's decoders; returns [(h0, w0)] in sample order."""
        _, jobs = ticket
        sizes = []
        for job in jobs:
            if self._pool is not None:
                sizes.append(job.result())
                continue
            p, name = job
            line = p.stdout.readline()
            if not line:
                raise RuntimeError("a decode worker ended (exit code %s) while decoding %s" % (p.poll(), name))
            if not line.startswith("OK "):
                raise RuntimeError("decoding %s failed: %s" % (name, line.strip()[4:]))
            _, h, w = line.split()
            sizes.append((int(h), int(w)))
        return sizes

    def preprocess(self, ticket, samples):
        """samples: [(h0, w0, h1, w1, flip)] of the batch in the ticket's slot.  Returns (images, labels), the consumer's own tensors."""
        import torch
        from . import ops
        slot, jobs = ticket
        n = len(samples)
        base = slot * self.slot_bytes
        main = torch.cuda.current_stream(self.device)
        with torch.cuda.stream(self._side):
            if self._consumed[slot] is not None:
                self._side.wait_event(self._consumed[slot])   # the outputs this launch overwrites have been copied out
            self._lib.call("emadapt_memcpy_h2d_async", ctypes_void_p(self._dev[slot].data_ptr()), ctypes_void_p(self._host_ptr + base),
                           n * self.cap, ctypes_void_p(self._side.cuda_stream))
            ev = torch.cuda.Event()
            ev.record(self._side)
            self._copied[slot] = ev
            descs = [(j * self.cap, j * self.cap + 3 * self.max_pixels, h0, w0, h1, w1, 1 if flip else 0)
                     for j, (h0, w0, h1, w1, flip) in enumerate(samples)]
            ops.voc_preprocess(self._dev[slot], descs, self.out_hw[0], self.out_hw[1], img_out=self._img[slot][:n], label_out=self._lab[slot][:n])
            done = torch.cuda.Event()
            done.record(self._side)
        main.wait_event(done)
        x, y = self._img[slot][:n].clone(), self._lab[slot][:n].clone()
        took = torch.cuda.Event()
        took.record(main)
        self._consumed[slot] = took
        return x, y

    def close(self):
        if self._closed:
            return
        self._closed = True
        import torch
        try:
            self._side.synchronize()
        except Exception:
            pass
        for p in self._procs:
            try:
                p.stdin.close()
            except Exception:
                pass
        for p in self._procs:
            try:
                p.wait(timeout=5)
            except Exception:
                p.kill()
                p.wait()
            try:
                p.stdout.close()
            except Exception:
                pass
        if self._pool is not None:
            self._pool.shutdown(wait=True, cancel_futures=True)
        if self._registered is not None:
            try:
                self._lib.call("emadapt_host_unregister", ctypes_void_p(self._registered))
            except Exception:
                pass
        if self._shm is not None:
            self._host = None
            try:
                self._shm.close()
                self._shm.unlink()
            except Exception:
                pass
        del torch

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def ctypes_void_p(v):
    import ctypes
    return ctypes.c_void_p(int(v))
