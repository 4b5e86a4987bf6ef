#!/usr/bin/env python3
"""Throughput of the input pipeline (the reference's dataset.py:107-185) on a generated VOC-style tree.

    python scripts/loader_bench.py [--n 512] [--batch 16] [--workers 0] [--steps 30] [--out profiles/r06_loader.txt]

What it measures, each as images/s:
  1. host decode alone (PIL JPEG 500x375 + PNG label, the thread pool of data.dataset, GIL released inside libjpeg / zlib);
  2. `data.dataset.next_batch("train", B)` alone: decode (prefetched a batch ahead) + H2D + the HIP kernels for random scale /
     bilinear + nearest resize / BGR - mean / flip;
  3. the training step fed by it (micro_step + apply_update, fp32 and bf16), beside the same step on resident synthetic batches --
     the loader's decode runs on its threads one batch ahead, its device work on the step's stream.
The tree is generated under a temporary directory: low-frequency noise JPEGs at quality 90 (natural-image-like coefficient counts;
pure white noise would decode ~2x slower than photographs), palette PNG labels with a few rectangles.
"""
import argparse
import os
import shutil
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def make_tree(root, n, classes=21, seed=0):
    from PIL import Image
    rng = np.random.default_rng(seed)
    os.makedirs(os.path.join(root, "VOC2012", "JPEGImages"))
    os.makedirs(os.path.join(root, "VOC2012", "SegmentationClassAug"))
    os.makedirs(os.path.join(root, "txt"))
    ids = []
    palette = []
    from em_adapt_tensorflow_amd.data import VOC_COLORS
    for c in VOC_COLORS:
        palette += list(c)
    palette += [0] * (768 - len(palette))
    for i in range(n):
        w, h = (500, 375) if i % 4 else (375, 500)
        low = rng.integers(0, 256, size=(h // 16 + 2, w // 16 + 2, 3), dtype=np.uint8)
        img = Image.fromarray(low).resize((w, h), Image.BICUBIC)
        tex = rng.normal(0, 6, size=(h, w, 3))
        img = Image.fromarray(np.clip(np.asarray(img, np.float32) + tex, 0, 255).astype(np.uint8))
        lab = np.zeros((h, w), np.uint8)
        for _ in range(int(rng.integers(1, 4))):
            c = int(rng.integers(1, classes))
            hh, ww = int(rng.integers(h // 5, h // 2)), int(rng.integers(w // 5, w // 2))
            y, x = int(rng.integers(0, h - hh)), int(rng.integers(0, w - ww))
            lab[y:y + hh, x:x + ww] = c
        lab[:4], lab[-4:], lab[:, :4], lab[:, -4:] = 255, 255, 255, 255
        name = "2007_%06d" % i
        img.save(os.path.join(root, "VOC2012", "JPEGImages", name + ".jpg"), quality=90)
        pl = Image.fromarray(lab, mode="P")
        pl.putpalette(palette)
        pl.save(os.path.join(root, "VOC2012", "SegmentationClassAug", name + ".png"))
        ids.append(name)
    for cat in ("train", "val"):
        with open(os.path.join(root, "txt", cat + ".txt"), "w") as f:
            f.write("\n".join(ids) + "\n")
    return ids


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=512)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--workers", type=int, default=0, help="decode threads (0 = the host cores this process may use)")
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--step-workers", type=int, default=0, help="decode processes beside the training step (0 = min(12, cores))")
    ap.add_argument("--out", default="")
    ap.add_argument("--no-steps", action="store_true", help="loader alone (no training step)")
    args = ap.parse_args()
    import concurrent.futures
    import torch
    import bench
    from em_adapt_tensorflow_amd import data as D
    from em_adapt_tensorflow_amd.deeplab import ADAPT

    cores = bench.host_cores()
    workers = args.workers or cores
    lines = []

    def say(s):
        print(s, flush=True)
        lines.append(s)

    tmp = tempfile.mkdtemp(prefix="emadapt_voc_")
    try:
        t0 = time.perf_counter()
        make_tree(tmp, args.n)
        jpg = sum(os.path.getsize(os.path.join(tmp, "VOC2012", "JPEGImages", f)) for f in os.listdir(os.path.join(tmp, "VOC2012", "JPEGImages")))
        say("# input pipeline on a generated VOC-style tree: %d samples (500x375 JPEG q90, mean %.0f KB; palette PNG labels), built in %.1f s" % (
            args.n, jpg / args.n / 1e3, time.perf_counter() - t0))
        say("# host: %d cores usable (%s); batch %d; B x 321 x 321 outputs" % (cores, bench.cpu_model(), args.batch))
        cfg = {"input_size": (321, 321), "main_path": os.path.join(tmp, "VOC2012"), "list_path": os.path.join(tmp, "txt"), "device": "cuda",
               "seed": 5, "batch_size": args.batch}
        # 1. decode alone
        for w in sorted({1, 4, workers}):
            ds = D.dataset(dict(cfg, decode_workers=w))
            f = ds.data_f["train"]
            with concurrent.futures.ThreadPoolExecutor(w) as pool:
                list(pool.map(lambda i: ds._decode(f["img"][i], f["label"][i]), range(min(64, args.n))))   # page cache + pool warm-up
                t0 = time.perf_counter()
                list(pool.map(lambda i: ds._decode(f["img"][i], f["label"][i]), range(args.n)))
                dt = time.perf_counter() - t0
            say("decode alone      workers %2d: %8.1f images/s (%.2f ms per image per thread)" % (w, args.n / dt, dt / args.n * w * 1e3))
        # 2. next_batch alone
        for mode, w in [("thread", 4), ("thread", workers), ("process", 4), ("process", min(8, workers)), ("process", workers)]:
            ds = D.dataset(dict(cfg, decode_workers=w, decode_mode=mode))
            it = ds.next_batch("train", args.batch)
            for _ in range(3):
                next(it)
            torch.cuda.synchronize()
            nb = max(args.n // args.batch, 8)
            t0 = time.perf_counter()
            for _ in range(nb):
                x, y, _ids = next(it)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            say("next_batch alone  %-7s x %2d: %8.1f images/s (%.2f ms per batch of %d)" % (mode, w, nb * args.batch / dt, dt / nb * 1e3, args.batch))
            it.close()
        if args.no_steps:
            return
        # 3. the step fed by the loader, beside the same step on resident batches
        for dtype in ("fp32", "bf16"):
            mcfg = {"input_size": (321, 321), "category_num": 21, "accum_num": 1, "batch_size": args.batch, "seed": 2, "device": "cuda",
                    "compute_dtype": dtype, "fused_update": True}
            syn = D.SyntheticData(dict(mcfg, seed=0, unique_batches=2))
            a = ADAPT(dict(mcfg, data=syn))
            a.build()
            a.pre_train(base_lr=1e-3, weight_decay=1e-5, momentum=0.9, batch_size=args.batch)
            sit = syn.next_batch("train", args.batch)

            def run(source, n):
                for _ in range(3):
                    b = next(source)
                    a.micro_step(b[0], b[1], keep_prob=0.5)
                    a.apply_update()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(n):
                    b = next(source)
                    a.micro_step(b[0], b[1], keep_prob=0.5)
                    a.apply_update()
                torch.cuda.synchronize()
                return (time.perf_counter() - t0) / n

            t_syn = run(sit, args.steps)
            ds = D.dataset(dict(cfg, decode_workers=args.step_workers or min(12, workers), decode_mode="process"))
            lit = ds.next_batch("train", args.batch)
            t_load = run(lit, args.steps)
            lit.close()
            say("step %-5s resident synthetic batches: %7.2f ms = %7.1f images/s | fed by next_batch (%d decode processes): %7.2f ms = %7.1f images/s (%.0f %%)" % (
                dtype, t_syn * 1e3, args.batch / t_syn, args.step_workers or min(12, workers), t_load * 1e3, args.batch / t_load, 100.0 * t_syn / t_load))
            del a
            torch.cuda.empty_cache()
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
        if args.out:
            with open(args.out, "w") as f:
                f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
