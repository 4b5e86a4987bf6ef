#!/usr/bin/env python3
"""Generate tests/golden/estep_*.npz by running the REFERENCE's own estep.py.

Runs only in the build container (needs /root/reference).  Nothing of the reference is
copied: we import /root/reference/estep.py in place (module ``estep``; it dlopens
``<cwd>/estep/libweaklabel.so`` at import, estep.py:9-14, which ``make -C oracle ref``
builds from the reference's weaklabel.cpp into oracle/_ref/estep/), call
``estep.estep(..., use_c=False)`` -- the path deeplab.py:114 uses -- on seeded inputs and
store inputs + outputs.  The hidden shuffle order is pinned with ``random.seed(s)`` right
before the call and recovered independently by ``oracle.estep_oracle.draw_perms(s)``.

Each fixture holds: logits (f32 or f16-exact), label f32 [B,H,W], perms i32
[num_iter,C-1], params, label_map u8 [B,H,W] (= argmax axis 3, first max, of the
reference's float map, i.e. deeplab.py:123), and for the small cases the float map itself.

Usage:  make -C oracle ref && python oracle/make_goldens.py
"""
import os
import random
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get("EMADAPT_REFERENCE", "/root/reference")
OUT = os.path.join(ROOT, "tests", "golden")


def import_reference_estep():
    scratch = os.path.join(HERE, "_ref")
    so = os.path.join(scratch, "estep", "libweaklabel.so")
    if not os.path.exists(so):
        raise SystemExit("run `make -C oracle ref` first (needs %s)" % so)
    sys.dont_write_bytecode = True
    os.chdir(scratch)  # estep.py resolves the .so relative to cwd
    sys.path.insert(0, REF)
    import estep  # noqa: the reference module /root/reference/estep.py

    assert os.path.realpath(estep.__file__) == os.path.realpath(os.path.join(REF, "estep.py")), estep.__file__
    return estep


def blobs_label(rng, B, S, C, classes_per_image, bg=True, void_border=0):
    """Pixel label maps made of axis-aligned rectangles, like a shrunk VOC ground truth."""
    lab = np.zeros((B, S, S), dtype=np.float32) if bg else None
    out = np.zeros((B, S, S), dtype=np.float32)
    for b in range(B):
        cls = classes_per_image[b]
        if bg:
            out[b] = 0
        else:
            out[b] = 255
        for c in cls:
            h = int(rng.integers(max(2, S // 6), max(3, S // 2)))
            w = int(rng.integers(max(2, S // 6), max(3, S // 2)))
            y = int(rng.integers(0, S - h + 1))
            x = int(rng.integers(0, S - w + 1))
            out[b, y:y + h, x:x + w] = c
        if void_border:
            out[b, :void_border] = 255
            out[b, -void_border:] = 255
            out[b, :, :void_border] = 255
            out[b, :, -void_border:] = 255
    del lab
    return out


def run_case(estep_mod, name, logits, label, seed, store_float=True, f16=False, **params):
    p = dict(suppress_others=True, num_iter=5, margin_others=1e-5, bg_p=0.4, fg_p=0.2)
    p.update(params)
    logits = np.ascontiguousarray(logits, dtype=np.float32)
    if f16:
        logits = logits.astype(np.float16).astype(np.float32)
    B, H, W, C = logits.shape
    random.seed(seed)
    fm = estep_mod.estep(logits.copy(), np.array(label, dtype=np.float32, copy=True),
                         p["suppress_others"], p["num_iter"], p["margin_others"], p["bg_p"], p["fg_p"],
                         use_c=False)
    assert fm.dtype == np.float32 and fm.shape == logits.shape
    label_map = np.argmax(fm, axis=3)
    r = random.Random(seed)
    perms = np.zeros((p["num_iter"], C - 1), dtype=np.int32)
    for it in range(p["num_iter"]):
        t = list(range(1, C))
        r.shuffle(t)
        perms[it] = t
    rec = dict(
        logits=logits.astype(np.float16) if f16 else logits,
        label=np.asarray(label, dtype=np.float32),
        perms=perms,
        seed=np.int64(seed),
        suppress_others=np.bool_(p["suppress_others"]),
        num_iter=np.int32(p["num_iter"]),
        margin_others=np.float64(p["margin_others"]),
        bg_p=np.float64(p["bg_p"]),
        fg_p=np.float64(p["fg_p"]),
        label_map=label_map.astype(np.uint8),
    )
    if store_float:
        rec["float_map"] = fm
    path = os.path.join(OUT, "estep_%s.npz" % name)
    np.savez_compressed(path, **rec)
    hist = np.bincount(label_map.reshape(-1), minlength=C)
    print("%-22s B=%d %dx%dx%d  %7.1f KB  labels used: %s" % (
        name, B, H, W, C, os.path.getsize(path) / 1024, np.nonzero(hist)[0].tolist()))


def main():
    est = import_reference_estep()
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.default_rng(20261003)

    # --- hand inputs the reference keeps (unanswered) in comments: estep/estep.py:94-97
    f = np.array([[[[1, 2, 3]], [[3, 2, 1]]], [[[4, 5, 6]], [[7, 8, 9]]]], dtype=np.float32)
    l = np.array([[[1], [2]], [[1], [0]]], dtype=np.float32)
    run_case(est, "tiny_a", f, l, seed=1, bg_p=0.5, fg_p=0.25)
    f = np.array([[[[1, 2, 3]], [[3, 2, 1]]]], dtype=np.float32)
    l = np.array([[[1], [1]]], dtype=np.float32)
    run_case(est, "tiny_b", f, l, seed=2, bg_p=0.5, fg_p=0.25)

    S, C = 41, 21
    # B=1, bg + 1 fg
    lab = blobs_label(rng, 1, S, C, [[7]])
    run_case(est, "s41_b1_p2", rng.normal(0, 3, (1, S, S, C)), lab, seed=3)
    # B=2: P=3 and P=5, void border
    lab = blobs_label(rng, 2, S, C, [[3, 15], [1, 8, 12, 20]], void_border=1)
    run_case(est, "s41_b2_mixed", rng.normal(0, 2, (2, S, S, C)), lab, seed=4)
    # bg only
    run_case(est, "s41_bg_only", rng.normal(1, 2, (1, S, S, C)), np.zeros((1, S, S), np.float32), seed=5)
    # bg absent: only fg + void
    lab = blobs_label(rng, 1, S, C, [[5, 9]], bg=False)
    run_case(est, "s41_bg_absent", rng.normal(0, 2, (1, S, S, C)), lab, seed=6)
    # nothing valid (all void) next to a normal image: exercises the batch-global max
    lab = np.full((2, S, S), 255, np.float32)
    lab[1] = blobs_label(rng, 1, S, C, [[2]])[0]
    run_case(est, "s41_all_void", rng.normal(0, 2, (2, S, S, C)), lab, seed=7)
    # all-equal logits: every comparison is a tie
    lab = blobs_label(rng, 1, S, C, [[4, 6]])
    run_case(est, "s41_all_equal", np.full((1, S, S, C), 0.75, np.float32), lab, seed=8)
    # strictly negative logits: the batch max is negative (suppress step changes character)
    lab = blobs_label(rng, 2, S, C, [[10], [11, 13]])
    run_case(est, "s41_negative", -np.abs(rng.normal(4, 2, (2, S, S, C))) - 0.5, lab, seed=9)
    # heavily quantised logits: many ties in selection and argmax
    lab = blobs_label(rng, 2, S, C, [[1, 2], [18, 19, 20]], void_border=2)
    q = np.round(rng.normal(0, 1.5, (2, S, S, C)) * 2) / 2
    run_case(est, "s41_quantised", q, lab, seed=10)
    # parameter variants
    lab = blobs_label(rng, 1, S, C, [[3, 4]])
    x = rng.normal(0, 2, (1, S, S, C))
    run_case(est, "s41_nosuppress", x, lab, seed=11, suppress_others=False)
    run_case(est, "s41_iter1_margin", x, lab, seed=12, num_iter=1, margin_others=0.5, bg_p=0.5, fg_p=0.25)
    # tiny logits like a freshly initialised net (sigma=0.01 weights): values ~1e-4 around 0
    lab = blobs_label(rng, 2, S, C, [[6, 7], [14]])
    run_case(est, "s41_tiny_values", rng.normal(0, 1e-4, (2, S, S, C)), lab, seed=13)
    # the benchmark batch: B=16, P in 1..5 (label map only; logits exactly representable in f16)
    cls = [[], [3], [5, 9], [1, 2, 3], [4, 8, 12, 16], [20], [7, 14], [2, 19], [11], [6, 10, 15],
           [13, 17], [18], [1, 20], [9], [3, 6, 9, 12], [5]]
    lab = blobs_label(rng, 16, S, C, cls, void_border=1)
    run_case(est, "s41_b16", rng.normal(0, 2.5, (16, S, S, C)), lab, seed=14, store_float=False, f16=True)

    # 513x513 / 41-class configuration: 65x65x41.  Image 1 has 10 classes present (more present
    # channels than fit the GPU kernel's LDS-resident fast path).
    S, C = 65, 41
    lab = blobs_label(rng, 2, S, C, [[5, 17, 33, 40], [1, 2, 3, 4, 5, 6, 7, 8, 9]], void_border=1)
    run_case(est, "s65_b2_c41", rng.normal(0, 2, (2, S, S, C)), lab, seed=15, store_float=False, f16=True)

    # odd geometry: non-square, C not 21/41, HW not a multiple of the workgroup size
    lab = blobs_label(rng, 3, 13, 5, [[1], [2, 3], [4]])[:, :9, :]
    run_case(est, "odd_9x13x5", rng.normal(0, 1, (3, 9, 13, 5)), lab, seed=16, bg_p=0.3, fg_p=0.1)


if __name__ == "__main__":
    main()
