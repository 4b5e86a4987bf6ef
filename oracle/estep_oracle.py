"""CPU oracle for the EM-Adapt E-step.  TEST INFRASTRUCTURE ONLY.

This is a NumPy restatement of the algorithm the reference runs in
``estep.py:33-84`` (``estep_py``, the path ``deeplab.py:114`` really uses) followed by
the ``tf.argmax(axis=3)`` of ``deeplab.py:123``.  It is *not* the product: only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and only as the checker.

Differences in *form* from the reference (same arithmetic, same results):

* the class visiting order is an explicit input ``perms[num_iter, C-1]`` instead of
  the hidden ``random.shuffle`` state (``estep.py:64-66``); :func:`draw_perms`
  reproduces the reference's draw sequence for a given ``random.seed``;
* images are processed one at a time (they are independent once the batch-global
  maximum used by the suppress step, ``estep.py:50``, is known);
* the running per-pixel maximum is kept incrementally (exact, because each
  threshold is >= +0 and only one channel changes per update);
* the input is not mutated.

Pinning: ``tests/test_oracle_estep.py`` checks this file bit-for-bit (label maps and
present channels) against golden vectors produced by importing the reference's own
``estep.py`` in the build container (``oracle/make_goldens.py``).
"""
from __future__ import annotations

import random as _random

import numpy as np

__all__ = [
    "draw_perms",
    "label_sets",
    "np_pairwise_sum_f32",
    "np_mean_f32",
    "estep_oracle",
    "resize_nearest_oracle",
]


def draw_perms(num_iter: int, num_classes: int, seed=None, rng=None) -> np.ndarray:
    """The class orders the reference would draw (``estep.py:62-66``).

    Each iteration shuffles ``list(range(1, C))`` with Python's Mersenne twister and
    visits class 0 first.  ``seed`` reproduces ``random.seed(seed)`` called right before
    the reference's ``estep``; ``rng`` may be the ``random`` module itself to consume the
    global state exactly as the reference does.
    """
    r = rng if rng is not None else _random.Random(seed)
    perms = np.zeros((num_iter, max(num_classes - 1, 0)), dtype=np.int32)
    for it in range(num_iter):
        order = list(range(1, num_classes))
        r.shuffle(order)
        perms[it] = order
    return perms


def label_sets(label: np.ndarray, num_classes: int) -> np.ndarray:
    """Image-level multi-hot ``[B, C]`` from a (shrunk) pixel label map (``estep.py:35-44``).

    The reference casts to uint8 first and ignores every value >= C (255 = void).
    """
    lab = np.asarray(label).astype(np.uint8)
    out = np.zeros((lab.shape[0], num_classes), dtype=np.uint8)
    for b in range(lab.shape[0]):
        for v in np.unique(lab[b]):
            if v < num_classes:
                out[b, v] = 1
    return out


def np_pairwise_sum_f32(a: np.ndarray) -> np.float32:
    """float32 sum in NumPy's pairwise order (numpy/_core/src/umath loops: ``pairwise_sum``).

    Restated so that the GPU kernel (and the C oracle) can follow the identical
    reduction tree: < 8 elements: left-to-right; <= 128: eight strided accumulators
    combined as ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) plus a scalar tail; otherwise split
    at ``n/2`` rounded down to a multiple of 8 and recurse.
    """
    a = np.ascontiguousarray(a, dtype=np.float32).reshape(-1)
    n = a.shape[0]
    f32 = np.float32
    if n < 8:
        res = f32(0.0)
        for i in range(n):
            res = f32(res + a[i])
        return res
    if n <= 128:
        r = [f32(a[j]) for j in range(8)]
        i = 8
        while i < n - (n % 8):
            for j in range(8):
                r[j] = f32(r[j] + a[i + j])
            i += 8
        res = f32(f32(f32(r[0] + r[1]) + f32(r[2] + r[3])) + f32(f32(r[4] + r[5]) + f32(r[6] + r[7])))
        while i < n:
            res = f32(res + a[i])
            i += 1
        return res
    n2 = n // 2
    n2 -= n2 % 8
    return f32(np_pairwise_sum_f32(a[:n2]) + np_pairwise_sum_f32(a[n2:]))


def np_mean_f32(a: np.ndarray) -> np.float32:
    """``np.mean`` of a contiguous float32 vector: pairwise sum, then one fp32 divide."""
    a = np.asarray(a, dtype=np.float32).reshape(-1)
    return np.float32(np_pairwise_sum_f32(a) / np.float32(a.shape[0]))


def resize_nearest_oracle(label: np.ndarray, out_hw) -> np.ndarray:
    """``tf.image.resize_nearest_neighbor(label, out_hw)`` with align_corners=False
    (``deeplab.py:110``): src = min(floor(dst * in/out), in-1), scale computed in fp32."""
    label = np.asarray(label)
    squeeze = False
    if label.ndim == 4:
        label = label[..., 0]
        squeeze = True
    B, H, W = label.shape
    oh, ow = out_hw
    hs = np.float32(H) / np.float32(oh)
    ws = np.float32(W) / np.float32(ow)
    yi = np.minimum(np.floor(np.arange(oh, dtype=np.float32) * hs).astype(np.int64), H - 1)
    xi = np.minimum(np.floor(np.arange(ow, dtype=np.float32) * ws).astype(np.int64), W - 1)
    out = label[:, yi][:, :, xi]
    del squeeze
    return out


def _kth_smallest(v: np.ndarray, k: int) -> np.float32:
    return np.partition(v, k)[k]


def estep_oracle(
    feature_map: np.ndarray,
    label: np.ndarray,
    perms: np.ndarray,
    suppress_others: bool = True,
    num_iter: int = 5,
    margin_others: float = 1e-5,
    bg_p: float = 0.5,
    fg_p: float = 0.25,
    multihot: np.ndarray | None = None,
    return_unshifted: bool = False,
):
    """E-step of ``estep.py:33-84`` with an explicit visiting order.

    Args:
      feature_map: float32 ``[B, H, W, C]`` raw fc8 logits (not mutated).
      label: ``[B, H, W]`` shrunk pixel labels (any real dtype; cast to uint8 like the
        reference), ignored when ``multihot`` is given.
      perms: int ``[num_iter, C-1]``, each row a permutation of 1..C-1.

    Returns ``(float_map float32 [B,H,W,C], label_map int64 [B,H,W])`` where
    ``label_map = argmax(float_map, axis=3)`` with first-max tie-break
    (``deeplab.py:123``).  With ``return_unshifted`` a third array holds the map before
    the final mean re-centring (``estep.py:81-82``).
    """
    f = np.array(feature_map, dtype=np.float32, copy=True)
    B, H, W, C = f.shape
    HW = H * W
    L = label_sets(label, C) if multihot is None else np.asarray(multihot).astype(np.uint8)
    margin = np.float32(margin_others)

    if suppress_others:
        # estep.py:46-55.  The maximum is taken over the WHOLE batch (np.amax, :50).
        gmax = np.float32(f.max()) if f.size else np.float32(0)
        for b in range(B):
            absent = L[b] < 1
            if not absent.any():
                continue
            tmp = f[b].copy()
            tmp[:, :, absent] = tmp[:, :, absent] + gmax
            m = tmp.min(axis=2)  # [H, W]
            cur = f[b][:, :, absent]
            repl = (m - margin)[:, :, None].astype(np.float32)
            f[b][:, :, absent] = np.where(cur > m[:, :, None], repl, cur)

    # estep.py:56
    mean0 = np.array([np_mean_f32(f[b].max(axis=2)) for b in range(B)], dtype=np.float32)
    k_bg = int(HW * bg_p)  # estep.py:58-59 (double arithmetic, truncation)
    k_fg = int(HW * fg_p)

    perms = np.asarray(perms).reshape(num_iter, max(C - 1, 0)) if num_iter > 0 else perms
    for b in range(B):
        fb = f[b].reshape(HW, C)
        mx = fb.max(axis=1)
        for it in range(num_iter):
            order = [0] + [int(c) for c in perms[it]]
            for j in order:
                if L[b, j] < 1:
                    continue
                d = (mx - fb[:, j]).astype(np.float32)
                t = _kth_smallest(d, k_bg if j == 0 else k_fg)
                fb[:, j] = fb[:, j] + t
                mx = np.maximum(mx, fb[:, j])

    unshifted = f.copy() if return_unshifted else None
    mean1 = np.array([np_mean_f32(f[b].max(axis=2)) for b in range(B)], dtype=np.float32)
    f += (mean0 - mean1).reshape(B, 1, 1, 1)  # estep.py:81-82
    labels = np.argmax(f, axis=3).astype(np.int64)
    if return_unshifted:
        return f, labels, unshifted
    return f, labels
