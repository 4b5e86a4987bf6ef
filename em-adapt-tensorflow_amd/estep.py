"""E-step operator with the reference's signature (estep.py:18-27), computed on the GPU.

    estep(feature_map, label, suppress_others=True, num_iter=5, margin_others=1e-5,
          bg_p=0.5, fg_p=0.25, use_c=False) -> np.ndarray float32 [B,H,W,C]

Drop-in notes
  * Like the reference the float32 input array is updated in place and a float32 copy is returned.
  * ``use_c`` is accepted for signature compatibility; both values run the HIP library (the
    reference's NumPy path and its -- broken for B>=2 -- C++ path are both replaced; semantics are
    those of estep_py, estep.py:33-84).
  * The class visiting order: by default it is drawn from Python's global ``random`` module with
    exactly the reference's calls (estep.py:64-66), so ``random.seed(s); estep(...)`` yields the same
    label map as the reference.  ``perm=`` (int [num_iter, C-1]) or ``seed=`` pin it explicitly.
"""
import ctypes
import random as _random

import numpy as np

from . import _lib

__all__ = ["estep", "estep_labels", "draw_perms"]


def draw_perms(num_iter, num_classes, seed=None, rng=None):
    """Class orders as the reference draws them: shuffle(list(range(1, C))) per iteration."""
    r = rng if rng is not None else (_random if seed is None else _random.Random(seed))
    perms = np.zeros((num_iter, max(num_classes - 1, 0)), dtype=np.int32)
    for it in range(num_iter):
        order = list(range(1, num_classes))
        r.shuffle(order)
        perms[it] = order
    return perms


def _run(feature_map, label, suppress_others, num_iter, margin_others, bg_p, fg_p, perm, seed, want_labels):
    fm = feature_map
    if not (isinstance(fm, np.ndarray) and fm.dtype == np.float32 and fm.flags["C_CONTIGUOUS"]):
        fm = np.ascontiguousarray(fm, dtype=np.float32)
    if fm.ndim != 4:
        raise ValueError("feature_map must be [B,H,W,C], got shape %s" % (fm.shape,))
    B, H, W, C = fm.shape
    lab = np.ascontiguousarray(np.asarray(label, dtype=np.float32).reshape(B, -1)) if B else np.zeros((0, 0), np.float32)
    if perm is None:
        perm = draw_perms(num_iter, C, seed=seed)
    perm = np.ascontiguousarray(perm, dtype=np.int32).reshape(num_iter, max(C - 1, 0))
    hw = H * W
    k_bg, k_fg = int(hw * bg_p), int(hw * fg_p)  # estep.py:58-59
    labels = np.empty((B, H, W), np.int64) if want_labels else None
    _lib.call("emadapt_estep_host", fm.ctypes.data, B, H, W, C, lab.ctypes.data, _lib.LABEL_F32,
              lab.shape[1] if B else 0, perm.ctypes.data if perm.size else None, int(bool(suppress_others)),
              int(num_iter), float(np.float32(margin_others)), k_bg, k_fg,
              labels.ctypes.data if want_labels else None)
    if fm is not feature_map and isinstance(feature_map, np.ndarray) and feature_map.shape == fm.shape:
        try:
            feature_map[...] = fm  # keep the reference's in-place side effect for other dtypes/layouts
        except (TypeError, ValueError):
            pass
    return fm, labels


def estep(feature_map, label, suppress_others=True, num_iter=5, margin_others=1e-5, bg_p=0.5, fg_p=0.25,
          use_c=False, perm=None, seed=None):
    fm, _ = _run(feature_map, label, suppress_others, num_iter, margin_others, bg_p, fg_p, perm, seed, False)
    return fm.astype(np.float32)  # a copy, like the reference's `f.astype(np.float32)`


def estep_labels(feature_map, label, suppress_others=True, num_iter=5, margin_others=1e-5, bg_p=0.5, fg_p=0.25,
                 perm=None, seed=None):
    """estep(...) followed by argmax over the class axis (deeplab.py:123): (float_map, int64 labels).
    The input is not modified."""
    fm = np.array(feature_map, dtype=np.float32, copy=True, order="C")
    fm, labels = _run(fm, label, suppress_others, num_iter, margin_others, bg_p, fg_p, perm, seed, True)
    return fm, labels


def estep_c(feature_map, label, suppress_others, num_iter, margin_others, bg_p, fg_p, perm=None):
    """The reference's ctypes call shape (estep.py:29-31) against our library's `e_step` entry."""
    lib = _lib.load()
    feature_map = np.ascontiguousarray(feature_map, dtype=ctypes.c_float)
    label = np.ascontiguousarray(label, dtype=ctypes.c_int)
    args = [feature_map.ctypes.data_as(ctypes.c_void_p), feature_map.ctypes.shape_as(ctypes.c_int),
            feature_map.ctypes.strides_as(ctypes.c_int), label.ctypes.data_as(ctypes.c_void_p)]
    if perm is None:
        ret = lib.e_step(*args, suppress_others, num_iter, margin_others, bg_p, fg_p)
    else:
        perm = np.ascontiguousarray(perm, dtype=np.int32)
        ret = lib.e_step_perm(*args, perm.ctypes.data_as(ctypes.c_void_p), suppress_others, num_iter, margin_others,
                              bg_p, fg_p)
    if not ret:
        raise RuntimeError("e_step failed")
    return feature_map
