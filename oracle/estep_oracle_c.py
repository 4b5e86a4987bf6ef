"""ctypes binding of oracle/estep_oracle.c (the C restatement).  TEST INFRASTRUCTURE ONLY."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


def load(build=True):
    global _lib
    if _lib is None:
        so = os.path.join(_HERE, "libestep_oracle.so")
        src = os.path.join(_HERE, "estep_oracle.c")
        if build and (not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src)):
            subprocess.check_call(["make", "-C", _HERE, "libestep_oracle.so"], stdout=subprocess.DEVNULL)
        _lib = ctypes.CDLL(so)
        _lib.estep_oracle_c.restype = ctypes.c_int
        _lib.estep_oracle_c.argtypes = [
            ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
            ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_float,
            ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        _lib.estep_oracle_label_sets.restype = None
        _lib.estep_oracle_label_sets.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                 ctypes.c_void_p]
    return _lib


def estep_oracle_c(logits, label, perms, suppress_others=True, num_iter=5, margin_others=1e-5,
                   bg_p=0.5, fg_p=0.25, multihot=None):
    """Same contract as oracle.estep_oracle.estep_oracle, computed by the C restatement."""
    lib = load()
    logits = np.ascontiguousarray(logits, dtype=np.float32)
    B, H, W, C = logits.shape
    HW = H * W
    if multihot is None:
        lab = np.ascontiguousarray(np.asarray(label, dtype=np.float32).reshape(B, -1))
        mh = np.zeros((B, C), np.uint8)
        # the reference takes unique() over whatever label array it is handed (estep.py:39)
        lib.estep_oracle_label_sets(lab.ctypes.data, B, lab.shape[1], C, mh.ctypes.data)
    else:
        mh = np.ascontiguousarray(multihot, dtype=np.uint8)
    perms = np.ascontiguousarray(perms, dtype=np.int32)
    fmap = np.empty((B, H, W, C), np.float32)
    labels = np.empty((B, H, W), np.int64)
    rc = lib.estep_oracle_c(logits.ctypes.data, B, HW, C, C, mh.ctypes.data, perms.ctypes.data,
                            int(bool(suppress_others)), int(num_iter), float(np.float32(margin_others)),
                            int(HW * bg_p), int(HW * fg_p), fmap.ctypes.data, labels.ctypes.data)
    if rc != 0:
        raise RuntimeError("estep_oracle_c failed: %d" % rc)
    return fmap, labels
