"""Shared pytest configuration.

Markers:
  gpu  -- needs a real MI355X (the driver runs `-m gpu` on the GPU box, `-m "not gpu"` in the
          CPU-only build container).
"""
import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X GPU")


def golden_cases(prefix="estep_"):
    return sorted(os.path.basename(p)[len(prefix):-4] for p in glob.glob(os.path.join(GOLDEN_DIR, prefix + "*.npz")))


def load_golden(name, prefix="estep_"):
    with np.load(os.path.join(GOLDEN_DIR, prefix + name + ".npz")) as z:
        g = {k: z[k] for k in z.files}
    g["logits"] = g["logits"].astype(np.float32)  # f16-stored cases are exact in f32
    return g


@pytest.fixture(scope="session")
def oracle_clib():
    """oracle/libestep_oracle.so, built on demand (gcc only; no GPU, no reference needed)."""
    import ctypes
    import subprocess

    so = os.path.join(ROOT, "oracle", "libestep_oracle.so")
    src = os.path.join(ROOT, "oracle", "estep_oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "libestep_oracle.so"])
    return ctypes.CDLL(so)
