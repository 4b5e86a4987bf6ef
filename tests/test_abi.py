"""CPU-side checks of the drop-in boundary: the C-ABI library builds for gfx950, loads without a GPU and
exports every symbol include/emadapt.h declares; argument validation that needs no device."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "emadapt.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"^\s*#.*$", "", src, flags=re.M)
    names = re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\([^;{}]*\)\s*;", src)
    return sorted(set(n for n in names if n not in ("defined",)))


@pytest.fixture(scope="module")
def lib():
    from em_adapt_tensorflow_amd import _lib
    return _lib.load()


def test_header_declares_the_expected_surface():
    names = declared_functions()
    for must in ["e_step", "e_step_perm", "emadapt_estep", "emadapt_estep_host", "emadapt_conv_fwd",
                 "emadapt_conv_dgrad", "emadapt_conv_wgrad", "emadapt_maxpool_fwd", "emadapt_maxpool_bwd",
                 "emadapt_softmax_ce", "emadapt_sgd_accumulate", "emadapt_sgd_momentum", "emadapt_prof_collect"]:
        assert must in names, must
    assert len(names) >= 30


def test_library_exports_every_declared_symbol(lib):
    from em_adapt_tensorflow_amd import _lib
    raw = ctypes.CDLL(_lib.LIB_PATH)
    missing = [n for n in declared_functions() if not hasattr(raw, n)]
    assert not missing, "declared in include/emadapt.h but not exported: %s" % missing
    unbound = [n for n in declared_functions() if n not in _lib.PROTOTYPES]
    assert not unbound, "declared but not bound in _lib.PROTOTYPES: %s" % unbound


def test_reference_e_step_signature_is_bindable(lib):
    """The reference binds e_step with these argtypes (estep.py:15); ours must accept the same."""
    from ctypes import POINTER, c_bool, c_float, c_int
    raw = ctypes.CDLL(lib._name)
    raw.e_step.argtypes = [POINTER(c_float), POINTER(c_int), POINTER(c_int), POINTER(c_int), c_bool, c_int, c_float,
                           c_float, c_float]
    assert raw.e_step is not None


def test_version_and_errors(lib):
    assert lib.emadapt_version() >= 200
    assert lib.emadapt_strerror(0) == b"ok"
    assert b"workspace" in lib.emadapt_strerror(-4)
    assert lib.emadapt_estep_needs_map(41, 41, 21) == 0   # 321x321 / 21 classes: everything fits in LDS
    assert lib.emadapt_estep_needs_map(65, 65, 41) == 0   # 513x513 / 41 classes: the overflow lives in the workspace
    assert lib.emadapt_estep_workspace_bytes(10, 65, 65, 41) == 256 + 10 * (41 + 3) * 4228 * 4 + 48   # class images, maxima, partial argmax, shifts
    assert lib.emadapt_estep_workspace_bytes(16, 41, 41, 21) == 256
    assert lib.emadapt_estep_needs_map(200, 200, 21) == -2
    assert lib.emadapt_estep_workspace_bytes(16, 41, 41, 21) >= 16


def test_argument_validation_without_device(lib):
    from em_adapt_tensorflow_amd import _lib
    # null pointers / bad sizes are rejected before any HIP call
    assert lib.emadapt_estep(None, 21, None, 0, 0, None, 1, 41, 41, 21, 1, 5, 1e-5, 672, 336, None, None, None, 0,
                             None) == -1
    assert lib.emadapt_maxpool_fwd(None, 1, 4, 4, 4, 2, None, None, None) == -1
    d = _lib.ConvDesc(1, 8, 8, 3, 8, 3, 3, 1)  # Cin not a multiple of 4
    assert lib.emadapt_conv_fwd(ctypes.byref(d), 16, 16, None, 16, 0, 1.0, None, 0, None, 0, None) == -2
    assert lib.emadapt_conv_wgrad_workspace_bytes(ctypes.byref(_lib.ConvDesc(16, 321, 321, 64, 64, 3, 3, 1))) > 0
    assert lib.emadapt_conv_wgrad_workspace_bytes(ctypes.byref(_lib.ConvDesc(16, 41, 41, 512, 4096, 4, 4, 4))) == 0


def test_host_estep_reports_missing_device(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from em_adapt_tensorflow_amd import _lib, estep
    with pytest.raises(_lib.EmadaptError) as e:
        estep.estep(np.zeros((1, 2, 2, 3), np.float32), np.zeros((1, 2, 2)))
    assert e.value.code in (-5, -3)


def test_ops_refuse_cpu_tensors():
    import torch
    from em_adapt_tensorflow_amd import ops
    with pytest.raises(ValueError):
        ops.conv_fwd(torch.zeros(1, 4, 4, 4), torch.zeros(3, 3, 4, 4))
