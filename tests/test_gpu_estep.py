"""GPU parity of the HIP E-step against the reference-generated goldens and the CPU oracle.

Bar (north_star): label maps bit-identical; here the float maps are bit-identical too.
Everything goes through the C ABI (libemadapt_hip.so): the host entries (`emadapt_estep_host`, `e_step`,
`e_step_perm`) and the device entry (`emadapt_estep`).
"""
import random

import numpy as np
import pytest

from conftest import golden_cases, load_golden

pytestmark = pytest.mark.gpu

CASES = golden_cases()


def _params(g):
    return dict(suppress_others=bool(g["suppress_others"]), num_iter=int(g["num_iter"]),
                margin_others=float(g["margin_others"]), bg_p=float(g["bg_p"]), fg_p=float(g["fg_p"]))


@pytest.fixture(scope="module")
def pkg():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import em_adapt_tensorflow_amd.estep as est
    import em_adapt_tensorflow_amd.ops as ops
    return est, ops, torch


@pytest.mark.parametrize("name", CASES)
def test_host_operator_matches_reference_goldens(pkg, name):
    est, _, _ = pkg
    g = load_golden(name)
    fm, labels = est.estep_labels(g["logits"], g["label"], perm=g["perms"], **_params(g))
    assert np.array_equal(labels, g["label_map"].astype(np.int64)), "label map differs (%d px)" % (
        labels != g["label_map"]).sum()
    if "float_map" in g:
        assert np.array_equal(fm, g["float_map"]), "float map max abs diff %g" % np.abs(fm - g["float_map"]).max()


@pytest.mark.parametrize("name", ["s41_b2_mixed", "tiny_a", "s41_negative"])
def test_drop_in_random_state_and_inplace(pkg, name):
    """random.seed(s); estep(...) reproduces the reference's output (same draws from `random`), mutates the
    float32 input in place like the reference and returns a copy."""
    est, _, _ = pkg
    g = load_golden(name)
    x = g["logits"].copy()
    random.seed(int(g["seed"]))
    out = est.estep(x, g["label"], **_params(g))
    assert out.dtype == np.float32 and out is not x
    assert np.array_equal(out, g["float_map"])
    assert np.array_equal(x, g["float_map"]), "input must be updated in place (reference side effect)"
    assert np.array_equal(np.argmax(out, axis=3), g["label_map"])


@pytest.mark.parametrize("name", ["s41_b16", "s65_b2_c41", "s41_all_void", "odd_9x13x5", "s41_quantised"])
@pytest.mark.parametrize("variant", ["padded_f32", "multihot_nomap", "i32"])
def test_device_entry(pkg, name, variant):
    """emadapt_estep on device buffers: padded class stride, multi-hot image labels, NULL float map."""
    est, ops, torch = pkg
    from oracle import estep_oracle as eo
    g = load_golden(name)
    B, H, W, C = g["logits"].shape
    dev = torch.device("cuda:0")
    perms = torch.tensor(g["perms"], dtype=torch.int32, device=dev)
    p = _params(g)
    want = g["label_map"].astype(np.int64)
    if variant == "padded_f32":
        ldl = (C + 3) // 4 * 4 + 4
        lg = torch.full((B, H, W, ldl), 1e30, dtype=torch.float32, device=dev)  # poison the padding
        lg[..., :C] = torch.tensor(g["logits"], device=dev)
        lab = torch.tensor(g["label"], dtype=torch.float32, device=dev)
        labels, fmap = ops.estep_device(lg, lab, perms, C, want_map=True, **p)
        if "float_map" in g:
            assert np.array_equal(fmap.cpu().numpy(), g["float_map"])
    elif variant == "multihot_nomap":
        mh = torch.tensor(eo.label_sets(g["label"], C), dtype=torch.uint8, device=dev)
        lg = torch.tensor(g["logits"], device=dev)
        labels, fmap = ops.estep_device(lg, mh, perms, C, label_kind=ops.LABEL_MULTIHOT, want_map=False, **p)
    else:
        lab = torch.tensor(g["label"].astype(np.int32), dtype=torch.int32, device=dev)
        lg = torch.tensor(g["logits"], device=dev)
        labels, fmap = ops.estep_device(lg, lab, perms, C, label_kind=ops.LABEL_I32, want_map=True, **p)
    torch.cuda.synchronize()
    got = labels.cpu().numpy()
    assert got.dtype == np.int64
    assert np.array_equal(got, want), "%d pixels differ" % (got != want).sum()


def test_reference_abi_e_step_perm(pkg):
    """The reference's own ctypes call shape (estep.py:29-31) against our `e_step_perm`/`e_step` exports,
    including a non-contiguous (strided) map like numpy would hand over."""
    est, _, _ = pkg
    from oracle import estep_oracle as eo
    g = load_golden("s41_b2_mixed")
    p = _params(g)
    want, want_labels = eo.estep_oracle(g["logits"], g["label"], g["perms"], **p)
    out = est.estep_c(g["logits"].copy(), g["label"], p["suppress_others"], p["num_iter"], p["margin_others"],
                      p["bg_p"], p["fg_p"], perm=g["perms"])
    assert np.array_equal(out, want)
    # strided view: channel-padded parent array
    B, H, W, C = g["logits"].shape
    parent = np.zeros((B, H, W, C + 3), np.float32)
    view = parent[..., :C]
    view[...] = g["logits"]
    import ctypes
    from em_adapt_tensorflow_amd import _lib
    lib = _lib.load()
    lab = np.ascontiguousarray(g["label"], dtype=np.int32)
    perm = np.ascontiguousarray(g["perms"], dtype=np.int32)
    ret = lib.e_step_perm(view.ctypes.data_as(ctypes.c_void_p), view.ctypes.shape_as(ctypes.c_int),
                          view.ctypes.strides_as(ctypes.c_int), lab.ctypes.data_as(ctypes.c_void_p),
                          perm.ctypes.data_as(ctypes.c_void_p), p["suppress_others"], p["num_iter"],
                          p["margin_others"], p["bg_p"], p["fg_p"])
    assert ret == view.ctypes.data
    assert np.array_equal(view, want)
    assert np.all(parent[..., C:] == 0)
    # e_step (internal seeded order): deterministic for a given seed, valid labels, batch >= 2 does not crash
    lib.emadapt_estep_seed(123)
    a = est.estep_c(g["logits"].copy(), g["label"], True, 5, 1e-5, 0.4, 0.2)
    lib.emadapt_estep_seed(123)
    b = est.estep_c(g["logits"].copy(), g["label"], True, 5, 1e-5, 0.4, 0.2)
    assert np.array_equal(a, b)
    present = set(np.unique(g["label"][g["label"] < C]).astype(int).tolist())
    assert set(np.unique(np.argmax(a, axis=3)).tolist()) <= present


def test_random_large_batches_against_c_oracle(pkg):
    """Seeded random inputs at the benchmark geometry (B=16, 41x41x21) and the 513 geometry (65x65x41)."""
    est, ops, torch = pkg
    from oracle import estep_oracle as eo
    from oracle import estep_oracle_c as eoc
    rng = np.random.default_rng(7)
    for (B, S, C, scale) in [(16, 41, 21, 3.0), (4, 65, 41, 2.0), (3, 41, 21, 1e-3)]:
        logits = rng.normal(0, scale, (B, S, S, C)).astype(np.float32)
        label = np.zeros((B, S, S), np.float32)
        for b in range(B):
            for c in rng.choice(np.arange(1, C), size=int(rng.integers(0, 6)), replace=False):
                y, x = rng.integers(0, S - 8, 2)
                label[b, y:y + 8, x:x + 8] = c
        perms = eo.draw_perms(5, C, seed=int(rng.integers(1 << 30)))
        want_map, want = eoc.estep_oracle_c(logits, label, perms, True, 5, 1e-5, 0.4, 0.2)
        fm, got = est.estep_labels(logits, label, True, 5, 1e-5, 0.4, 0.2, perm=perms)
        assert np.array_equal(got, want), (B, S, C, int((got != want).sum()))
        assert np.array_equal(fm, want_map)


def test_every_pixels_per_thread_variant_against_c_oracle(pkg):
    """The kernel is instantiated for 1 / 4 / 9 / 16 pixels per thread (512 threads per image); 41 x 41 and 65 x 65 take the
    middle two.  Here the smallest and the largest, each with every class LDS-resident and with the LDS overflowing (the class
    images that do not fit live class-major in the workspace, written by estep_prep_kernel; labels-only calls finish their argmax in
    estep_finish_kernel), label maps and float maps bit for bit against the C restatement of estep.py:33-84."""
    est, ops, torch = pkg
    from oracle import estep_oracle as eo
    from oracle import estep_oracle_c as eoc
    rng = np.random.default_rng(11)
    for (B, H, W, C) in [(3, 20, 25, 6), (2, 81, 81, 4), (1, 90, 90, 30), (2, 70, 71, 9)]:
        logits = rng.normal(0, 2.5, (B, H, W, C)).astype(np.float32)
        label = np.zeros((B, H, W), np.float32)
        for b in range(B):
            for c in rng.choice(np.arange(1, C), size=int(rng.integers(1, min(C, 6))), replace=False):
                y, x = rng.integers(0, H - 8), rng.integers(0, W - 8)
                label[b, y:y + 8, x:x + 8] = c
        label[:, :2, :] = 255   # a border of void labels, as the synthetic batches have
        perms = eo.draw_perms(5, C, seed=int(rng.integers(1 << 30)))
        want_map, want = eoc.estep_oracle_c(logits, label, perms, True, 5, 1e-5, 0.4, 0.2)
        fm, got = est.estep_labels(logits, label, True, 5, 1e-5, 0.4, 0.2, perm=perms)
        assert np.array_equal(got, want), (B, H, W, C, int((got != want).sum()))
        assert np.array_equal(fm, want_map), (B, H, W, C)


def test_resize_nearest(pkg):
    _, ops, torch = pkg
    from oracle import estep_oracle as eo
    rng = np.random.default_rng(0)
    for (H, oh) in [(321, 41), (513, 65), (65, 9), (7, 7)]:
        lab = rng.integers(0, 255, (2, H, H, 1)).astype(np.float32)
        got = ops.resize_nearest(torch.tensor(lab, device="cuda:0"), oh, oh).cpu().numpy()
        assert np.array_equal(got, eo.resize_nearest_oracle(lab, (oh, oh)))


def test_argument_errors(pkg):
    est, ops, torch = pkg
    from em_adapt_tensorflow_amd import _lib
    x = np.zeros((1, 4, 4, 3), np.float32)
    with pytest.raises(_lib.EmadaptError):
        est.estep_labels(x, np.zeros((1, 4, 4)), bg_p=1.5, perm=np.array([[1, 2]] * 5))  # k out of range
    # empty batch is a no-op
    fm, lab = est.estep_labels(np.zeros((0, 4, 4, 3), np.float32), np.zeros((0, 4, 4)), perm=np.array([[1, 2]] * 5))
    assert fm.shape == (0, 4, 4, 3) and lab.shape == (0, 4, 4)


def test_labels_only_path_of_the_overflow_geometry_against_c_oracle(pkg):
    """The training step asks for the label map alone.  Where the class images overflow the LDS (65 x 65 x 41) that call runs
    estep_prep_kernel (suppress step on the whole chip), the E-step kernel (argmax over its LDS-resident classes only) and
    estep_finish_kernel (the other class images, merged: larger value, lower class on a tie) -- round 4.  Bit for bit against the C
    oracle's label map, on: random logits; MORE present classes than fit in LDS (present classes among the non-resident ones);
    logits quantised to a few values (exact ties between resident and non-resident classes, all-equal rows); suppression off;
    multi-hot labels; and an f32 row pitch wider than the class count (the logits read in place from a padded buffer)."""
    est, ops, torch = pkg
    from oracle import estep_oracle as eo
    from oracle import estep_oracle_c as eoc
    rng = np.random.default_rng(23)
    S, C = 65, 41
    assert ops._lib.load().emadapt_estep_workspace_bytes(2, S, S, C) > 2 * C * S * S * 4      # overflow geometry
    cases = []
    for name, npresent, quant, suppress in [("random", 3, 0, True), ("many present", 12, 0, True), ("ties", 4, 4, True),
                                            ("all equal", 2, 1, True), ("no suppress", 5, 0, False), ("ties, no suppress", 9, 3, False)]:
        B = 2
        logits = rng.normal(0, 2.0, (B, S, S, C)).astype(np.float32)
        if quant == 1:
            logits[...] = 0.75
        elif quant:
            logits = (np.round(logits * quant) / quant).astype(np.float32)
        label = np.zeros((B, S, S), np.float32)
        for b in range(B):
            for c in rng.choice(np.arange(1, C), size=npresent, replace=False):
                y, x = rng.integers(0, S - 6, 2)
                label[b, y:y + 6, x:x + 6] = c
        cases.append((name, logits, label, suppress))
    for name, logits, label, suppress in cases:
        B = logits.shape[0]
        perms = eo.draw_perms(5, C, seed=int(rng.integers(1 << 30)))
        _, want = eoc.estep_oracle_c(logits, label, perms, suppress, 5, 1e-5, 0.4, 0.2)
        lg = torch.tensor(logits, device="cuda:0")
        lb = torch.tensor(label, device="cuda:0")
        pm = torch.tensor(np.asarray(perms, dtype=np.int32), device="cuda:0")
        got, fmap = ops.estep_device(lg, lb, pm, C, suppress, 5, 1e-5, 0.4, 0.2)
        assert fmap is None and np.array_equal(got.cpu().numpy(), want), (name, int((got.cpu().numpy() != want).sum()))
        # the same through a padded row pitch (44 floats, as the model's fc8 buffer) and multi-hot labels
        padded = torch.zeros((B, S, S, 44), device="cuda:0")
        padded[..., :C] = lg
        mh = np.zeros((B, C), np.uint8)
        for b in range(B):
            for v in np.unique(label[b]).astype(np.int64):
                mh[b, v] = 1
        got2, _ = ops.estep_device(padded, torch.tensor(mh, device="cuda:0"), pm, C, suppress, 5, 1e-5, 0.4, 0.2,
                                   label_kind=ops.LABEL_MULTIHOT)
        assert np.array_equal(got2.cpu().numpy(), want), name
