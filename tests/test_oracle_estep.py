"""Pin the CPU oracle (NumPy and C restatements) against golden vectors produced by the
reference's own estep.py (oracle/make_goldens.py; reference estep.py:33-84, deeplab.py:123).

Bar: label maps bit-exact; float maps bit-exact (the mean re-centring follows NumPy's pairwise
summation order, restated in the oracle).
"""
import random

import numpy as np
import pytest

from conftest import golden_cases, load_golden
from oracle import estep_oracle as eo
from oracle import estep_oracle_c as eoc

CASES = golden_cases()


def _params(g):
    return dict(suppress_others=bool(g["suppress_others"]), num_iter=int(g["num_iter"]),
                margin_others=float(g["margin_others"]), bg_p=float(g["bg_p"]), fg_p=float(g["fg_p"]))


def test_goldens_present():
    assert len(CASES) >= 14, CASES


@pytest.mark.parametrize("name", CASES)
def test_draw_perms_matches_reference_shuffle(name):
    g = load_golden(name)
    C = g["logits"].shape[3]
    assert np.array_equal(eo.draw_perms(int(g["num_iter"]), C, seed=int(g["seed"])), g["perms"])
    # the module-level generator is consumed exactly like the reference does (estep.py:64-66)
    random.seed(int(g["seed"]))
    assert np.array_equal(eo.draw_perms(int(g["num_iter"]), C, rng=random), g["perms"])


@pytest.mark.parametrize("name", CASES)
def test_numpy_oracle_matches_reference(name):
    g = load_golden(name)
    logits_before = g["logits"].copy()
    fmap, labels = eo.estep_oracle(g["logits"], g["label"], g["perms"], **_params(g))
    assert np.array_equal(g["logits"], logits_before), "oracle must not mutate its input"
    assert labels.dtype == np.int64
    assert np.array_equal(labels, g["label_map"].astype(np.int64)), "label map differs from reference"
    if "float_map" in g:
        assert fmap.dtype == np.float32
        assert np.array_equal(fmap, g["float_map"]), "float map differs from reference (max abs %g)" % np.abs(
            fmap - g["float_map"]).max()


@pytest.mark.parametrize("name", CASES)
def test_c_oracle_matches_reference(name):
    g = load_golden(name)
    fmap, labels = eoc.estep_oracle_c(g["logits"], g["label"], g["perms"], **_params(g))
    assert np.array_equal(labels, g["label_map"].astype(np.int64))
    if "float_map" in g:
        assert np.array_equal(fmap, g["float_map"])


def test_np_pairwise_sum_is_numpys():
    rng = np.random.default_rng(0)
    for n in [1, 7, 8, 9, 127, 128, 129, 1000, 1681, 4225, 5000]:
        a = rng.normal(0, 10, n).astype(np.float32)
        assert eo.np_pairwise_sum_f32(a) == np.add.reduce(a), n
        assert eo.np_mean_f32(a) == np.mean(a), n
    # the shape the reference reduces: mean over axes (1,2) of a contiguous [B,H,W] array (estep.py:56)
    x = rng.normal(0, 3, (3, 41, 41)).astype(np.float32)
    want = np.mean(x, (1, 2))
    got = np.array([eo.np_mean_f32(x[b]) for b in range(3)], np.float32)
    assert np.array_equal(want, got)


def test_resize_nearest_rule():
    # 321 -> 41 (deeplab.py:110): src = floor(dst * 321/41)
    lab = np.arange(321 * 321, dtype=np.float32).reshape(1, 321, 321, 1)
    out = eo.resize_nearest_oracle(lab, (41, 41))
    idx = np.minimum(np.floor(np.arange(41) * (321.0 / 41.0)).astype(int), 320)
    assert np.array_equal(out[0], lab[0, :, :, 0][idx][:, idx])
    assert out.shape == (1, 41, 41)


def test_label_sets_ignores_void_and_out_of_range():
    lab = np.array([[[0, 3, 255], [21, 20, 3]]], np.float32)
    assert eo.label_sets(lab, 21)[0].nonzero()[0].tolist() == [0, 3, 20]


def test_order_dependence_is_real():
    """The E-step result depends on the visiting order: different perms may change labels,
    identical perms never do."""
    g = load_golden("s41_b2_mixed")
    p = _params(g)
    _, a = eo.estep_oracle(g["logits"], g["label"], g["perms"], **p)
    _, b = eo.estep_oracle(g["logits"], g["label"], g["perms"], **p)
    assert np.array_equal(a, b)
