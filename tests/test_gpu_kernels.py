"""GPU parity of every HIP kernel of the conv / pool / loss / optimiser path against the CPU oracle
(oracle/deeplab_oracle.py: PyTorch-CPU ops with the TF semantics of deeplab.py spelled out, fp64 here so
that the check measures the kernel's own error).  All calls go through the C ABI.

Tolerances: conv kernels accumulate in fp32 (v_mfma_f32_32x32x2_f32 = exact fp32 FMA chain), so the
relative error vs fp64 is ~1e-6; the bar below is 2e-5 relative to the tensor's max magnitude, far inside
north_star's 1e-3.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch
    assert torch.cuda.is_available()
    import em_adapt_tensorflow_amd.ops as ops
    from oracle import deeplab_oracle as do
    return ops, do, torch


def _rel(got, want):
    want = np.asarray(want, dtype=np.float64)
    got = np.asarray(got, dtype=np.float64)
    return np.abs(got - want).max() / max(np.abs(want).max(), 1e-30)


def _t(torch, a, dtype=None):
    return torch.tensor(np.ascontiguousarray(a), dtype=dtype or torch.float32, device="cuda:0")


# (B, H, W, Cin, Cout, k, dil): covers 1x1 / 3x3 / 4x4, dilation 1/2/4/12, multi-tile M and N, N<=64 and
# N>64 tile configs, K tails (K % 32 != 0), Cin=4 (padded RGB), Cout=24 (padded 21 classes), M tails.
CONV_SHAPES = [
    (1, 7, 9, 4, 8, 3, 1),
    (2, 13, 11, 8, 16, 3, 1),
    (2, 12, 12, 16, 64, 3, 2),
    (1, 19, 17, 16, 136, 3, 1),
    (2, 11, 11, 32, 160, 4, 4),
    (1, 25, 25, 24, 40, 3, 12),
    (3, 9, 9, 160, 24, 1, 1),
    (1, 33, 31, 4, 64, 3, 1),
    (2, 23, 23, 64, 256, 3, 2),
    (1, 41, 41, 24, 288, 1, 1),
    (2, 49, 47, 256, 128, 3, 1),   # few tiles, long K: takes the split-K path (slabs + fused reduce/epilogue)
    (1, 41, 41, 512, 136, 4, 4),   # fc6-like, split-K, N tail
]


@pytest.mark.parametrize("shape", CONV_SHAPES)
def test_conv_fwd_dgrad_wgrad(env, shape):
    ops, do, torch = env
    B, H, W, Cin, Cout, k, dil = shape
    rng = np.random.default_rng(hash(shape) % (1 << 31))
    x = rng.normal(0, 1, (B, H, W, Cin)).astype(np.float32)
    w = rng.normal(0, 0.2, (k, k, Cin, Cout)).astype(np.float32)
    b = rng.normal(0, 0.5, (Cout,)).astype(np.float32)
    dy = rng.normal(0, 1, (B, H, W, Cout)).astype(np.float32)

    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    wt = torch.tensor(w, dtype=torch.float64, requires_grad=True)
    bt = torch.tensor(b, dtype=torch.float64, requires_grad=True)
    y_lin = do.conv_same(xt, wt, bt, dil)
    y_lin.backward(torch.tensor(dy, dtype=torch.float64))

    xg, wg, bg, dyg = _t(torch, x), _t(torch, w), _t(torch, b), _t(torch, dy)
    # forward: plain, +bias, +bias+relu
    assert _rel(ops.conv_fwd(xg, wg, None, dil).cpu(), (y_lin - bt).detach()) < 2e-5
    assert _rel(ops.conv_fwd(xg, wg, bg, dil, relu=False).cpu(), y_lin.detach()) < 2e-5
    assert _rel(ops.conv_fwd(xg, wg, bg, dil, relu=True).cpu(), torch.relu(y_lin).detach()) < 2e-5
    # data gradient, plain and gated by a stored activation (ReluGrad / dropout scale)
    assert _rel(ops.conv_dgrad(dyg, wg, dil).cpu(), xt.grad) < 2e-5
    gate = rng.normal(0, 1, (B, H, W, Cin)).astype(np.float32)
    want = np.where(gate > 0, xt.grad.numpy() * 2.0, 0.0)
    assert _rel(ops.conv_dgrad(dyg, wg, dil, gate=_t(torch, gate), gate_scale=2.0).cpu(), want) < 2e-5
    # filter and bias gradients
    assert _rel(ops.conv_wgrad(xg, dyg, k, k, dil).cpu(), wt.grad) < 2e-5
    assert _rel(ops.bias_grad(dyg).cpu(), bt.grad) < 2e-5


def test_conv_wgrad_split_m_is_deterministic_and_right(env):
    """Large pixel count, small filter: the split-M path (partial slabs + ordered sum)."""
    ops, do, torch = env
    rng = np.random.default_rng(5)
    B, H, W, Cin, Cout, k = 2, 97, 95, 8, 64, 3
    x = rng.normal(0, 1, (B, H, W, Cin)).astype(np.float32)
    dy = rng.normal(0, 1, (B, H, W, Cout)).astype(np.float32)
    wt = torch.zeros((k, k, Cin, Cout), dtype=torch.float64, requires_grad=True)
    y = do.conv_same(torch.tensor(x, dtype=torch.float64), wt, torch.zeros(Cout, dtype=torch.float64), 1)
    y.backward(torch.tensor(dy, dtype=torch.float64))
    xg, dyg = _t(torch, x), _t(torch, dy)
    a = ops.conv_wgrad(xg, dyg, k, k, 1).cpu().numpy()
    b = ops.conv_wgrad(xg, dyg, k, k, 1).cpu().numpy()
    assert np.array_equal(a, b), "filter gradient must be bitwise reproducible"
    assert _rel(a, wt.grad) < 2e-5


@pytest.mark.parametrize("shape", [(2, 97, 95, 8, 64, 3, 1), (3, 33, 31, 24, 40, 3, 2), (2, 50, 49, 16, 128, 3, 1), (1, 41, 41, 64, 24, 1, 1),
                                   (2, 37, 35, 4, 72, 3, 1), (1, 19, 17, 256, 256, 3, 2)])
def test_filter_gradient_with_the_bias_gradient_in_the_same_launch(env, shape):
    """emadapt_conv_wgrad_bias (BiasAddGrad summed by the producers of the round-1 filter-gradient kernel): column tails (N not a
    multiple of the tile), one and many pixel splits, k tiles beyond the first; dw bit-identical to the plain entry, db against
    fp64, repeatable bit for bit; the last shape is one the 256 x 256 kernel takes -- ops falls back to the separate kernel."""
    ops, do, torch = env
    B, H, W, Cin, Cout, k, dil = shape
    rng = np.random.default_rng(sum(shape))
    x = rng.normal(0, 1, (B, H, W, Cin)).astype(np.float32)
    dy = rng.normal(0.3, 1, (B, H, W, Cout)).astype(np.float32)
    xg, dyg = _t(torch, x), _t(torch, dy)
    d = ops.conv_desc(B, H, W, Cin, Cout, k, k, dil)
    fused = bool(ops._lib.load().emadapt_conv_wgrad_bias_fused(__import__("ctypes").byref(d)))
    assert fused == (shape != (1, 19, 17, 256, 256, 3, 2))
    for cus in (0, 1, 7):
        with ops.tuned(plan_cus=cus):
            db = torch.full((Cout,), 7.0, device="cuda")
            dw = ops.conv_wgrad(xg, dyg, k, k, dil, bias_out=db)
            db2 = torch.empty((Cout,), device="cuda")
            dw2 = ops.conv_wgrad(xg, dyg, k, k, dil, bias_out=db2)
            assert torch.equal(db, db2) and torch.equal(dw, dw2)
            assert torch.equal(dw, ops.conv_wgrad(xg, dyg, k, k, dil)), "the filter gradient must not depend on the bias output"
        want = dy.astype(np.float64).sum((0, 1, 2))
        assert np.abs(db.cpu().numpy() - want).max() < 2e-6 * np.abs(dy).sum((0, 1, 2)).max(), (shape, cus)
        assert _rel(ops.bias_grad(dyg).cpu().numpy(), want) < 1e-5
    with ops.tuned(wgrad_bias=1):   # the knob that turns the fusion off: same filter gradient, the separate kernel's bias gradient
        db3 = torch.empty((Cout,), device="cuda")
        assert torch.equal(ops.conv_wgrad(xg, dyg, k, k, dil, bias_out=db3), ops.conv_wgrad(xg, dyg, k, k, dil))
        assert torch.equal(db3, ops.bias_grad(dyg))


def test_dropout_epilogue(env):
    ops, do, torch = env
    rng = np.random.default_rng(3)
    B, H, W, Cin, Cout = 2, 9, 9, 16, 72
    x = rng.normal(0, 1, (B, H, W, Cin)).astype(np.float32)
    w = rng.normal(0, 0.3, (1, 1, Cin, Cout)).astype(np.float32)
    b = rng.normal(0, 0.1, (Cout,)).astype(np.float32)
    xg, wg, bg = _t(torch, x), _t(torch, w), _t(torch, b)
    base = ops.conv_fwd(xg, wg, bg, 1, relu=True).cpu().numpy()
    mask = (rng.random((B, H, W, Cout)) < 0.5).astype(np.uint8)
    got = ops.conv_fwd(xg, wg, bg, 1, relu=True, keep_prob=0.5, drop_mask=_t(torch, mask, torch.uint8)).cpu().numpy()
    assert np.array_equal(got, np.where(mask > 0, base / np.float32(0.5), 0).astype(np.float32))
    # built-in generator: kept values are exactly x * (1/keep), the keep rate is ~keep, seeds differ, same seed repeats
    big = ops.conv_fwd(_t(torch, rng.normal(0, 1, (4, 32, 32, Cin)).astype(np.float32)), wg, bg, 1, relu=False)
    xb = _t(torch, np.abs(rng.normal(0, 1, (4, 32, 32, Cin))).astype(np.float32) + 0.1)
    wpos = _t(torch, np.abs(w))
    ref = ops.conv_fwd(xb, wpos, _t(torch, np.abs(b)), 1, relu=True).cpu().numpy()
    d1 = ops.conv_fwd(xb, wpos, _t(torch, np.abs(b)), 1, relu=True, keep_prob=0.75, drop_seed=11).cpu().numpy()
    d1b = ops.conv_fwd(xb, wpos, _t(torch, np.abs(b)), 1, relu=True, keep_prob=0.75, drop_seed=11).cpu().numpy()
    d2 = ops.conv_fwd(xb, wpos, _t(torch, np.abs(b)), 1, relu=True, keep_prob=0.75, drop_seed=12).cpu().numpy()
    assert np.array_equal(d1, d1b) and not np.array_equal(d1, d2)
    kept = d1 != 0
    assert abs(kept.mean() - 0.75) < 0.01
    # kept elements are x * (1 / keep_prob) (TF >= 1.13's dropout; the same number as x / keep_prob for keep_prob = 2^-n)
    assert np.array_equal(d1[kept], (ref * (np.float32(1.0) / np.float32(0.75)))[kept])
    del big


def test_split_k_path_keeps_the_fused_epilogue(env):
    """A shape the split-K heuristic splits: bias+ReLU+dropout(mask) and the dgrad gate must come out the same as
    from the single-pass kernel (checked against the oracle)."""
    ops, do, torch = env
    from em_adapt_tensorflow_amd import _lib
    import ctypes
    rng = np.random.default_rng(12)
    B, H, W, Cin, Cout, k, dil = 2, 49, 47, 256, 128, 3, 2
    d = _lib.ConvDesc(B, H, W, Cin, Cout, k, k, dil)
    assert _lib.load().emadapt_conv_fwd_workspace_bytes(ctypes.byref(d)) > 0, "expected a split-K plan for this shape"
    x = rng.normal(0, 1, (B, H, W, Cin)).astype(np.float32)
    w = rng.normal(0, 0.05, (k, k, Cin, Cout)).astype(np.float32)
    b = rng.normal(0, 0.5, (Cout,)).astype(np.float32)
    mask = (rng.random((B, H, W, Cout)) < 0.5).astype(np.uint8)
    y = do.conv_same(torch.tensor(x, dtype=torch.float64), torch.tensor(w, dtype=torch.float64),
                     torch.tensor(b, dtype=torch.float64), dil)
    want = torch.relu(y).numpy() / 0.5 * mask
    got = ops.conv_fwd(_t(torch, x), _t(torch, w), _t(torch, b), dil, relu=True, keep_prob=0.5,
                       drop_mask=_t(torch, mask, torch.uint8)).cpu().numpy()
    assert _rel(got, want) < 2e-5
    a = ops.conv_fwd(_t(torch, x), _t(torch, w), _t(torch, b), dil).cpu().numpy()
    a2 = ops.conv_fwd(_t(torch, x), _t(torch, w), _t(torch, b), dil).cpu().numpy()
    assert np.array_equal(a, a2), "split-K result must be bitwise reproducible"


@pytest.mark.parametrize("shape", [(2, 21, 21, 8, 2), (1, 16, 18, 16, 2), (2, 11, 11, 12, 1), (1, 41, 41, 64, 1),
                                   (1, 33, 33, 4, 2),
                                   (2, 40, 50, 64, 2),          # even sizes: SAME pads only bottom/right
                                   (3, 1, 9, 8, 2), (3, 9, 1, 8, 1), (2, 1, 1, 4, 2),
                                   (40, 420, 40, 32, 2),        # > 16 K rows: one block per row, looping along it
                                   (40, 420, 40, 32, 1),
                                   (1, 3, 70, 32768, 2),        # W * (C/4)^2 >= 2^32: the generic-index gradient kernel
                                   (1, 2, 70, 32768, 1)])
def test_maxpool(env, shape):
    ops, do, torch = env
    B, H, W, C, stride = shape
    rng = np.random.default_rng(sum(shape))
    x = rng.normal(0, 1, (B, H, W, C)).astype(np.float32)
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    y = do.maxpool_same(xt, stride)
    dy = rng.normal(0, 1, tuple(y.shape)).astype(np.float32)
    y.backward(torch.tensor(dy, dtype=torch.float64))
    xg = _t(torch, x)
    yg, idx = ops.maxpool_fwd(xg, stride)
    assert np.array_equal(yg.cpu().numpy(), y.detach().numpy().astype(np.float32))
    dx = ops.maxpool_bwd(_t(torch, dy), idx, (B, H, W, C), stride)
    assert _rel(dx.cpu(), xt.grad) < 1e-6
    gated = ops.maxpool_bwd(_t(torch, dy), idx, (B, H, W, C), stride, gate=xg)
    assert _rel(gated.cpu(), np.where(x > 0, xt.grad.numpy(), 0)) < 1e-6


@pytest.mark.parametrize("shape", [(2, 21, 21, 8, 2), (2, 11, 11, 16, 1), (2, 40, 50, 64, 2), (3, 1, 9, 8, 2), (3, 9, 1, 8, 2), (2, 1, 1, 8, 2),
                                   (2, 2, 2, 8, 2), (1, 41, 41, 64, 1), (20, 420, 40, 32, 2), (1, 3, 70, 32768, 2)])
@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_maxpool_gate_in_the_index(env, shape, dtype):
    """emadapt_maxpool_fwd_gated: same y, same winner, bit 4 of the index byte = the winner is not > 0; the gradient on that index with
    no gate tensor is BITWISE the gradient on the plain index gated by the pooled tensor -- also where gradients are NaN / infinite
    (a NaN passes a closed gate either way) and where the input holds zeros of both signs and NaNs."""
    ops, do, torch = env
    B, H, W, C, stride = shape
    rng = np.random.default_rng(sum(shape) + 5)
    x = rng.normal(0, 1, (B, H, W, C)).astype(np.float32)
    x[rng.random(x.shape) < 0.3] = 0.0            # post-ReLU zeros: whole windows of them
    x[rng.random(x.shape) < 0.02] = -0.0
    x[rng.random(x.shape) < 0.01] = np.nan
    xg = _t(torch, x)
    if dtype == "bf16":
        xg = xg.bfloat16()
    y0, idx0 = ops.maxpool_fwd(xg, stride)
    y1, idx1 = ops.maxpool_fwd(xg, stride, relu_gate=True)
    assert torch.equal(y0.view(torch.int16 if dtype == "bf16" else torch.int32), y1.view(torch.int16 if dtype == "bf16" else torch.int32))
    assert torch.equal(idx1 & 0xF, idx0) and int(idx0.max()) <= 8
    closed = ~(y0.float() > 0)
    assert torch.equal((idx1 & ops.POOL_CLOSED) != 0, closed) and int((idx1 & 0xE0).max()) == 0
    dy = rng.normal(0, 1, tuple(y0.shape)).astype(np.float32)
    dy[rng.random(dy.shape) < 0.02] = np.nan
    dy[rng.random(dy.shape) < 0.02] = np.inf
    dy[rng.random(dy.shape) < 0.02] = -np.inf
    dyg = _t(torch, dy)
    if dtype == "bf16":
        dyg = dyg.bfloat16()
    want = ops.maxpool_bwd(dyg, idx0, (B, H, W, C), stride, gate=xg)
    got = ops.maxpool_bwd(dyg, idx1, (B, H, W, C), stride)
    it = torch.int16 if dtype == "bf16" else torch.int32
    nan_w, nan_g = torch.isnan(want), torch.isnan(got)
    assert torch.equal(nan_w, nan_g) and (want.numel() < 4096 or bool(nan_w.any()))
    assert torch.equal(torch.where(nan_w, torch.zeros_like(want), want).view(it), torch.where(nan_g, torch.zeros_like(got), got).view(it))
    if stride == 2:   # the quad form (default) against the row form
        with ops.tuned(pool_rows=1):
            rows_w = ops.maxpool_bwd(dyg, idx0, (B, H, W, C), stride, gate=xg)
            rows_g = ops.maxpool_bwd(dyg, idx1, (B, H, W, C), stride)
        for r in (rows_w, rows_g):
            assert torch.equal(torch.isnan(r), nan_w)
            assert torch.equal(torch.where(nan_w, torch.zeros_like(r), r).view(it), torch.where(nan_w, torch.zeros_like(want), want).view(it))
    # an explicit gate on top of a gated index changes nothing
    both = ops.maxpool_bwd(dyg, idx1, (B, H, W, C), stride, gate=xg)
    assert torch.equal(torch.isnan(both), nan_w) and torch.equal(torch.where(nan_w, torch.zeros_like(both), both).view(it),
                                                                 torch.where(nan_w, torch.zeros_like(want), want).view(it))


def test_maxpool_ties_go_to_first_maximum(env):
    ops, do, torch = env
    x = np.zeros((1, 5, 5, 4), np.float32)  # all equal (post-ReLU zeros)
    yg, idx = ops.maxpool_fwd(_t(torch, x), 1)
    idx = idx.cpu().numpy()
    assert idx[0, 2, 2, 0] == 0 and idx[0, 0, 0, 0] == 4  # interior: first tap; corner: first valid tap (1,1)


def test_softmax_ce_and_l2(env):
    ops, do, torch = env
    rng = np.random.default_rng(9)
    M, C, ldl = 3000, 21, 24
    logits = rng.normal(0, 3, (M, ldl)).astype(np.float32)
    labels = rng.integers(0, C, (M,)).astype(np.int64)
    lt = torch.tensor(logits[:, :C], dtype=torch.float64, requires_grad=True)
    loss = torch.nn.functional.cross_entropy(lt, torch.tensor(labels), reduction="sum")
    (loss / M).backward()
    ls, dl = ops.softmax_ce(_t(torch, logits), _t(torch, labels, torch.int64), C)
    assert abs(float(ls.cpu()) - float(loss)) / float(loss) < 1e-5
    dl = dl.cpu().numpy()
    assert _rel(dl[:, :C], lt.grad) < 1e-5
    assert np.all(dl[:, C:] == 0)
    x = rng.normal(0, 1, (100003,)).astype(np.float32)
    assert abs(float(ops.l2_loss(_t(torch, x)).cpu()) - float((x.astype(np.float64) ** 2).sum() / 2)) < 1e-2


def test_sgd_kernels_and_padding(env):
    ops, do, torch = env
    rng = np.random.default_rng(4)
    for n, nd in [(4096, 4000), (1003, 500)]:
        w = rng.normal(0, 1, n).astype(np.float32)
        g = rng.normal(0, 1, n).astype(np.float32)
        acc = rng.normal(0, 1, n).astype(np.float32)
        v = rng.normal(0, 1, n).astype(np.float32)
        wd, scale, lr, mu = np.float32(1e-2), np.float32(0.2), np.float32(1e-3), np.float32(0.9)
        accg, wg_, vg = _t(torch, acc), _t(torch, w), _t(torch, v)
        ops.sgd_accumulate(accg, _t(torch, g), wg_, nd, wd, scale)
        gg = g.copy()
        gg[:nd] = gg[:nd] + wd * w[:nd]
        want_acc = acc + gg * scale
        assert _rel(accg.cpu(), want_acc) < 1e-6
        ops.sgd_momentum(wg_, vg, accg, lr, mu, 1.0)
        want_v = mu * v + want_acc
        assert _rel(vg.cpu(), want_v) < 1e-6
        assert _rel(wg_.cpu(), w - lr * want_v) < 1e-6
        assert float(accg.abs().max().cpu()) == 0.0
    src = rng.normal(0, 1, (5, 7, 3)).astype(np.float32)
    got = ops.copy_pad_channels(_t(torch, src), 4).cpu().numpy()
    assert np.array_equal(got[..., :3], src) and np.all(got[..., 3] == 0)
    back = ops.copy_pad_channels(_t(torch, got), 3).cpu().numpy()
    assert np.array_equal(back, src)


@pytest.mark.parametrize("n,n_decay", [(64 * 1024 + 512, 64 * 1024), (1027, 1001), (8, 8), (4096, 0)])
def test_fused_accumulate_momentum_is_the_two_kernels_bit_for_bit(env, n, n_decay):
    """emadapt_sgd_accumulate_momentum (accum_num == 1: accumulate + apply_gradients + clean in one pass, deeplab.py:202-205) must
    leave the weights and the momentum exactly as emadapt_sgd_accumulate followed by emadapt_sgd_momentum do on a zero accumulator --
    vector and scalar forms, with and without decayed elements, and negative zeros in the gradient."""
    ops, do, torch = env
    g = torch.Generator(device="cuda").manual_seed(n)
    w = torch.randn(n, device="cuda", generator=g)
    v = torch.randn(n, device="cuda", generator=g) * 0.1
    grad = torch.randn(n, device="cuda", generator=g) * 1e-2
    grad[::7] = -0.0
    grad[3::11] = 0.0
    for gscale, ascale in ((1.0, 1.0), (1.0, 0.5), (0.25, 1.0)):
        w1, v1, acc = w.clone(), v.clone(), torch.zeros(n, device="cuda")
        ops.sgd_accumulate(acc, grad, w1, n_decay, 5e-4, gscale)
        ops.sgd_momentum(w1, v1, acc, 1e-3, 0.9, ascale)
        w2, v2 = w.clone(), v.clone()
        ops.sgd_accumulate_momentum(w2, v2, grad, n_decay, 5e-4, gscale, 1e-3, 0.9, ascale)
        assert torch.equal(w1, w2) and torch.equal(v1.view(torch.int32), v2.view(torch.int32)), (gscale, ascale)


@pytest.mark.parametrize("payload", ["fp32", "bf16"])
@pytest.mark.parametrize("with_accum", [True, False])
def test_momentum_on_the_payload_that_writes_the_tiled_images(env, payload, with_accum):
    """emadapt_sgd_momentum_payload_tiles: weights, momentum and accumulator bit for bit emadapt_sgd_momentum_payload's, the images
    emadapt_filter_tiles_bf16's on the new weights."""
    ops, do, torch = env
    pdt = torch.bfloat16 if payload == "bf16" else torch.float32
    for shape, nb in (((3, 3, 256, 256), 256), ((1, 1, 512, 768), 776)):
        nw = shape[0] * shape[1] * shape[2] * shape[3]
        n = nw + nb
        g = torch.Generator(device="cuda").manual_seed(n + 3)
        w = torch.randn(n, device="cuda", generator=g) * 0.05
        v = torch.randn(n, device="cuda", generator=g) * 0.01
        pay = (torch.randn(n, device="cuda", generator=g) * 1e-2).to(pdt)
        pay[::9] = -0.0
        w1, v1, a1 = w.clone(), v.clone(), (torch.ones(n, device="cuda") if with_accum else None)
        ops.sgd_momentum_payload(w1, v1, a1, pay, 1e-3, 0.9, 0.5)
        tf1 = ops.filter_tiles_bf16(w1[:nw].view(shape))
        td1 = ops.filter_tiles_bf16(w1[:nw].view(shape), dgrad=True)
        w2, v2, a2 = w.clone(), v.clone(), (torch.ones(n, device="cuda") if with_accum else None)
        tf2, td2 = torch.full_like(tf1, 7.0), torch.full_like(td1, 7.0)
        ops.sgd_momentum_payload_tiles(w2, v2, a2, pay, shape, 1e-3, 0.9, 0.5, tf2, td2)
        assert torch.equal(w1.view(torch.int32), w2.view(torch.int32)) and torch.equal(v1.view(torch.int32), v2.view(torch.int32))
        assert torch.equal(tf1.view(torch.int16), tf2.view(torch.int16)) and torch.equal(td1.view(torch.int16), td2.view(torch.int16))
        if with_accum:
            assert float(a2.abs().max()) == 0.0 and float(a1.abs().max()) == 0.0


def test_optimizer_pass_over_several_segments_in_one_launch(env):
    """emadapt_sgd_accumulate_momentum_multi: bit for bit one emadapt_sgd_accumulate_momentum call per segment; what lies between the
    segments is not touched; ragged or overlapping segments are refused."""
    ops, do, torch = env
    n = 300000
    g = torch.Generator(device="cuda").manual_seed(77)
    w = torch.randn(n, device="cuda", generator=g)
    v = torch.randn(n, device="cuda", generator=g) * 0.1
    grad = torch.randn(n, device="cuda", generator=g) * 1e-2
    grad[::5] = -0.0
    segs = [(0, 1728 + 64, 1728), (4096, 36864 + 64, 36864), (50000, 8, 4), (60000, 0, 0), (100000, 150000, 149996), (290000, 4, 0)]
    segs += [(260000 + 16 * k, 8, 8) for k in range(10)]          # 16 segments in all
    assert len(segs) == ops.SGD_MAX_SEGS
    w1, v1 = w.clone(), v.clone()
    for s, m, nd in segs:
        if m:
            ops.sgd_accumulate_momentum(w1[s:s + m], v1[s:s + m], grad[s:s + m], nd, 5e-4, 0.5, 1e-3, 0.9, 0.25)
    w2, v2 = w.clone(), v.clone()
    ops.sgd_accumulate_momentum_multi(w2, v2, grad, segs, 5e-4, 0.5, 1e-3, 0.9, 0.25)
    assert torch.equal(w1.view(torch.int32), w2.view(torch.int32)) and torch.equal(v1.view(torch.int32), v2.view(torch.int32))
    assert not torch.equal(w2, w) and torch.equal(w2[2000:4096], w[2000:4096])
    for bad in ([(0, 10, 8)], [(2, 8, 8)], [(0, 8, 6)], [(0, 64, 64), (32, 64, 64)], [(0, 8, 12)], [(8 * k, 8, 8) for k in range(17)]):
        with pytest.raises(Exception):
            ops.sgd_accumulate_momentum_multi(w2, v2, grad, bad, 5e-4, 0.5, 1e-3, 0.9, 0.25)
    assert torch.equal(w1.view(torch.int32), w2.view(torch.int32))


@pytest.mark.parametrize("shape,nb", [((1, 1, 256, 256), 256), ((3, 3, 256, 512), 512), ((2, 2, 512, 256), 264), ((1, 1, 256, 768), 0)])
def test_optimizer_pass_that_writes_the_tiled_images(env, shape, nb):
    """emadapt_sgd_accumulate_momentum_tiles: weights and momentum bit for bit emadapt_sgd_accumulate_momentum's (filter decayed, biases
    not), both tiled bf16 images bit for bit emadapt_filter_tiles_bf16_both's on the new weights; unsupported shapes are refused."""
    ops, do, torch = env
    KH, KW, Cin, Cout = shape
    assert ops.sgd_accumulate_momentum_tiles_supported(*shape)
    assert not ops.sgd_accumulate_momentum_tiles_supported(3, 3, 128, 256) and not ops.sgd_accumulate_momentum_tiles_supported(3, 3, 256, 320)
    nw = KH * KW * Cin * Cout
    n = nw + nb
    g = torch.Generator(device="cuda").manual_seed(n)
    w = torch.randn(n, device="cuda", generator=g) * 0.05
    v = torch.randn(n, device="cuda", generator=g) * 0.01
    grad = torch.randn(n, device="cuda", generator=g) * 1e-2
    grad[::7] = -0.0
    for gscale, ascale in ((1.0, 1.0), (0.25, 0.5)):
        w1, v1 = w.clone(), v.clone()
        ops.sgd_accumulate_momentum(w1, v1, grad, nw, 5e-4, gscale, 1e-3, 0.9, ascale)
        tf1 = ops.filter_tiles_bf16(w1[:nw].view(shape))
        td1 = ops.filter_tiles_bf16(w1[:nw].view(shape), dgrad=True)
        w2, v2 = w.clone(), v.clone()
        tf2, td2 = torch.full_like(tf1, 7.0), torch.full_like(td1, 7.0)
        ops.sgd_accumulate_momentum_tiles(w2, v2, grad, shape, 5e-4, gscale, 1e-3, 0.9, ascale, tf2, td2)
        assert torch.equal(w1.view(torch.int32), w2.view(torch.int32)) and torch.equal(v1.view(torch.int32), v2.view(torch.int32))
        assert torch.equal(tf1.view(torch.int16), tf2.view(torch.int16)) and torch.equal(td1.view(torch.int16), td2.view(torch.int16))
    with pytest.raises(Exception):
        ops.sgd_accumulate_momentum_tiles(w[:3 * 3 * 128 * 256], v[:3 * 3 * 128 * 256], grad[:3 * 3 * 128 * 256], (3, 3, 128, 256), 5e-4, 1.0, 1e-3, 0.9,
                                          1.0, tf2, td2)


@pytest.mark.parametrize("n,n_decay", [(64 * 1024 + 512, 64 * 1024), (1027, 1001), (8, 8), (4096, 0), (4099, 4096)])
@pytest.mark.parametrize("payload", ["fp32", "bf16"])
def test_accumulate_pack_is_accumulate_then_pack_bit_for_bit(env, n, n_decay, payload):
    """emadapt_sgd_accumulate_pack (accum_num == 1 in a data-parallel job: backward feeds the exchange without an accumulator) must produce
    exactly what emadapt_sgd_accumulate leaves in a zeroed accumulator and emadapt_grad_pack then rounds; and emadapt_sgd_momentum_payload
    without an accumulator must update weights and momentum exactly as with one."""
    ops, do, torch = env
    pdt = torch.bfloat16 if payload == "bf16" else torch.float32
    g = torch.Generator(device="cuda").manual_seed(n + 17)
    w = torch.randn(n, device="cuda", generator=g)
    v = torch.randn(n, device="cuda", generator=g) * 0.1
    grad = torch.randn(n, device="cuda", generator=g) * torch.exp(torch.randn(n, device="cuda", generator=g) * 2) * 1e-2
    grad[::7] = -0.0
    grad[3::11] = 0.0
    for gscale in (1.0, 0.125, 1.0 / 3.0):
        acc = torch.zeros(n, device="cuda")
        ops.sgd_accumulate(acc, grad, w, n_decay, 5e-4, gscale)
        want = ops.grad_pack(acc, torch.empty(n, dtype=pdt, device="cuda"))
        got = ops.sgd_accumulate_pack(grad, w, n_decay, 5e-4, gscale, torch.full((n,), float("nan"), dtype=pdt, device="cuda"))
        bits = torch.int16 if payload == "bf16" else torch.int32
        assert torch.equal(got.view(bits), want.view(bits)), gscale
        w1, v1 = w.clone(), v.clone()
        ops.sgd_momentum_payload(w1, v1, acc, want, 1e-3, 0.9, 0.5)
        w2, v2 = w.clone(), v.clone()
        ops.sgd_momentum_payload(w2, v2, None, got, 1e-3, 0.9, 0.5)
        assert torch.equal(w1, w2) and torch.equal(v1.view(torch.int32), v2.view(torch.int32)) and not bool(acc.any())


@pytest.mark.parametrize("world,n", [(2, 4096), (3, 4099), (8, 1476), (8, 32768 + 8), (8, 65536), (5, 7)])
@pytest.mark.parametrize("payload", ["fp32", "bf16"])
def test_gradient_exchange_staging_kernels(env, world, n, payload):
    """The three kernels of the data-parallel exchange (dist.py algo "direct"; deeplab.py:202 with micro-batches mapped to ranks)
    against the arithmetic model the gloo tests pin on CPU tensors: pack = one RNE rounding + zero padding to whole shards; owner
    sum = fp32 adds in rank order, rounded once; Momentum on the gathered payload = Momentum on the accumulator it would have been
    copied into.  Bit for bit, vector and scalar forms, wide dynamic range."""
    ops, do, torch = env
    pdt = torch.bfloat16 if payload == "bf16" else torch.float32
    g = torch.Generator(device="cuda").manual_seed(world * 100003 + n)
    src = torch.randn(n, device="cuda", generator=g) * torch.exp(torch.randn(n, device="cuda", generator=g) * 3)
    npad = -(-n // world) * world
    sh = npad // world
    send = torch.full((npad,), 7.0, dtype=pdt, device="cuda")
    ops.grad_pack(src, send)
    want = torch.zeros(npad, dtype=pdt, device="cuda")
    want[:n] = src.to(pdt)
    assert torch.equal(send.view(torch.int16 if payload == "bf16" else torch.int32), want.view(torch.int16 if payload == "bf16" else torch.int32))
    # owner sum over `world` contributions of wildly different magnitudes
    recv = (torch.randn(world * sh, device="cuda", generator=g) * torch.exp(torch.randn(world * sh, device="cuda", generator=g) * 3)).to(pdt)
    out = torch.empty(sh, dtype=pdt, device="cuda")
    ops.grad_owner_sum(recv, world, out)
    acc = recv[:sh].float()
    for r in range(1, world):
        acc = acc + recv[r * sh:(r + 1) * sh].float()
    assert torch.equal(out.float(), acc.to(pdt).float())
    # Momentum reading the payload
    nw = (n // 4) * 4 if n >= 8 else n
    w = torch.randn(nw, device="cuda", generator=g)
    v = torch.randn(nw, device="cuda", generator=g) * 0.1
    gathered = (torch.randn(npad, device="cuda", generator=g) * 1e-2).to(pdt)
    w1, v1, a1 = w.clone(), v.clone(), gathered[:nw].float().clone()
    ops.sgd_momentum(w1, v1, a1, 1e-3, 0.9, 1.0 / world)
    w2, v2, a2 = w.clone(), v.clone(), torch.ones(nw, device="cuda")
    ops.sgd_momentum_payload(w2, v2, a2, gathered, 1e-3, 0.9, 1.0 / world)
    assert torch.equal(w1, w2) and torch.equal(v1, v2) and float(a2.abs().max()) == 0.0


@pytest.mark.parametrize("payload", ["fp32", "bf16"])
def test_collective_status_mode_travels_with_the_gradient(env, payload):
    """emadapt_status_collective(1), the data-parallel rule (round-5 advisor finding: a rank whose word is raised skipped its update alone
    while its NaN tile went through the exchange and the clean ranks applied it): (a) while the LOCAL word is raised the producers of the
    exchange -- accumulate, accumulate + pack, pack -- emit NaN for everything; (b) the optimiser entries ignore the local word, leave an
    element whose summed gradient is NaN alone (weight and momentum kept, accumulator zeroed), update every other element exactly as
    before and raise STATUS_NAN_GRADIENT; (c) with the mode off nothing changes."""
    ops, do, torch = env
    pdt = torch.bfloat16 if payload == "bf16" else torch.float32
    n, nd = 4096 + 5, 2048   # (vector body + scalar tail)
    g = torch.Generator(device="cuda").manual_seed(77)
    grad = torch.randn(n, device="cuda", generator=g)
    w = torch.randn(n, device="cuda", generator=g)
    v = torch.randn(n, device="cuda", generator=g) * 0.1
    assert ops.status_poll(clear=True, raise_on_error=False) == 0
    assert ops.status_collective(True) is False
    try:
        # (b) clean word, NaNs at a few elements of the summed gradient
        pay = (torch.randn(n, device="cuda", generator=g) * 1e-2).to(pdt)
        bad = torch.tensor([0, 3, 1000, 4095, 4097, n - 1], device="cuda")
        payn = pay.clone()
        payn[bad] = float("nan")
        w0, v0, a0 = w.clone(), v.clone(), torch.ones(n, device="cuda")
        ops.sgd_momentum_payload(w0, v0, a0, pay, 1e-3, 0.9, 0.5)
        assert ops.status_poll(clear=True, raise_on_error=False) == 0
        w1, v1, a1 = w.clone(), v.clone(), torch.ones(n, device="cuda")
        ops.sgd_momentum_payload(w1, v1, a1, payn, 1e-3, 0.9, 0.5)
        keep = torch.ones(n, dtype=torch.bool, device="cuda")
        keep[bad] = False
        assert torch.equal(w1[keep], w0[keep]) and torch.equal(v1[keep], v0[keep]) and float(a1.abs().max()) == 0.0
        assert torch.equal(w1[bad], w[bad]) and torch.equal(v1[bad], v[bad])
        assert ops.status_poll(clear=True, raise_on_error=False) == ops.STATUS_NAN_GRADIENT
        # ... the accumulator form (the ring all-reduce path) alike
        acc = pay.float().clone()
        acc[bad] = float("nan")
        w2, v2 = w.clone(), v.clone()
        ops.sgd_momentum(w2, v2, acc, 1e-3, 0.9, 0.5)
        assert torch.equal(w2, w1) and torch.equal(v2, v1) and float(acc.abs().max()) == 0.0
        assert ops.status_poll(clear=True, raise_on_error=False) == ops.STATUS_NAN_GRADIENT
        # ... and the pass that also writes the tiled images
        shape, nb = (1, 1, 256, 256), 256
        nw = 256 * 256
        pt = (torch.randn(nw + nb, device="cuda", generator=g) * 1e-2).to(pdt)
        pt[5] = float("nan"); pt[nw + 7] = float("nan")
        wt, vt = torch.randn(nw + nb, device="cuda", generator=g), torch.zeros(nw + nb, device="cuda")
        w3, v3, w4, v4 = wt.clone(), vt.clone(), wt.clone(), vt.clone()
        ops.sgd_momentum_payload(w3, v3, None, pt, 1e-3, 0.9, 0.5)
        tf = torch.empty_like(ops.filter_tiles_bf16(w3[:nw].view(shape)))
        td = torch.empty_like(ops.filter_tiles_bf16(w3[:nw].view(shape), dgrad=True))
        ops.sgd_momentum_payload_tiles(w4, v4, None, pt, shape, 1e-3, 0.9, 0.5, tf, td)
        assert torch.equal(w3, w4) and torch.equal(v3, v4) and float(w4[5]) == float(wt[5]) and float(w4[nw + 7]) == float(wt[nw + 7])
        assert torch.equal(tf.view(torch.int16), ops.filter_tiles_bf16(w4[:nw].view(shape)).view(torch.int16))
        assert ops.status_poll(clear=True, raise_on_error=False) == ops.STATUS_NAN_GRADIENT
        # (a) raised word: everything the producers emit is NaN; the consumers do NOT look at the local word
        ops.status_raise(ops.STATUS_STREAMK_TIMEOUT)
        accum = torch.zeros(n, device="cuda")
        ops.sgd_accumulate(accum, grad, w, nd, 1e-3, 1.0)
        assert bool(torch.isnan(accum).all())
        out = torch.zeros(n, dtype=pdt, device="cuda")
        ops.sgd_accumulate_pack(grad, w, nd, 1e-3, 1.0, out)
        assert bool(torch.isnan(out.float()).all())
        send = torch.zeros(n + 3, dtype=pdt, device="cuda")
        ops.grad_pack(grad, send)
        assert bool(torch.isnan(send[:n].float()).all())
        w5, v5 = w.clone(), v.clone()
        ops.sgd_momentum_payload(w5, v5, None, pay, 1e-3, 0.9, 0.5)   # a finite summed gradient is applied whatever the local word says
        assert torch.equal(w5, w0) and torch.equal(v5, v0)
        w6, v6 = w.clone(), v.clone()
        ops.sgd_momentum_payload(w6, v6, None, out, 1e-3, 0.9, 0.5)   # an all-NaN one leaves everything alone
        assert torch.equal(w6, w) and torch.equal(v6, v)
        assert ops.status_poll(clear=True, raise_on_error=False) == ops.STATUS_STREAMK_TIMEOUT | ops.STATUS_NAN_GRADIENT
    finally:
        assert ops.status_collective(False) is True
        ops.status_poll(clear=True, raise_on_error=False)
    # (c) mode off: the producers ignore the word, the consumers skip everything while it is raised (the single-process rule)
    ops.status_raise(ops.STATUS_STREAMK_TIMEOUT)
    accum = torch.zeros(n, device="cuda")
    ops.sgd_accumulate(accum, grad, w, nd, 1e-3, 1.0)
    assert bool(torch.isfinite(accum).all())
    w7, v7 = w.clone(), v.clone()
    ops.sgd_momentum_payload(w7, v7, None, pay, 1e-3, 0.9, 0.5)
    assert torch.equal(w7, w) and torch.equal(v7, v)
    assert ops.status_poll(clear=True, raise_on_error=False) == ops.STATUS_STREAMK_TIMEOUT


def test_profiler_counts_launches(env):
    ops, do, torch = env
    ops.prof_reset()
    ops.prof_enable(True)
    x = torch.zeros((1, 8, 8, 8), device="cuda:0")
    w = torch.zeros((3, 3, 8, 8), device="cuda:0")
    ops.conv_fwd(x, w, None, 1)
    ops.conv_dgrad(x, w, 1)
    torch.cuda.synchronize()
    r = ops.prof_collect()
    ops.prof_enable(False)
    ops.prof_reset()
    assert r["igemm_fwd"]["launches"] == 1 and r["igemm_dgrad"]["launches"] == 1
    assert r["igemm_fwd"]["flops"] == 2.0 * 64 * 72 * 8 and r["igemm_fwd"]["ms"] > 0
