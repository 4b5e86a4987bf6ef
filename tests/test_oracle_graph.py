"""The graph half of the oracle (oracle/deeplab_oracle.py) cross-checked against a SECOND, independent restatement.

The reference's conv / pool / loss / optimiser arithmetic is TensorFlow 1.x's, which is absent here and holds no fixtures (SURVEY 4,
8c): that half of the oracle is "parity unpinned" and nothing in this file changes that.  What it does is catch a slip of the oracle
itself: the oracle expresses TF's documented semantics with PyTorch-CPU operators (F.conv2d on an explicitly padded tensor,
F.max_pool2d on a -inf-padded tensor, autograd for every gradient); below, the same semantics are written out as plain NumPy loops
straight from the definitions -- SAME padding of stride-1 (dilated) convolutions incl. the EVEN 4x4 rate-4 kernel of fc6
(deeplab.py:92: taps at -6, -2, +2, +6), 3x3 max-pool SAME at stride 2 and 1 with padded cells ignored (deeplab.py:74-83), MaxPoolGrad
to the first maximum of a window, the gradients of the convolution, sparse softmax cross-entropy averaged over pixels + weight_decay x
sum(w^2)/2 (deeplab.py:180-186) and non-Nesterov Momentum (deeplab.py:190,204-205) -- and the two must agree on small random cases
(fp64, 1e-12).  CPU only."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from oracle import deeplab_oracle as do  # noqa: E402


def conv_same_loops(x, w, b, rate):
    """tf.nn.conv2d / atrous_conv2d, stride 1, padding SAME [TF-doc]: out = in; effective kernel ke = (k - 1) rate + 1; total padding
    ke - 1, of which floor((ke - 1) / 2) goes in front; cross-correlation, HWIO filter, zero padding."""
    B, H, W, Cin = x.shape
    KH, KW, _, Cout = w.shape
    ph, pw = ((KH - 1) * rate) // 2, ((KW - 1) * rate) // 2
    y = np.zeros((B, H, W, Cout))
    for kh in range(KH):
        for kw in range(KW):
            dh, dw = kh * rate - ph, kw * rate - pw
            for oh in range(H):
                ih = oh + dh
                if not 0 <= ih < H:
                    continue
                for ow in range(W):
                    iw = ow + dw
                    if 0 <= iw < W:
                        y[:, oh, ow, :] += x[:, ih, iw, :] @ w[kh, kw]
    return y + b


def conv_grads_loops(x, w, dy, rate):
    B, H, W, Cin = x.shape
    KH, KW, _, Cout = w.shape
    ph, pw = ((KH - 1) * rate) // 2, ((KW - 1) * rate) // 2
    dx, dw = np.zeros_like(x), np.zeros_like(w)
    for kh in range(KH):
        for kw in range(KW):
            dh, dw_ = kh * rate - ph, kw * rate - pw
            for oh in range(H):
                ih = oh + dh
                if not 0 <= ih < H:
                    continue
                for ow in range(W):
                    iw = ow + dw_
                    if 0 <= iw < W:
                        dx[:, ih, iw, :] += dy[:, oh, ow, :] @ w[kh, kw].T
                        dw[kh, kw] += x[:, ih, iw, :].T @ dy[:, oh, ow, :]
    return dx, dw, dy.sum((0, 1, 2))


def pool_windows(n, stride):
    """tf.nn.max_pool ksize 3, SAME [TF-doc]: out = ceil(n / stride); total padding max((out - 1) stride + 3 - n, 0), floor(half) in front."""
    out = -(-n // stride)
    total = max((out - 1) * stride + 3 - n, 0)
    before = total // 2
    return [[i for i in range(o * stride - before, o * stride - before + 3) if 0 <= i < n] for o in range(out)]


def maxpool_loops(x, stride):
    B, H, W, C = x.shape
    rows, cols = pool_windows(H, stride), pool_windows(W, stride)
    y = np.empty((B, len(rows), len(cols), C))
    arg = np.empty((B, len(rows), len(cols), C, 2), np.int64)
    for oh, rr in enumerate(rows):
        for ow, cc in enumerate(cols):
            best = np.full((B, C), -np.inf)
            bi = np.zeros((B, C, 2), np.int64)
            for r in rr:            # window scanned row-major: the FIRST maximum keeps the gradient (MaxPoolGrad)
                for c in cc:
                    v = x[:, r, c, :]
                    take = v > best
                    best = np.where(take, v, best)
                    bi[take] = (r, c)
            y[:, oh, ow, :] = best
            arg[:, oh, ow] = bi
    return y, arg


def maxpool_grad_loops(x_shape, arg, dy):
    dx = np.zeros(x_shape)
    B, OH, OW, C = dy.shape
    for b in range(B):
        for oh in range(OH):
            for ow in range(OW):
                for c in range(C):
                    r, q = arg[b, oh, ow, c]
                    dx[b, r, q, c] += dy[b, oh, ow, c]
    return dx


@pytest.mark.parametrize("shape", [(2, 9, 7, 3, 5, 3, 1), (1, 11, 11, 4, 6, 3, 2), (2, 13, 10, 3, 4, 4, 4), (1, 7, 9, 5, 3, 1, 1),
                                   (1, 15, 15, 2, 3, 3, 12), (1, 6, 5, 2, 2, 4, 1)])
def test_conv_same_forward_and_gradients(shape):
    B, H, W, Cin, Cout, k, rate = shape
    rng = np.random.default_rng(sum(shape))
    x, w = rng.normal(0, 1, (B, H, W, Cin)), rng.normal(0, 1, (k, k, Cin, Cout))
    b, dy = rng.normal(0, 1, Cout), rng.normal(0, 1, (B, H, W, Cout))
    xt = torch.tensor(x, requires_grad=True)
    wt = torch.tensor(w, requires_grad=True)
    bt = torch.tensor(b, requires_grad=True)
    y = do.conv_same(xt, wt, bt, rate)
    y.backward(torch.tensor(dy))
    assert np.abs(y.detach().numpy() - conv_same_loops(x, w, b, rate)).max() < 1e-12
    dx, dw, db = conv_grads_loops(x, w, dy, rate)
    assert np.abs(xt.grad.numpy() - dx).max() < 1e-12 and np.abs(wt.grad.numpy() - dw).max() < 1e-12
    assert np.abs(bt.grad.numpy() - db).max() < 1e-12


def test_fc6_kernel_taps_sit_at_minus6_minus2_plus2_plus6():
    """The even 4x4 rate-4 kernel (deeplab.py:92,145): SAME puts 6 in front and 6 behind, so the taps read -6, -2, +2, +6."""
    x = np.zeros((1, 21, 21, 1))
    x[0, 10, 10, 0] = 1.0
    w = np.arange(1, 17, dtype=np.float64).reshape(4, 4, 1, 1)
    y = do.conv_same(torch.tensor(x), torch.tensor(w), torch.zeros(1, dtype=torch.float64), 4).numpy()[0, :, :, 0]
    hits = {(int(r), int(c)): y[r, c] for r, c in zip(*np.nonzero(y))}
    # output (oh, ow) sees the impulse through tap (kh, kw) when oh + 4 kh - 6 == 10, i.e. oh = 16 - 4 kh
    assert hits == {(16 - 4 * kh, 16 - 4 * kw): w[kh, kw, 0, 0] for kh in range(4) for kw in range(4)}


@pytest.mark.parametrize("H,W,stride", [(9, 9, 2), (10, 7, 2), (5, 6, 1), (321, 3, 2), (41, 2, 1)])
def test_maxpool_same_and_its_gradient(H, W, stride):
    rng = np.random.default_rng(H * 31 + W + stride)
    # values on a coarse grid: ties inside windows are common (the tie-break of the gradient is part of the semantics)
    x = np.round(rng.normal(0, 1, (2, H, W, 3)) * 2) / 2
    dy_shape = (2, -(-H // stride), -(-W // stride), 3)
    dy = rng.normal(0, 1, dy_shape)
    xt = torch.tensor(x, requires_grad=True)
    y = do.maxpool_same(xt, stride)
    assert tuple(y.shape) == dy_shape
    want, arg = maxpool_loops(x, stride)
    assert np.array_equal(y.detach().numpy(), want)
    y.backward(torch.tensor(dy))
    assert np.abs(xt.grad.numpy() - maxpool_grad_loops(x.shape, arg, dy)).max() < 1e-12


def test_pool_geometry_of_the_reference_sizes():
    """321 -> 161 -> 81 -> 41 (stride 2, one padded cell on each side), then 41 -> 41 twice (stride 1): deeplab.py:74-83."""
    n = 321
    for _ in range(3):
        wins = pool_windows(n, 2)
        assert len(wins) == -(-n // 2) and wins[0] == [0, 1] and wins[-1] == [n - 2, n - 1]      # total padding 2: 1 in front, 1 behind
        n = len(wins)
    assert n == 41 and len(pool_windows(41, 1)) == 41 and pool_windows(41, 1)[0] == [0, 1]
    assert [len(pool_windows(513, 2)), len(pool_windows(257, 2)), len(pool_windows(129, 2))] == [257, 129, 65]


def test_loss_gradient_and_momentum_of_a_whole_step():
    """One micro-step of a narrow net through the oracle against the definitions: norm = mean CE over pixels, total = norm + wd *
    sum_w w^2 / 2 (biases excluded, deeplab.py:184), d norm / d logits = (softmax - onehot) / (B HW), and the Momentum update."""
    from oracle import estep_oracle as eo
    C, size, B = 4, 17, 2
    oracle = do.DeepLabOracle(category_num=C, width=2, dtype=torch.float64)
    params = do.init_params(oracle.specs, seed=3, he=True)
    images, labels = do.synthetic_batch(B, size, C, seed=1)
    images = images / 50
    perms = eo.draw_perms(5, C, seed=2)
    wd = 1e-2
    r = oracle.micro_step(params, images, labels, perms, weight_decay=wd)
    logits, lab = r["logits"].astype(np.float64), r["labels"]
    z = logits - logits.max(-1, keepdims=True)
    logp = z - np.log(np.exp(z).sum(-1, keepdims=True))
    ce = -np.take_along_axis(logp, lab[..., None], -1)[..., 0]
    assert abs(r["loss_norm"] - ce.mean()) < 1e-6             # (the logits in r are rounded to fp32 for the E-step)
    l2 = sum((w.astype(np.float64) ** 2).sum() / 2 for (w, b) in params.values())
    assert abs(r["loss_l2"] - l2) < 1e-9 * max(l2, 1)
    # fc8: dL/db = sum over pixels of (softmax - onehot) / (B HW); weights carry the extra wd * w
    p = np.exp(logp)
    onehot = np.eye(C)[lab]
    db = ((p - onehot) / ce.size).sum((0, 1, 2))
    assert np.abs(r["grads"]["fc8"][1] - db).max() < 1e-6
    # Momentum, non-Nesterov (deeplab.py:190,204-205): v = mu v + g, w -= lr v; the accumulators are zeroed by the clean op
    plist = {k: [w.astype(np.float32).copy(), b.astype(np.float32).copy()] for k, (w, b) in params.items()}
    vel = {k: [np.full_like(w, 0.5, np.float32), np.zeros_like(b, np.float32)] for k, (w, b) in params.items()}
    acc = {k: [g[0].astype(np.float32).copy(), g[1].astype(np.float32).copy()] for k, g in r["grads"].items()}
    w0, g0 = plist["fc7"][0].copy(), acc["fc7"][0].copy()
    do.momentum_update(plist, vel, acc, 1e-2, 0.9)
    v1 = np.float32(0.9) * np.float32(0.5) + g0
    assert np.allclose(vel["fc7"][0], v1, rtol=0, atol=1e-7) and np.allclose(plist["fc7"][0], w0 - np.float32(1e-2) * v1, rtol=0, atol=1e-7)
    assert not acc["fc7"][0].any()
