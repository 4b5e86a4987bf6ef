"""GPU parity of the bf16-compute convolution kernels (bf16 compute, fp32 accumulate).

The kernels round every operand of an inner product to bf16 (round-to-nearest-even) and then multiply/accumulate in
fp32.  bf16 x bf16 products are exact in fp32, so against an fp64 oracle fed with the SAME rounded operands the only
error left is the fp32 summation: the bar is 2e-5 of the tensor's max magnitude, as for the fp32 kernels.  A second,
looser check against the unrounded fp64 result documents the bf16 quantisation itself (about 2^-8 per operand).
All calls go through the C ABI.
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


def _l2(got, want):
    want = np.asarray(want, dtype=np.float64)
    got = np.asarray(got, dtype=np.float64)
    return np.linalg.norm(got - want) / max(np.linalg.norm(want), 1e-30)


def _t(torch, a, dtype=None):
    return torch.tensor(np.ascontiguousarray(a), dtype=dtype or torch.float32, device="cuda:0")


def _bf(torch, a):
    """fp32 array rounded to bf16 (RNE), back as fp64 tensor."""
    return torch.tensor(a, dtype=torch.float32).bfloat16().double()


# (B, H, W, Cin, Cout, k, dil): channel counts that are / are not multiples of 8 (4 -> padded RGB, 12, 44 -> 41 classes),
# Cin % 64 == 0 (whole K step inside one tap) and not, N <= 64 and N > 128, K and M tails, long-K split-K shapes.
SHAPES = [
    (1, 7, 9, 4, 8, 3, 1),
    (2, 13, 11, 8, 16, 3, 1),
    (2, 12, 12, 12, 64, 3, 2),
    (1, 19, 17, 16, 136, 3, 1),
    (2, 11, 11, 32, 160, 4, 4),
    (1, 25, 25, 24, 44, 3, 12),
    (3, 9, 9, 160, 24, 1, 1),
    (1, 33, 31, 4, 64, 3, 1),
    (2, 23, 23, 64, 256, 3, 2),
    (1, 41, 41, 24, 288, 1, 1),
    (2, 49, 47, 256, 128, 3, 1),
    (1, 41, 41, 512, 136, 4, 4),
    (1, 41, 41, 128, 44, 1, 1),
]


@pytest.fixture(params=["auto", "128", "256"])
def tile(request):
    """Run the forward / data-gradient kernel in its automatic plan and with each tile configuration forced."""
    import em_adapt_tensorflow_amd.ops as ops
    ops.tune_set("bf16_tile", 0 if request.param == "auto" else int(request.param))
    yield request.param
    ops.tune_set("bf16_tile", 0)


@pytest.mark.parametrize("shape", SHAPES)
def test_bf16_conv_fwd_dgrad_wgrad(env, shape, tile):
    ops, do, torch = env
    B, H, W, Cin, Cout, k, dil = shape
    rng = np.random.default_rng(hash(shape) % (1 << 31))
    x = rng.normal(0, 1, (B, H, W, Cin)).astype(np.float32)
    w = rng.normal(0, 0.2, (k, k, Cin, Cout)).astype(np.float32)
    b = rng.normal(0, 0.5, (Cout,)).astype(np.float32)
    dy = rng.normal(0, 1, (B, H, W, Cout)).astype(np.float32)

    # oracle on bf16-rounded operands, fp64 arithmetic
    xr = _bf(torch, x).requires_grad_(True)
    wr = _bf(torch, w).requires_grad_(True)
    bt = torch.tensor(b, dtype=torch.float64)
    y = do.conv_same(xr, wr, bt, dil)
    y.backward(_bf(torch, dy))
    # and on the unrounded operands (quantisation check)
    y_full = do.conv_same(torch.tensor(x, dtype=torch.float64), torch.tensor(w, dtype=torch.float64), bt, dil)

    xg, wg, bg, dyg = _t(torch, x), _t(torch, w), _t(torch, b), _t(torch, dy)
    wf, wd = ops.weights_to_bf16(wg)
    # the filter images are the RNE-rounded filter, re-laid-out and zero padded
    wb = wg.bfloat16().reshape(k * k, Cin, Cout)
    assert torch.equal(wd[:, :, :Cout], wb)
    assert torch.equal(wf[:, :, :Cin], wb.permute(0, 2, 1))
    assert not wd[:, :, Cout:].any() and not wf[:, :, Cin:].any()

    got = ops.conv_fwd_bf16(xg, wf, (k, k), bg, dil, relu=False).cpu()
    assert _rel(got, y.detach()) < 2e-5
    assert _rel(got, y_full) < 2e-2
    assert _rel(ops.conv_fwd_bf16(xg, wf, (k, k), None, dil).cpu(), (y - bt).detach()) < 2e-5
    assert _rel(ops.conv_fwd_bf16(xg, wf, (k, k), bg, dil, relu=True).cpu(), torch.relu(y).detach()) < 2e-5

    assert _rel(ops.conv_dgrad_bf16(dyg, wd, (k, k), dil).cpu(), xr.grad) < 2e-5
    gate = rng.normal(0, 1, (B, H, W, Cin)).astype(np.float32)
    want = np.where(gate > 0, xr.grad.numpy() * 2.0, 0.0)
    assert _rel(ops.conv_dgrad_bf16(dyg, wd, (k, k), dil, gate=_t(torch, gate), gate_scale=2.0).cpu(), want) < 2e-5

    assert _rel(ops.conv_wgrad_bf16(xg, dyg, k, k, dil).cpu(), wr.grad) < 2e-5


def _within_bf16_ulp(got, want, slack=1e-5):
    """|got - want| <= one bf16 ulp of want (2^-8 relative) + slack * max|want|: what a correctly rounded bf16 store of
    an fp32-accumulated value can differ from the fp64 result by."""
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    return bool(np.all(np.abs(got - want) <= np.abs(want) * 2.0 ** -8 + slack * np.abs(want).max()))


@pytest.mark.parametrize("shape", [sh for sh in SHAPES if sh[3] % 8 == 0 and sh[4] % 8 == 0])
def test_bf16_storage_variants(env, shape, tile):
    """Activations / gradients stored as bf16 (include/emadapt.h "Storage types"): bf16 in -> same sums as f32 in holding
    the same values; bf16 out -> the f32 result rounded once."""
    ops, do, torch = env
    B, H, W, Cin, Cout, k, dil = shape
    rng = np.random.default_rng(hash(shape) % (1 << 31) + 1)
    x = _t(torch, rng.normal(0, 1, (B, H, W, Cin)).astype(np.float32)).bfloat16()
    w = _t(torch, rng.normal(0, 0.2, (k, k, Cin, Cout)).astype(np.float32))
    b = _t(torch, rng.normal(0, 0.5, (Cout,)).astype(np.float32))
    dy = _t(torch, rng.normal(0, 1, (B, H, W, Cout)).astype(np.float32)).bfloat16()
    gate = _t(torch, rng.normal(0, 1, (B, H, W, Cin)).astype(np.float32)).bfloat16()
    wf, wd = ops.weights_to_bf16(w)
    # bf16-stored inputs, f32 outputs: bit-identical to the f32-stored path on the same values
    y32 = ops.conv_fwd_bf16(x.float(), wf, (k, k), b, dil, relu=True)
    assert torch.equal(ops.conv_fwd_bf16(x, wf, (k, k), b, dil, relu=True), y32)
    dx32 = ops.conv_dgrad_bf16(dy.float(), wd, (k, k), dil, gate=gate.float(), gate_scale=2.0)
    assert torch.equal(ops.conv_dgrad_bf16(dy, wd, (k, k), dil, gate=gate.float(), gate_scale=2.0), dx32)
    dw32 = ops.conv_wgrad_bf16(x.float(), dy.float(), k, k, dil)
    assert torch.equal(ops.conv_wgrad_bf16(x, dy, k, k, dil), dw32)
    assert torch.equal(ops.conv_wgrad_bf16(x.float(), dy, k, k, dil), dw32)
    assert torch.equal(ops.conv_wgrad_bf16(x, dy.float(), k, k, dil), dw32)
    # bf16 outputs: the same values rounded to nearest-even
    assert torch.equal(ops.conv_fwd_bf16(x, wf, (k, k), b, dil, relu=True, out_dtype=torch.bfloat16), y32.bfloat16())
    assert torch.equal(ops.conv_dgrad_bf16(dy, wd, (k, k), dil, gate=gate, gate_scale=2.0), dx32.bfloat16())
    # and against the fp64 oracle
    xr = x.cpu().double().requires_grad_(True)
    wr = w.bfloat16().cpu().double().requires_grad_(True)
    yo = do.conv_same(xr, wr, b.cpu().double(), dil)
    yo.backward(dy.cpu().double())
    got = ops.conv_fwd_bf16(x, wf, (k, k), b, dil, relu=False, out_dtype=torch.bfloat16).float().cpu()
    assert _within_bf16_ulp(got, yo.detach())
    assert _within_bf16_ulp(ops.conv_dgrad_bf16(dy, wd, (k, k), dil, out_dtype=torch.bfloat16).float().cpu(), xr.grad)
    assert _rel(dw32.cpu(), wr.grad) < 2e-5


def test_bf16_elementwise_kernels(env):
    """max-pool forward/backward, bias gradient and the softmax-CE gradient on bf16-stored tensors."""
    ops, do, torch = env
    rng = np.random.default_rng(21)
    for (B, H, W, C, stride) in [(2, 21, 21, 8, 2), (1, 16, 18, 16, 1), (2, 11, 13, 24, 2)]:
        x = _t(torch, rng.normal(0, 1, (B, H, W, C)).astype(np.float32)).bfloat16()
        y, idx = ops.maxpool_fwd(x, stride)
        y32, idx32 = ops.maxpool_fwd(x.float(), stride)
        assert y.dtype == torch.bfloat16 and torch.equal(y.float(), y32) and torch.equal(idx, idx32)
        want = do.maxpool_same(x.cpu().double(), stride)
        assert np.array_equal(y.float().cpu().numpy(), want.numpy())
        dy = _t(torch, rng.normal(0, 1, tuple(y.shape)).astype(np.float32)).bfloat16()
        gate = _t(torch, rng.normal(0, 1, (B, H, W, C)).astype(np.float32)).bfloat16()
        dx = ops.maxpool_bwd(dy, idx, (B, H, W, C), stride, gate=gate)
        dx32 = ops.maxpool_bwd(dy.float(), idx32, (B, H, W, C), stride, gate=gate.float())
        assert dx.dtype == torch.bfloat16 and torch.equal(dx, dx32.bfloat16())
    dyb = _t(torch, rng.normal(0, 1, (3, 17, 19, 40)).astype(np.float32)).bfloat16()
    assert _rel(ops.bias_grad(dyb).cpu(), dyb.cpu().double().sum(dim=(0, 1, 2))) < 2e-5
    logits = _t(torch, rng.normal(0, 2, (2, 9, 9, 24)).astype(np.float32))
    labels = torch.tensor(rng.integers(0, 21, (2, 9, 9)), dtype=torch.int64, device="cuda:0")
    l32, d32 = ops.softmax_ce(logits, labels, 21)
    l16, d16 = ops.softmax_ce(logits, labels, 21, dlogits=torch.empty((2, 9, 9, 24), dtype=torch.bfloat16, device="cuda:0"))
    assert torch.equal(l16, l32) and torch.equal(d16, d32.bfloat16())


def test_bf16_wgrad_split_m_is_deterministic_and_right(env):
    ops, do, torch = env
    rng = np.random.default_rng(5)
    B, H, W, Cin, Cout, k = 2, 97, 95, 8, 64, 3
    x = rng.normal(0, 1, (B, H, W, Cin)).astype(np.float32)
    dy = rng.normal(0, 1, (B, H, W, Cout)).astype(np.float32)
    wt = torch.zeros((k, k, Cin, Cout), dtype=torch.float64, requires_grad=True)
    y = do.conv_same(_bf(torch, x), wt, torch.zeros(Cout, dtype=torch.float64), 1)
    y.backward(_bf(torch, dy))
    xg, dyg = _t(torch, x), _t(torch, dy)
    a = ops.conv_wgrad_bf16(xg, dyg, k, k, 1).cpu().numpy()
    b = ops.conv_wgrad_bf16(xg, dyg, k, k, 1).cpu().numpy()
    assert np.array_equal(a, b), "filter gradient must be bitwise reproducible"
    assert _rel(a, wt.grad) < 2e-5


# (B, H, W), plan_cus: the direct kernel for 3 x 3, 64 -> 64 channels on bf16-stored tensors (csrc/conv_d64.h: conv1_2) -- maps smaller than
# a 16 x 16 tile, ragged tile rows / columns, several tiles per workgroup (plan_cus small: both patch buffers, the persistent loop), one
# workgroup with an odd number of tiles, more workgroups than tiles never happens (the grid is min(tiles, CUs))
D64 = [
    ((1, 5, 7), 256),
    ((2, 16, 16), 256),
    ((1, 33, 47), 3),
    ((3, 17, 50), 2),
    ((1, 70, 9), 1),
    ((2, 40, 40), 5),
]


@pytest.mark.parametrize("shape,cus", D64)
def test_bf16_direct_kernel_64_to_64(env, shape, cus):
    ops, do, torch = env
    B, H, W = shape
    k, C = 3, 64
    rng = np.random.default_rng(hash(shape) % (1 << 31))
    x = torch.tensor(rng.normal(0, 1, (B, H, W, C)).astype(np.float32)).bfloat16()
    w = torch.tensor(rng.normal(0, 0.2, (k, k, C, C)).astype(np.float32))
    b = torch.tensor(rng.normal(0, 0.5, (C,)).astype(np.float32))
    xr = x.double().requires_grad_(True)
    y = do.conv_same(xr, w.bfloat16().double(), b.double(), 1)
    xg, wg, bg = x.cuda(), w.cuda(), b.cuda()
    wf, wd = ops.weights_to_bf16(wg)

    def close(got, want, tol=2.0 ** -8):   # a bf16 result against the exact one: rounding (half an ulp <= 2^-9 relative) + the fp32 sums' noise
        got, want = got.double().cpu(), want.double()
        err = (got - want).abs()
        return bool((err <= want.abs() * tol + 3e-5).all())   # (+ the fp32 sums' absolute noise where the terms cancel)

    with ops.tuned(plan_cus=cus):
        assert ops.conv_bf16_plan(B, H, W, C, C, k, 1) == (4, 16, 16)
        got = ops.conv_fwd_bf16(xg, wf, (k, k), bg, 1, relu=False, out_dtype=torch.bfloat16)
        assert close(got, y.detach())
        assert torch.equal(got, ops.conv_fwd_bf16(xg, wf, (k, k), bg, 1, relu=False, out_dtype=torch.bfloat16)), "bitwise reproducible"
        assert close(ops.conv_fwd_bf16(xg, wf, (k, k), None, 1, out_dtype=torch.bfloat16), (y - b.double()).detach())
        got = ops.conv_fwd_bf16(xg, wf, (k, k), bg, 1, relu=True, out_dtype=torch.bfloat16)
        assert close(got, torch.relu(y.detach()))
        with ops.tuned(bf16_tile=128):   # the implicit-GEMM kernel: the same function, another order of the fp32 sums
            assert ops.conv_bf16_plan(B, H, W, C, C, k, 1)[0] == 128
            old = ops.conv_fwd_bf16(xg, wf, (k, k), bg, 1, relu=True, out_dtype=torch.bfloat16)
        same = (got == old).float().mean().item()
        assert same > 0.98 and close(got, old.double().cpu(), 2.0 ** -7), same   # (two bf16 roundings of sums that differ in their last bits: one ulp)
        # a NaN in the input reaches every output its taps touch and nothing else; the ReLU lets it through
        xn = xg.clone()
        xn[0, H // 2, W // 2, 5] = float("nan")
        yn = ops.conv_fwd_bf16(xn, wf, (k, k), bg, 1, relu=True, out_dtype=torch.bfloat16)
        bad = torch.isnan(yn).any(dim=3)
        assert int(bad.sum()) == len(range(max(H // 2 - 1, 0), min(H // 2 + 2, H))) * len(range(max(W // 2 - 1, 0), min(W // 2 + 2, W)))
        # data gradient + gate
        dy = torch.tensor(rng.normal(0, 1, (B, H, W, C)).astype(np.float32)).bfloat16()
        gate = torch.tensor(rng.normal(0, 1, (B, H, W, C)).astype(np.float32)).bfloat16()
        y.backward(dy.double())
        dyg, gg = dy.cuda(), gate.cuda()
        assert ops.conv_bf16_plan(B, H, W, C, C, k, 1, "dgrad") == (4, 16, 16)
        got = ops.conv_dgrad_bf16(dyg, wd, (k, k), 1, out_dtype=torch.bfloat16)
        assert close(got, xr.grad)
        want = torch.where(gate.double() > 0, xr.grad * 2.0, torch.zeros((), dtype=torch.float64))
        gb = ops.conv_dgrad_bf16(dyg, wd, (k, k), 1, gate=gg, gate_scale=2.0)
        assert gb.dtype == torch.bfloat16 and close(gb, want)
        assert torch.equal(gb, ops.conv_dgrad_bf16(dyg, wd, (k, k), 1, gate=gg, gate_scale=2.0))
        with ops.tuned(bf16_tile=128):
            old = ops.conv_dgrad_bf16(dyg, wd, (k, k), 1, gate=gg, gate_scale=2.0)
        assert (gb == old).float().mean().item() > 0.98 and close(gb, old.double().cpu(), 2.0 ** -7)
        # dropout in the epilogue is not the direct kernel's: the call takes the implicit-GEMM kernel and still answers
        mask = torch.tensor((rng.random((B, H, W, C)) < 0.5).astype(np.uint8)).cuda()
        got = ops.conv_fwd_bf16(xg, wf, (k, k), bg, 1, relu=True, keep_prob=0.5, drop_mask=mask, out_dtype=torch.bfloat16)
        assert close(got, torch.relu(y.detach()) * 2.0 * mask.cpu().double())


@pytest.mark.parametrize("shape", [(1, 5, 7), (2, 16, 16), (1, 33, 47), (3, 17, 50), (1, 70, 9)], ids=lambda s: "%dx%dx%d" % s)
def test_bf16_direct_kernel_image_to_64(env, shape):
    """conv1_1 as the model runs it: the 4-channel f32 image -> 64 bf16 channels, 3 x 3 (csrc/conv_d4.h; one workgroup per 16 x 16 tile)."""
    ops, do, torch = env
    B, H, W = shape
    k, Cin, C = 3, 4, 64
    rng = np.random.default_rng(hash(shape) % (1 << 31))
    x = torch.tensor(rng.normal(0, 50, (B, H, W, Cin)).astype(np.float32))
    x[..., 3] = 0                                                     # (the pad channel, as copy_pad_channels leaves it)
    w = torch.tensor(rng.normal(0, 0.05, (k, k, Cin, C)).astype(np.float32))
    b = torch.tensor(rng.normal(0, 0.5, (C,)).astype(np.float32))
    y = do.conv_same(x.bfloat16().double(), w.bfloat16().double(), b.double(), 1)
    xg, wg, bg = x.cuda(), w.cuda(), b.cuda()
    wf, _ = ops.weights_to_bf16(wg, want_dgrad=False)

    def close(got, want, tol=2.0 ** -8):
        err = (got.double().cpu() - want.double()).abs()
        return bool((err <= want.double().abs() * tol + 1e-3).all())   # (sums of terms up to 150 * 0.15: the fp32 noise floor is ~1e-4)

    assert ops.conv_bf16_plan(B, H, W, Cin, C, k, 1) == (5, 16, 16)
    got = ops.conv_fwd_bf16(xg, wf, (k, k), bg, 1, relu=True, out_dtype=torch.bfloat16)
    assert close(got, torch.relu(y))
    assert torch.equal(got, ops.conv_fwd_bf16(xg, wf, (k, k), bg, 1, relu=True, out_dtype=torch.bfloat16)), "bitwise reproducible"
    assert close(ops.conv_fwd_bf16(xg, wf, (k, k), None, 1, out_dtype=torch.bfloat16), y - b.double())
    with ops.tuned(bf16_tile=128):   # the implicit-GEMM kernel: the same function, another order of the fp32 sums
        assert ops.conv_bf16_plan(B, H, W, Cin, C, k, 1)[0] == 128
        old = ops.conv_fwd_bf16(xg, wf, (k, k), bg, 1, relu=True, out_dtype=torch.bfloat16)
    assert (got == old).float().mean().item() > 0.98 and close(got, old.double().cpu(), 2.0 ** -7)
    # an f32 output, or dropout in the epilogue, is not this kernel's: the call takes the implicit-GEMM kernel and still answers
    y32 = ops.conv_fwd_bf16(xg, wf, (k, k), bg, 1, relu=True, out_dtype=torch.float32)
    assert _rel(y32.cpu(), torch.relu(y)) < 2e-5
    mask = torch.tensor((rng.random((B, H, W, C)) < 0.5).astype(np.uint8)).cuda()
    got = ops.conv_fwd_bf16(xg, wf, (k, k), bg, 1, relu=True, keep_prob=0.5, drop_mask=mask, out_dtype=torch.bfloat16)
    assert close(got, torch.relu(y) * 2.0 * mask.cpu().double())


@pytest.mark.parametrize("shape", [(1, 3, 3, 24, 256), (2, 9, 7, 24, 512), (1, 41, 41, 24, 1024), (3, 5, 5, 8, 256), (1, 13, 11, 32, 768)],
                         ids=lambda s: "%dx%dx%d_%dto%d" % s)
def test_bf16_short_reduction_data_gradient(env, shape):
    """fc8's data gradient as the model runs it: 1 x 1, 24 gathered channels (the padded classes) -> fc7's channels, bf16-stored dy / dx / gate
    (csrc/conv_k32.h: one MFMA step, operands straight from memory); pixel tails inside a 64-pixel tile, 8 - 32 gathered channels."""
    ops, do, torch = env
    B, H, W, Cout, Cin = shape     # (the layer's forward is Cin -> Cout; its data gradient gathers the Cout channels of dy)
    rng = np.random.default_rng(hash(shape) % (1 << 31))
    w = torch.tensor(rng.normal(0, 0.2, (1, 1, Cin, Cout)).astype(np.float32))
    dy = torch.tensor(rng.normal(0, 1, (B, H, W, Cout)).astype(np.float32)).bfloat16()
    gate = torch.tensor(rng.normal(0, 1, (B, H, W, Cin)).astype(np.float32)).bfloat16()
    want = torch.einsum("bhwo,io->bhwi", dy.double(), w[0, 0].bfloat16().double())
    wg, dyg, gg = w.cuda(), dy.cuda(), gate.cuda()
    _, wd = ops.weights_to_bf16(wg, want_fwd=False)

    def close(got, want, tol=2.0 ** -8):
        err = (got.double().cpu() - want.double()).abs()
        return bool((err <= want.double().abs() * tol + 3e-5).all())

    assert ops.conv_bf16_plan(B, H, W, Cin, Cout, 1, 1, "dgrad") == (6, 64, 256)
    got = ops.conv_dgrad_bf16(dyg, wd, (1, 1), 1, out_dtype=torch.bfloat16)
    assert close(got, want)
    wantg = torch.where(gate.double() > 0, want * 2.0, torch.zeros((), dtype=torch.float64))
    gb = ops.conv_dgrad_bf16(dyg, wd, (1, 1), 1, gate=gg, gate_scale=2.0)
    assert gb.dtype == torch.bfloat16 and close(gb, wantg)
    assert torch.equal(gb, ops.conv_dgrad_bf16(dyg, wd, (1, 1), 1, gate=gg, gate_scale=2.0)), "bitwise reproducible"
    with ops.tuned(bf16_tile=128):
        assert ops.conv_bf16_plan(B, H, W, Cin, Cout, 1, 1, "dgrad")[0] != 6
        old = ops.conv_dgrad_bf16(dyg, wd, (1, 1), 1, gate=gg, gate_scale=2.0)
    assert (gb == old).float().mean().item() > 0.98 and close(gb, old.double().cpu(), 2.0 ** -7)
    # a NaN gradient passes the gate on either branch
    dyn = dyg.clone()
    dyn[0, 0, 0, 3] = float("nan")
    gn = ops.conv_dgrad_bf16(dyn, wd, (1, 1), 1, gate=gg, gate_scale=2.0)
    assert bool(torch.isnan(gn[0, 0, 0]).all()) and int(torch.isnan(gn).sum()) == Cin


@pytest.mark.parametrize("shape,cus", D64)
def test_bf16_direct_filter_gradient_64_to_64(env, shape, cus):
    """conv1_2's filter gradient on the direct kernel (csrc/wgrad_d64.h): the whole 9 x 64 x 64 gradient of a workgroup's tiles in registers,
    one slab per workgroup; the bias gradient as one more MFMA tile.  Same shapes as the forward's direct kernel."""
    ops, do, torch = env
    B, H, W = shape
    k, C = 3, 64
    rng = np.random.default_rng(hash(shape) % (1 << 31) + 1)
    x = rng.normal(0, 1, (B, H, W, C)).astype(np.float32)
    dy = rng.normal(0.2, 1, (B, H, W, C)).astype(np.float32)
    wt = torch.zeros((k, k, C, C), dtype=torch.float64, requires_grad=True)
    y = do.conv_same(_bf(torch, x), wt, torch.zeros(C, dtype=torch.float64), 1)
    y.backward(_bf(torch, dy))
    xg, dyg = _t(torch, x).bfloat16(), _t(torch, dy).bfloat16()
    want_b = dyg.double().sum(dim=(0, 1, 2)).cpu()
    with ops.tuned(plan_cus=cus):
        plan = ops.conv_bf16_plan(B, H, W, C, C, k, 1, "wgrad")
        assert plan[:2] == (4, 16) and plan[2] == min(cus, B * ((H + 15) // 16) * ((W + 15) // 16))
        new = ops.conv_wgrad_bf16(xg, dyg, k, k, 1)
        assert _rel(new.cpu(), wt.grad) < 2e-5
        assert torch.equal(new, ops.conv_wgrad_bf16(xg, dyg, k, k, 1)), "filter gradient must be bitwise reproducible"
        db = torch.full((C,), float("nan"), device=xg.device)
        assert torch.equal(ops.conv_wgrad_bf16(xg, dyg, k, k, 1, bias_out=db), new)
        assert _rel(db.cpu().double(), want_b) < 1e-5
        with ops.tuned(wgrad_gen=1):
            assert ops.conv_bf16_plan(B, H, W, C, C, k, 1, "wgrad")[:2] == (128, 64)
            old = ops.conv_wgrad_bf16(xg, dyg, k, k, 1)        # the split-M kernel
        assert _rel(new.cpu(), old.cpu()) < 2e-5
        # f32-stored operands are not the direct kernel's: the same call answers through the split-M kernel
        assert _rel(ops.conv_wgrad_bf16(xg.float(), dyg, k, k, 1).cpu(), wt.grad) < 2e-5


# (B, H, W, Cin, Cout, k, dil), plan_cus: the LDS-DMA filter-gradient kernel (csrc/wgrad_b16t.h) takes bf16-stored tensors with
# Cin % 256 == 0 where the plan uses 256 x 256 filter tiles: M tails inside a 32-pixel step, splits that end inside a step's pairs,
# one to many K steps (all three bodies of the fragment rotation), Cout tails, every tap geometry of the model (1x1, 3x3 rate 1 / 2,
# 4x4 rate 4, 3x3 rate 12 reaching outside the map)
WGRAD_T = [
    ((1, 5, 7, 256, 256, 1, 1), 256),       # 35 pixels: two steps, the second almost empty
    ((2, 13, 11, 256, 264, 3, 1), 256),     # Cout tail (two n tiles, the second 8 wide)
    ((2, 19, 17, 512, 136, 3, 2), 3),       # few "CUs": split-M with several steps per split
    ((1, 16, 16, 256, 320, 4, 4), 256),     # fc6's filter
    ((1, 25, 25, 256, 128, 3, 12), 5),      # rate 12
    ((3, 41, 41, 512, 512, 3, 1), 256),     # conv4_2 at batch 3: the automatic plan's splits
]


@pytest.mark.parametrize("shape,cus", WGRAD_T)
def test_bf16_wgrad_on_lds_dma(env, shape, cus):
    ops, do, torch = env
    B, H, W, Cin, Cout, k, dil = shape
    rng = np.random.default_rng(hash(shape) % (1 << 31))
    x = rng.normal(0, 1, (B, H, W, Cin)).astype(np.float32)
    dy = rng.normal(0, 1, (B, H, W, Cout)).astype(np.float32)
    wt = torch.zeros((k, k, Cin, Cout), dtype=torch.float64, requires_grad=True)
    y = do.conv_same(_bf(torch, x), wt, torch.zeros(Cout, dtype=torch.float64), dil)
    y.backward(_bf(torch, dy))
    xg, dyg = _t(torch, x).bfloat16(), _t(torch, dy).bfloat16()
    with ops.tuned(plan_cus=cus, bf16_tile=256):
        assert ops.conv_bf16_plan(B, H, W, Cin, Cout, k, dil, "wgrad")[:2] == (256, 256)
        new = ops.conv_wgrad_bf16(xg, dyg, k, k, dil)
        assert torch.equal(new, ops.conv_wgrad_bf16(xg, dyg, k, k, dil)), "filter gradient must be bitwise reproducible"
        with ops.tuned(wgrad_gen=1):
            old = ops.conv_wgrad_bf16(xg, dyg, k, k, dil)        # the round-1 kernel on the same plan
    assert _rel(new.cpu(), wt.grad) < 2e-5
    # (measured: the two kernels agree BIT FOR BIT -- the 16x16x32 MFMA sums its 32 k slots as two groups of 16, each exactly, like two
    # 32x32x16 instructions, and the pixel <-> k-slot assignment keeps pixels 0-15 / 16-31 of a step in those groups -- but only the
    # bar is a contract)
    assert _rel(new.cpu(), old.cpu()) < 2e-5


# the bias gradient inside the filter-gradient launch (emadapt_conv_wgrad_bias_bf16): both kernel families, one split and many, column
# tails, the narrow tiles whose loader lanes share columns (64 / 128 wide), fc8's 24 columns
WGRAD_BIAS = WGRAD_T + [
    ((2, 33, 31, 8, 64, 3, 1), 256),        # conv1_1: 128 x 64 tiles
    ((1, 40, 40, 64, 64, 3, 1), 256),       # conv1_2
    ((2, 21, 21, 64, 128, 3, 1), 7),        # 128 x 128 tiles
    ((1, 21, 21, 128, 136, 3, 1), 256),     # a second n tile 8 wide
    ((2, 9, 9, 512, 24, 1, 1), 256),        # fc8
    ((1, 3, 3, 256, 256, 1, 1), 256),       # 9 pixels: one split, one step
]


@pytest.mark.parametrize("shape,cus", WGRAD_BIAS)
def test_bf16_wgrad_carries_the_bias_gradient(env, shape, cus):
    ops, do, torch = env
    B, H, W, Cin, Cout, k, dil = shape
    rng = np.random.default_rng(hash(shape) % (1 << 31))
    xg = _t(torch, rng.normal(0, 1, (B, H, W, Cin)).astype(np.float32)).bfloat16()
    dyg = _t(torch, rng.normal(0.3, 1, (B, H, W, Cout)).astype(np.float32)).bfloat16()
    want = dyg.double().sum(dim=(0, 1, 2)).cpu()
    d = ops.conv_desc(B, H, W, Cin, Cout, k, k, dil)
    from em_adapt_tensorflow_amd import _lib
    for gen in (0, 1):
        with ops.tuned(plan_cus=cus, wgrad_gen=gen):
            assert _lib.load().emadapt_conv_wgrad_bias_bf16_fused(d, 1, 1) == 1
            plain = ops.conv_wgrad_bf16(xg, dyg, k, k, dil)
            db = torch.full((Cout,), float("nan"), device=xg.device)
            dw = ops.conv_wgrad_bf16(xg, dyg, k, k, dil, bias_out=db)
            assert torch.equal(dw, plain), "the filter gradient does not change when the bias gradient rides along"
            assert _rel(db.cpu().double(), want) < 1e-5
            db2 = torch.zeros_like(db)
            ops.conv_wgrad_bf16(xg, dyg, k, k, dil, bias_out=db2)
            assert torch.equal(db, db2), "bias gradient must be bitwise reproducible"
            with ops.tuned(wgrad_bias=1):     # the separate kernel (what f32-stored dy gets)
                assert _lib.load().emadapt_conv_wgrad_bias_bf16_fused(d, 1, 1) == 0
                db3 = torch.zeros_like(db)
                assert torch.equal(ops.conv_wgrad_bf16(xg, dyg, k, k, dil, bias_out=db3), plain)
            assert _rel(db3.cpu().double(), want) < 1e-5
    # f32-stored dy: not fused, same call
    assert _lib.load().emadapt_conv_wgrad_bias_bf16_fused(d, 1, 0) == 0
    db4 = torch.zeros_like(db)
    ops.conv_wgrad_bf16(xg, dyg.float(), k, k, dil, bias_out=db4)
    assert _rel(db4.cpu().double(), want) < 1e-5


def test_bf16_dropout_epilogue_matches_fp32_kernel_mask(env):
    """The counter-based dropout mask depends only on (seed, element index): the bf16 and fp32 kernels drop the same
    elements, and kept values are relu(conv)/keep."""
    ops, do, torch = env
    rng = np.random.default_rng(9)
    x = _t(torch, rng.normal(0, 1, (2, 21, 21, 64)).astype(np.float32))
    w = _t(torch, rng.normal(0, 0.1, (3, 3, 64, 136)).astype(np.float32))
    b = _t(torch, rng.normal(0.5, 0.5, (136,)).astype(np.float32))
    wf, _ = ops.weights_to_bf16(w, want_dgrad=False)
    y0 = ops.conv_fwd_bf16(x, wf, (3, 3), b, 1, relu=True)
    y1 = ops.conv_fwd_bf16(x, wf, (3, 3), b, 1, relu=True, keep_prob=0.5, drop_seed=77)
    y32 = ops.conv_fwd(x, w, b, 1, relu=True, keep_prob=0.5, drop_seed=77)
    kept = y1 != 0
    assert torch.equal(y1[kept], (y0 / 0.5)[kept])
    pos = (y0 > 0) & (ops.conv_fwd(x, w, b, 1, relu=True) > 0)
    assert torch.equal((y1 != 0) & pos, (y32 != 0) & pos)
    frac = kept.sum().item() / max((y0 > 0).sum().item(), 1)
    assert 0.45 < frac < 0.55


@pytest.mark.parametrize("act,bt", [("bf16", "auto"), ("fp32", "auto"), ("bf16", "always")])
def test_bf16_micro_step_tracks_the_bf16_oracle(act, bt):
    """ADAPT(compute_dtype="bf16"): forward, E-step, backward, accumulate and update against the oracle's bf16
    arithmetic model (operands of every conv inner product rounded to bf16, exact accumulation; fp64 elsewhere).
    Two bf16 computations that differ only in summation order agree to ~1e-7 until the first activation lands close
    enough to a bf16 rounding boundary to round the other way; from there they decorrelate to the bf16 noise floor
    (a relative perturbation e becomes ~sqrt(e)/16 after one rounding, fixed point 2^-8): measured 4e-3 on the logits.
    In backward, ReLU gates of near-zero activations flip for that fraction of elements, an L2 error of
    sqrt(fraction): measured <= 4e-2 per layer, while the bf16 model itself sits 4..24 % from the fp32 graph.
    Bars: logits 1e-2 (max), loss 5e-3, gradients 8e-2 (relative L2) per layer and 5e-2 on average.  The kernels
    themselves are held to 2e-5 above.  Label maps must be bit-identical to the oracle E-step on the GPU's logits.
    bt = "always": every layer the LDS-DMA kernel can take (csrc/igemm_b16t.h: conv1_2 .. fc7 forward, conv2_1 .. fc7 data gradient
    at this width) runs on it -- what "auto" does for fc6 / fc7 at the benchmark's batch size."""
    import torch
    from em_adapt_tensorflow_amd.deeplab import ADAPT
    from oracle import deeplab_oracle as do
    from oracle import estep_oracle as eo
    B, width, size, C = 2, 8, 65, 5
    adapt = ADAPT({"input_size": (size, size), "category_num": C, "width": width, "accum_num": 1, "batch_size": B,
                   "compute_dtype": "bf16", "act_dtype": act, "bf16t": bt})
    adapt.build()
    oracle = do.DeepLabOracle(category_num=C, width=width, dtype=torch.float64, compute_dtype="bf16", act_dtype=act)
    oracle32 = do.DeepLabOracle(category_num=C, width=width, dtype=torch.float64)
    params = do.init_params(oracle.specs, seed=11, he=True)
    for layer, (w, b) in params.items():
        adapt.weights[layer][0].copy_(torch.tensor(w).to(adapt.device))
        adapt.weights[layer][1].copy_(torch.tensor(b).to(adapt.device))
    adapt.pre_train(base_lr=1e-3, weight_decay=1e-2, momentum=0.9, batch_size=B)
    images, labels = do.synthetic_batch(B, size, C, seed=5)
    images = (images / 50.0).astype(np.float32)
    S = adapt.out_hw[0]
    rng = np.random.default_rng(1)
    masks = {l: (rng.random((B, S, S, adapt.weights[l][1].shape[0])) < 0.5).astype(np.uint8) for l in ("fc6", "fc7")}
    perms = eo.draw_perms(5, C, seed=77)

    logits = adapt.forward(images, keep_prob=0.5, drop_masks=masks).cpu().numpy()
    assert adapt.net["relu4_2"].dtype == (torch.bfloat16 if act == "bf16" else torch.float32)
    assert adapt.net["output"].dtype == torch.float32
    want = oracle.forward(oracle.tensors(params, requires_grad=False), images, 0.5, masks)["output"].numpy()
    full = oracle32.forward(oracle32.tensors(params, requires_grad=False), images, 0.5, masks)["output"].numpy()
    assert _rel(logits, want) < 1e-2
    assert _rel(logits, full) < 5e-2        # bf16 quantisation against the fp32-semantics graph
    assert _rel(want, full) > 1e-5          # ... and the bf16 model really is a different arithmetic

    # the filter images follow the master weights
    w6 = adapt._segments["fc6"]
    on_t = adapt._bt["fc6"]
    if bt == "always":
        import em_adapt_tensorflow_amd.ops as ops
        assert on_t["fwd"] and on_t["dgrad"] and adapt._bt["conv3_2"]["fwd"] and adapt._bt["conv3_2"]["dgrad"], adapt._bt
        assert torch.equal(w6["wt16"], ops.filter_tiles_bf16(w6["w_pad"])) and torch.equal(w6["wtd16"], ops.filter_tiles_bf16(w6["w_pad"], dgrad=True))
    else:
        assert not any(e["fwd"] or e["dgrad"] for e in adapt._bt.values())   # (too few tiles at this size for "auto")
        assert torch.equal(w6["wd16"][:, :, :w6["pshape"][3]], w6["w_pad"].bfloat16().reshape(w6["wd16"].shape[0], w6["pshape"][2], -1))

    adapt.micro_step(images, labels, keep_prob=0.5, perms=perms, drop_masks=masks)
    torch.cuda.synchronize()
    got_labels = adapt.net["e_argmax"].cpu().numpy()
    want_labels, _ = oracle.estep_labels(logits, labels, perms)
    assert np.array_equal(got_labels, want_labels)

    ref = oracle.micro_step(params, images, labels, perms, weight_decay=1e-2, keep_prob=0.5, masks=masks,
                            labels_override=want_labels)
    assert abs(float(adapt.loss["norm"].cpu()) - ref["loss_norm"]) / abs(ref["loss_norm"]) < 5e-3
    errs = []
    for layer, (dw, db) in ref["grads"].items():
        seg = adapt._segments[layer]
        s, nw = seg["start"], seg["nw"]
        kh, kw, cin, cout = seg["shape"]
        acc_w = adapt._flat["accum"][s:s + nw].view(seg["pshape"])[:, :, :cin, :cout].cpu().numpy()
        acc_b = adapt._flat["accum"][s + nw:s + nw + cout].cpu().numpy()
        errs.append(_l2(acc_w, dw))
        assert errs[-1] < 8e-2, layer
        assert _l2(acc_b, db) < 8e-2, layer
    assert np.mean(errs) < 5e-2

    key = "wt16" if adapt._bt["fc8"]["fwd"] else "wf16"   # (the image fc8's forward pass reads in this mode)
    before = adapt._segments["fc8"][key].clone()
    adapt.apply_update()
    adapt.forward(images, keep_prob=1.0)
    torch.cuda.synchronize()
    seg = adapt._segments["fc8"]
    assert not torch.equal(before, seg[key]), "the bf16 images must be refreshed after the optimizer step"
    kh, kw, cin_p, cout_p = seg["pshape"]
    if key == "wf16":
        assert torch.equal(seg["wf16"][:, :, :cin_p], seg["w_pad"].bfloat16().reshape(1, cin_p, cout_p).permute(0, 2, 1))
    if bt == "always":
        assert torch.equal(w6["wt16"], ops.filter_tiles_bf16(w6["w_pad"])), "the tiled images must be refreshed too"


def test_bf16_training_tracks_fp32_training():
    """40 optimizer steps of the EM-Adapt loop on a fixed synthetic batch, once in fp32 and once in bf16 (same weights,
    data, E-step visiting orders and dropout masks): both must learn, and the bf16 loss must stay close to the fp32 loss.
    Also exercises the refresh of the bf16 filter images after every update and accum_num > 1."""
    import random
    import torch
    from em_adapt_tensorflow_amd.deeplab import ADAPT
    from oracle import deeplab_oracle as do
    B, width, size, C = 4, 16, 65, 5
    images, labels = do.synthetic_batch(B, size, C, seed=3)
    images = (images / 50.0).astype(np.float32)
    losses = {}
    for mode in ("fp32", "bf16"):
        random.seed(5)   # E-step visiting orders are drawn from Python's global generator (estep.py:64-66)
        adapt = ADAPT({"input_size": (size, size), "category_num": C, "width": width, "accum_num": 2, "batch_size": B,
                       "compute_dtype": mode, "seed": 7})
        adapt.build()
        params = do.init_params(do.layer_specs(C, "ref", width), seed=11, he=True)
        for layer, (w, b) in params.items():
            adapt.weights[layer][0].copy_(torch.tensor(w).to(adapt.device))
            adapt.weights[layer][1].copy_(torch.tensor(b).to(adapt.device))
        adapt.pre_train(base_lr=2e-3, weight_decay=1e-5, momentum=0.9, batch_size=B)
        trace = []
        for it in range(80):
            loss = adapt.micro_step(images, labels, keep_prob=0.5)
            if it % 2 == 1:
                adapt.apply_update()
            trace.append(float(loss.cpu()))
        losses[mode] = np.array(trace)
        assert np.all(np.isfinite(trace))
    f, h = losses["fp32"], losses["bf16"]
    assert f[-10:].mean() < 0.7 * f[:4].mean(), "fp32 run did not learn: %s" % f[::8]
    assert h[-10:].mean() < 0.7 * h[:4].mean(), "bf16 run did not learn: %s" % h[::8]
    assert abs(h[0] - f[0]) / f[0] < 2e-2                       # same start (one bf16 forward of difference)
    assert abs(h[-10:].mean() - f[-10:].mean()) / f[-10:].mean() < 0.25, (f[::8], h[::8])


def test_weights_to_bf16_of_several_filters_in_one_launch(env):
    """emadapt_weights_to_bf16_multi: every image bit for bit the single-filter entry's, either image optional per layer, 16 layers."""
    ops, do, torch = env
    rng = np.random.default_rng(5)
    shapes = [(3, 3, 4, 64), (3, 3, 64, 64), (3, 3, 64, 128), (1, 1, 4096, 24), (3, 3, 136, 72), (1, 1, 8, 8), (4, 4, 40, 136), (3, 3, 256, 256)]
    shapes = shapes + shapes
    items, want = [], []
    for k, shp in enumerate(shapes):
        w = _t(torch, rng.normal(0, 0.1, shp).astype(np.float32))
        wf0, wd0 = ops.weights_to_bf16(w)
        wf = torch.full_like(wf0, 3.0) if k % 3 != 1 else None
        wd = torch.full_like(wd0, 3.0) if k % 3 != 2 else None
        items.append((w, wf, wd))
        want.append((wf0, wd0))
    ops.weights_to_bf16_multi(items)
    for (w, wf, wd), (wf0, wd0) in zip(items, want):
        assert wf is None or torch.equal(wf.view(torch.int16), wf0.view(torch.int16))
        assert wd is None or torch.equal(wd.view(torch.int16), wd0.view(torch.int16))
    with pytest.raises(Exception):
        ops.weights_to_bf16_multi(items + items[:1])
    with pytest.raises(Exception):
        ops.weights_to_bf16_multi([(items[0][0], None, None)])
