"""GPU parity of the bf16x6 convolution kernels (fp32-grade GEMMs on the bf16 matrix pipe: exact 3-way bf16 split of
both operands, six products per term, fp32 accumulation -- csrc/conv_x6.hip).

The bar is the one the native fp32 kernels are held to (tests/test_gpu_kernels.py): 2e-5 of the tensor's max magnitude
against the fp64 oracle evaluated on the UNROUNDED fp32 inputs.  A plain bf16 kernel misses that bar by two orders of
magnitude (checked below), so passing it shows the split really carries the full fp32 mantissa.
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


def _t(torch, a):
    return torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device="cuda:0")


SHAPES = [
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
    (2, 49, 47, 256, 128, 3, 1),
    (1, 41, 41, 512, 136, 4, 4),
]


@pytest.fixture(params=["auto", "128", "256"])
def tile(request):
    """Run the kernels in their automatic plan and with each tile configuration forced."""
    import em_adapt_tensorflow_amd.ops as ops
    ops.tune_set("x6_tile", 0 if request.param == "auto" else int(request.param))
    yield request.param
    ops.tune_set("x6_tile", 0)


@pytest.mark.parametrize("shape", SHAPES)
def test_x6_conv_fwd_dgrad_wgrad(env, shape, tile):
    ops, do, torch = env
    B, H, W, Cin, Cout, k, dil = shape
    rng = np.random.default_rng(hash(shape) % (1 << 31))
    x = rng.normal(0, 1, (B, H, W, Cin)).astype(np.float32)
    w = rng.normal(0, 0.2, (k, k, Cin, Cout)).astype(np.float32)
    b = rng.normal(0, 0.5, (Cout,)).astype(np.float32)
    dy = rng.normal(0, 1, (B, H, W, Cout)).astype(np.float32)
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    wt = torch.tensor(w, dtype=torch.float64, requires_grad=True)
    bt = torch.tensor(b, dtype=torch.float64)
    y = do.conv_same(xt, wt, bt, dil)
    y.backward(torch.tensor(dy, dtype=torch.float64))

    xg, wg, bg, dyg = _t(torch, x), _t(torch, w), _t(torch, b), _t(torch, dy)
    wk = ops.filter_kmajor(wg)
    assert torch.equal(wk, wg.reshape(k * k, Cin, Cout).permute(0, 2, 1).contiguous())
    assert _rel(ops.conv_fwd_x6(xg, wk, (k, k), None, dil).cpu(), (y - bt).detach()) < 2e-5
    assert _rel(ops.conv_fwd_x6(xg, wk, (k, k), bg, dil, relu=False).cpu(), y.detach()) < 2e-5
    assert _rel(ops.conv_fwd_x6(xg, wk, (k, k), bg, dil, relu=True).cpu(), torch.relu(y).detach()) < 2e-5
    assert _rel(ops.conv_dgrad_x6(dyg, wg, dil).cpu(), xt.grad) < 2e-5
    gate = rng.normal(0, 1, (B, H, W, Cin)).astype(np.float32)
    want = np.where(gate > 0, xt.grad.numpy() * 2.0, 0.0)
    assert _rel(ops.conv_dgrad_x6(dyg, wg, dil, gate=_t(torch, gate), gate_scale=2.0).cpu(), want) < 2e-5
    assert _rel(ops.conv_wgrad_x6(xg, dyg, k, k, dil).cpu(), wt.grad) < 2e-5


def test_x6_is_as_close_to_fp64_as_the_native_fp32_kernels(env):
    """Error against fp64 on a long reduction (K = 4608): bf16x6 within 2x of the native fp32 MFMA kernels, plain bf16
    two orders of magnitude away."""
    ops, do, torch = env
    rng = np.random.default_rng(3)
    B, H, W, Cin, Cout, k = 2, 21, 21, 512, 128, 3
    x = rng.normal(0, 1, (B, H, W, Cin)).astype(np.float32)
    w = rng.normal(0, 0.05, (k, k, Cin, Cout)).astype(np.float32)
    y64 = do.conv_same(torch.tensor(x, dtype=torch.float64), torch.tensor(w, dtype=torch.float64), None, 1).numpy()
    xg, wg = _t(torch, x), _t(torch, w)
    e_native = _rel(ops.conv_fwd(xg, wg, None, 1).cpu(), y64)
    e_x6 = _rel(ops.conv_fwd_x6(xg, ops.filter_kmajor(wg), (k, k), None, 1).cpu(), y64)
    wf, _ = ops.weights_to_bf16(wg, want_dgrad=False)
    e_bf16 = _rel(ops.conv_fwd_bf16(xg, wf, (k, k), None, 1).cpu(), y64)
    print("max error / max |y| vs fp64: native fp32 %.2e, bf16x6 %.2e, bf16 %.2e" % (e_native, e_x6, e_bf16))
    assert e_native < 2e-6 and e_x6 < 2e-6
    assert e_x6 < 2 * e_native + 2e-7
    assert e_bf16 > 50 * e_x6


def test_x6_wgrad_split_m_is_deterministic(env):
    ops, do, torch = env
    rng = np.random.default_rng(5)
    xg = _t(torch, rng.normal(0, 1, (2, 97, 95, 8)).astype(np.float32))
    dyg = _t(torch, rng.normal(0, 1, (2, 97, 95, 64)).astype(np.float32))
    a = ops.conv_wgrad_x6(xg, dyg, 3, 3, 1)
    b = ops.conv_wgrad_x6(xg, dyg, 3, 3, 1)
    assert torch.equal(a, b)
    assert _rel(a.cpu(), ops.conv_wgrad(xg, dyg, 3, 3, 1).cpu()) < 2e-5


def test_x6_micro_step_matches_the_fp32_oracle_at_the_fp32_bars():
    """ADAPT(compute_dtype="bf16x6") against the fp64 oracle of the fp32 graph -- the SAME bars as the native fp32 path
    (tests/test_gpu_model.py): logits 1e-4, loss 1e-4, gradients 2e-4, label maps identical."""
    import torch
    from em_adapt_tensorflow_amd.deeplab import ADAPT
    from oracle import deeplab_oracle as do
    from oracle import estep_oracle as eo
    B, width, size, C = 2, 8, 65, 5
    adapt = ADAPT({"input_size": (size, size), "category_num": C, "width": width, "accum_num": 1, "batch_size": B,
                   "compute_dtype": "bf16x6", "x6_planes": "always"})   # (at this size "auto" would hand every GEMM to the native kernels)
    adapt.build()
    oracle = do.DeepLabOracle(category_num=C, width=width, dtype=torch.float64)
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
    acts = oracle.forward(oracle.tensors(params, requires_grad=False), images, 0.5, masks)
    assert _rel(logits, acts["output"].numpy()) < 1e-4
    adapt.micro_step(images, labels, keep_prob=0.5, perms=perms, drop_masks=masks)
    torch.cuda.synchronize()
    want_labels, _ = oracle.estep_labels(logits, labels, perms)
    assert np.array_equal(adapt.net["e_argmax"].cpu().numpy(), want_labels)
    ref = oracle.micro_step(params, images, labels, perms, weight_decay=1e-2, keep_prob=0.5, masks=masks,
                            labels_override=want_labels)
    assert abs(float(adapt.loss["norm"].cpu()) - ref["loss_norm"]) / abs(ref["loss_norm"]) < 1e-4
    for layer, (dw, db) in ref["grads"].items():
        seg = adapt._segments[layer]
        s, nw = seg["start"], seg["nw"]
        kh, kw, cin, cout = seg["shape"]
        acc_w = adapt._flat["accum"][s:s + nw].view(seg["pshape"])[:, :, :cin, :cout].cpu().numpy()
        acc_b = adapt._flat["accum"][s + nw:s + nw + cout].cpu().numpy()
        assert _rel(acc_w, dw) < 2e-4, layer
        assert _rel(acc_b, db) < 2e-4, layer
    # the k-major filter image follows the optimizer
    before = adapt._segments["fc8"]["wk32"].clone()
    adapt.apply_update()
    adapt.forward(images, keep_prob=1.0)
    torch.cuda.synchronize()
    seg = adapt._segments["fc8"]
    assert not torch.equal(before, seg["wk32"])
    kh, kw, cin_p, cout_p = seg["pshape"]
    assert torch.equal(seg["wk32"], seg["w_pad"].reshape(1, cin_p, cout_p).permute(0, 2, 1).contiguous())
