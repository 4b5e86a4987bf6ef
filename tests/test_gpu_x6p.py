"""GPU parity of the bf16x6 kernels on pre-split operands (csrc/conv_x6p.hip, igemm_x6p.h, wgrad_x6p.h): plane images and the
256x256 LDS-DMA kernels that read them.

Same bar as the native fp32 kernels and the in-kernel-split bf16x6 kernels (tests/test_gpu_kernels.py, tests/test_gpu_x6.py):
2e-5 of the tensor's max magnitude against the fp64 oracle evaluated on the UNROUNDED fp32 inputs; the plane image itself is
checked bit for bit against a NumPy restatement of the split.  Launch plans (plain tiles, stream-K with slab levels, packed rows,
tail launches) are forced onto small shapes through the "plan_cus" / "fp32_loader" knobs.
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


def _bf16_rne(x):
    """fp32 -> bf16 bits (round to nearest even), as uint16; finite inputs."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    return (((u + 0x7FFF + ((u >> 16) & 1)) >> 16) & 0xFFFF).astype(np.uint16)


def _bf16_to_f32(b):
    return (b.astype(np.uint32) << 16).view(np.float32)


def planes_reference(x):
    """[rows][C] fp32 -> [rows][C/16][3][16] bf16 bits: p0 = bf16(x), p1 = bf16(x - p0), p2 = bf16(x - p0 - p1)."""
    rows, C = x.shape
    p0 = _bf16_rne(x)
    r1 = x - _bf16_to_f32(p0)
    p1 = _bf16_rne(r1)
    r2 = r1 - _bf16_to_f32(p1)
    p2 = _bf16_rne(r2)
    out = np.stack([p.reshape(rows, C // 16, 16) for p in (p0, p1, p2)], axis=2)   # [rows][C/16][3][16]
    return out


def test_split_planes_is_exact_and_laid_out_as_documented(env):
    ops, do, torch = env
    rng = np.random.default_rng(0)
    rows, C = 37, 48
    x = (rng.normal(0, 1, (rows, C)) * 10.0 ** rng.integers(-20, 20, (rows, C))).astype(np.float32)
    x[3, 5] = 0.0
    x[4, :] = np.float32(1.0)
    got = ops.split_planes(_t(torch, x)).cpu().numpy().view(np.uint16).reshape(rows, C // 16, 3, 16)
    want = planes_reference(x)
    assert np.array_equal(got, want)
    # the three planes add up to x exactly (8 + 8 + 8 significant bits)
    s = _bf16_to_f32(got[:, :, 0]).astype(np.float64) + _bf16_to_f32(got[:, :, 1]).astype(np.float64) + _bf16_to_f32(got[:, :, 2]).astype(np.float64)
    assert np.array_equal(s.reshape(rows, C).astype(np.float32), x)
    assert np.array_equal(s.reshape(rows, C), x.astype(np.float64))
    with pytest.raises(Exception):
        ops.split_planes(_t(torch, x[:, :40]))   # channel counts must be whole groups of 16


def test_filter_tiles_layout(env):
    """[tap][k / 16][row, zero rows up to a multiple of 256][6 chunks of 16 B]: chunk (2 plane + half) of a row sits at position
    chunk ^ (bit 3 of the row) -- the LDS order of igemm_x6p_kernel's filter tile."""
    ops, do, torch = env
    rng = np.random.default_rng(4)
    k, Cin, Cout = 3, 32, 40
    w = rng.normal(0, 0.1, (k, k, Cin, Cout)).astype(np.float32)
    for dgrad in (False, True):
        rows, kdim = (Cin, Cout) if dgrad else (Cout, Cin)
        if kdim % 16:
            with pytest.raises(Exception):
                ops.filter_tiles(_t(torch, w), dgrad=dgrad)
            continue
        got = ops.filter_tiles(_t(torch, w), dgrad=dgrad).cpu().numpy().view(np.uint16).reshape(k * k, kdim // 16, 256, 6, 8)
        m = w.reshape(k * k, Cin, Cout)
        m = m if dgrad else m.transpose(0, 2, 1)                       # [tap][row][k]
        planes = planes_reference(m.reshape(-1, kdim)).reshape(k * k, rows, kdim // 16, 3, 2, 8)   # [tap][row][g][plane][half][8]
        want = np.zeros((k * k, kdim // 16, 256, 6, 8), np.uint16)
        for row in range(rows):
            par = (row >> 3) & 1
            for p in range(3):
                for h in range(2):
                    want[:, :, row, (2 * p + h) ^ par, :] = planes[:, row, :, p, h, :]
        assert np.array_equal(got, want)


# (B, H, W, Cin, Cout, k, dil): shapes at least one of the plane kernels takes
SHAPES = [
    (2, 13, 11, 16, 256, 3, 1),
    (1, 19, 17, 32, 320, 3, 2),     # a partial second N tile
    (6, 11, 11, 64, 256, 4, 4),     # 4x4 rate 4 on a small map: packed rows (3 M tiles; plans laid out for 3 CUs)
    (3, 9, 9, 256, 256, 1, 1),      # 1x1; filter gradient taken (Cin % 256 == 0)
    (2, 23, 23, 256, 512, 3, 2),    # filter gradient with pixel skipping
    (1, 41, 41, 512, 272, 3, 1),    # 7 M tiles x 2 N tiles
    (2, 12, 12, 256, 48, 3, 1),     # data gradient only (forward N = 48 stays on the in-kernel-split kernels)
    (1, 20, 20, 48, 256, 3, 12),    # rate 12 (LargeFOV): most taps outside the map
]


@pytest.fixture(params=["auto", "cus3", "cus8_dense", "cus16"])
def plan(request):
    """auto: the plans of a 256-CU part; cus3 / cus8: plans laid out for 3 / 8 CUs, which turns these small shapes into several
    rounds of plain tiles + a stream-K launch over the rows of the partial last round (round 4: whole rounds always run as plain
    tiles), packed rows stream-K; cus16: fewer tiles than CUs -- stream-K over everything; dense: no packed rows, no pixel skipping."""
    import em_adapt_tensorflow_amd.ops as ops
    if request.param == "cus3":
        ops.tune_set("plan_cus", 3)
    elif request.param == "cus16":     # more CUs than most of these shapes have tiles: stream-K over everything, every tile shared
        ops.tune_set("plan_cus", 16)
    elif request.param == "cus8_dense":
        ops.tune_set("plan_cus", 8)
        ops.tune_set("fp32_loader", 2)
    yield request.param
    ops.tune_set("plan_cus", 0)
    ops.tune_set("fp32_loader", 0)


@pytest.mark.parametrize("shape", SHAPES)
def test_plane_kernels_against_the_fp64_oracle(env, shape, plan):
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

    d = ops.conv_desc(B, H, W, Cin, Cout, k, k, dil)
    xg, wg, bg, dyg = _t(torch, x), _t(torch, w), _t(torch, b), _t(torch, dy)
    taken = [ops.conv_x6p_supported(d, kind) for kind in (0, 1, 2)]
    assert any(taken)
    if taken[0]:
        xp, wkp = ops.split_planes(xg), ops.filter_tiles(wg)
        assert _rel(ops.conv_fwd_x6p(xp, wkp, d, None).cpu(), (y - bt).detach()) < 2e-5
        assert _rel(ops.conv_fwd_x6p(xp, wkp, d, bg, relu=False).cpu(), y.detach()) < 2e-5
        got = ops.conv_fwd_x6p(xp, wkp, d, bg, relu=True)
        assert _rel(got.cpu(), torch.relu(y).detach()) < 2e-5
        assert torch.equal(got, ops.conv_fwd_x6p(xp, wkp, d, bg, relu=True))   # slab levels are added in a fixed order
        mask = (rng.random((B, H, W, Cout)) < 0.5).astype(np.uint8)
        want = np.where(mask != 0, torch.relu(y).detach().numpy() * 2.0, 0.0)
        assert _rel(ops.conv_fwd_x6p(xp, wkp, d, bg, relu=True, keep_prob=0.5, drop_mask=torch.tensor(mask, device="cuda:0")).cpu(), want) < 2e-5
    else:
        with pytest.raises(Exception):
            ops.conv_fwd_x6p(ops.split_planes(xg), ops.filter_tiles(wg), d, None)
    if taken[1]:
        dyp, wp = ops.split_planes(dyg), ops.filter_tiles(wg, dgrad=True)
        got = ops.conv_dgrad_x6p(dyp, wp, d)
        assert _rel(got.cpu(), xt.grad) < 2e-5
        assert torch.equal(got, ops.conv_dgrad_x6p(dyp, wp, d))
        gate = rng.normal(0, 1, (B, H, W, Cin)).astype(np.float32)
        want = np.where(gate > 0, xt.grad.numpy() * 2.0, 0.0)
        assert _rel(ops.conv_dgrad_x6p(dyp, wp, d, gate=_t(torch, gate), gate_scale=2.0).cpu(), want) < 2e-5
    if taken[2]:
        xp, dyp = ops.split_planes(xg), ops.split_planes(dyg)
        got = ops.conv_wgrad_x6p(xp, dyp, d)
        assert _rel(got.cpu(), wt.grad) < 2e-5
        assert torch.equal(got, ops.conv_wgrad_x6p(xp, dyp, d))   # split-M slabs are summed in a fixed order


def test_plans_of_the_forced_layouts_differ(env):
    """The knobs above really change the launch plan (plain tiles vs stream-K with slab levels vs packed rows vs a tail)."""
    ops, do, torch = env
    d = ops.conv_desc(1, 41, 41, 512, 272, 3, 3, 1)
    auto = ops.conv_x6p_plan(d)
    assert auto[1] == 0 and auto[0] == 1681                      # 14 tiles on 256 CUs: plain tiles
    ops.tune_set("plan_cus", 16)
    try:
        sk = ops.conv_x6p_plan(d)
        assert sk[1] == 1 and sk[2] >= 2                         # 14 tiles on 16 CUs: stream-K, tiles shared by several workgroups
        ops.tune_set("plan_cus", 3)
        t3 = ops.conv_x6p_plan(d)
        assert t3[1] == 0 and t3[0] == 6 * 256 and t3[4] >= 2    # on 3 CUs: 4 whole rounds of plain tiles (6 M tiles x 2) + a stream-K tail
        d4 = ops.conv_desc(6, 11, 11, 64, 256, 4, 4, 4)
        assert ops.conv_x6p_plan(d4)[3] > 0                      # 4x4 rate 4 on an 11x11 map: packed rows
        ops.tune_set("fp32_loader", 2)
        assert ops.conv_x6p_plan(d4)[3] == 0
        ops.tune_set("fp32_loader", 0)
        dn = ops.conv_desc(2, 41, 41, 64, 2048, 1, 1, 1)            # N > 1024: whole rounds as plain tiles + a stream-K tail
        tail = ops.conv_x6p_plan(dn)
        assert tail[1] == 0 and 0 < tail[0] < 2 * 41 * 41
    finally:
        ops.tune_set("plan_cus", 0)
        ops.tune_set("fp32_loader", 0)


def test_whole_rounds_plus_tail_launch(env):
    """N > 1024: rows of whole rounds as plain tiles, the remaining rows as a second launch (forced onto a small shape)."""
    ops, do, torch = env
    rng = np.random.default_rng(9)
    B, H, W, Cin, Cout = 2, 41, 41, 512, 2048
    x = rng.normal(0, 1, (B, H, W, Cin)).astype(np.float32)
    w = rng.normal(0, 0.2, (1, 1, Cin, Cout)).astype(np.float32)
    want = x.reshape(-1, Cin).astype(np.float64) @ w.reshape(Cin, Cout).astype(np.float64)
    d = ops.conv_desc(B, H, W, Cin, Cout, 1, 1, 1)
    xp, wkp = ops.split_planes(_t(torch, x)), ops.filter_tiles(_t(torch, w))
    for cus in (5, 12, 0):
        ops.tune_set("plan_cus", cus)
        try:
            if cus == 5:
                assert ops.conv_x6p_plan(d)[4] > 0   # the tail is a stream-K launch with slab levels
            got = ops.conv_fwd_x6p(xp, wkp, d, None)
        finally:
            ops.tune_set("plan_cus", 0)
        assert _rel(got.cpu().numpy().reshape(-1, Cout), want) < 2e-5, cus


def test_x6p_is_as_close_to_fp64_as_the_native_fp32_kernels(env):
    """Error against fp64 on a long reduction (K = 4608), all three GEMMs, every kernel running the reduction UNSPLIT (as at the
    benchmark shapes; on a shape this small the native plans would split K nine ways and add the partial sums pairwise, which
    hides most of a sequential fp32 accumulation's error): the plane kernels round once per MFMA, 6 * K / 16 times per output, the
    native kernel K / 2 times -- the six-MFMA sum is no further from fp64 than the native fp32 MFMA chain."""
    ops, do, torch = env
    rng = np.random.default_rng(3)
    B, H, W, Cin, Cout, k = 2, 21, 21, 512, 256, 3
    x = rng.normal(0, 1, (B, H, W, Cin)).astype(np.float32)
    w = rng.normal(0, 0.05, (k, k, Cin, Cout)).astype(np.float32)
    dy = rng.normal(0, 1, (B, H, W, Cout)).astype(np.float32)
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    wt = torch.tensor(w, dtype=torch.float64, requires_grad=True)
    y = do.conv_same(xt, wt, None, 1)
    y.backward(torch.tensor(dy, dtype=torch.float64))
    xg, wg, dyg = _t(torch, x), _t(torch, w), _t(torch, dy)
    d = ops.conv_desc(B, H, W, Cin, Cout, k, k, 1)
    xp, dyp = ops.split_planes(xg), ops.split_planes(dyg)
    ops.tune_set("splitk", 1)
    try:
        native_fwd, native_dgrad = ops.conv_fwd(xg, wg, None, 1), ops.conv_dgrad(dyg, wg, 1)
    finally:
        ops.tune_set("splitk", 0)
    assert ops.conv_x6p_plan(d)[1:3] == (0, 0) and ops.conv_x6p_plan(d, True)[1:3] == (0, 0)   # plain tiles: one accumulator per output
    pairs = [
        ("fwd", native_fwd, ops.conv_fwd_x6p(xp, ops.filter_tiles(wg), d, None), y.detach().numpy()),
        ("dgrad", native_dgrad, ops.conv_dgrad_x6p(dyp, ops.filter_tiles(wg, dgrad=True), d), xt.grad.numpy()),
        ("wgrad", ops.conv_wgrad(xg, dyg, k, k, 1), ops.conv_wgrad_x6p(xp, dyp, d), wt.grad.numpy()),
    ]
    for name, native, planes, want in pairs:
        e_native, e_planes = _rel(native.cpu(), want), _rel(planes.cpu(), want)
        print("%s: max error / max vs fp64: native fp32 %.2e, planes %.2e" % (name, e_native, e_planes))
        assert e_native < 5e-6 and e_planes < 3e-6, name
        assert e_planes < 1.5 * e_native + 2e-7, name


@pytest.mark.parametrize("layer", ["fc7", "fc6", "conv5_2", "conv3_2"])
def test_plane_kernels_at_the_benchmark_shapes(env, layer):
    """Batch-16 shapes of BASELINE configs[1] against the native fp32 kernels (themselves held to the CPU oracle at these
    shapes in tests/test_gpu_fullshape.py): two fp32-grade sums of up to 65536 terms in different orders."""
    ops, do, torch = env
    H, Cin, Cout, k, rate = {"fc7": (41, 4096, 4096, 1, 1), "fc6": (41, 512, 4096, 4, 4), "conv5_2": (41, 512, 512, 3, 2),
                             "conv3_2": (81, 256, 256, 3, 1)}[layer]
    B = 16
    g = torch.Generator(device="cuda:0").manual_seed(7)
    x = torch.randn((B, H, H, Cin), device="cuda:0", generator=g).relu_()
    w = torch.randn((k, k, Cin, Cout), device="cuda:0", generator=g) * 0.02
    b = torch.randn((Cout,), device="cuda:0", generator=g)
    dy = torch.randn((B, H, H, Cout), device="cuda:0", generator=g)
    d = ops.conv_desc(B, H, H, Cin, Cout, k, k, rate)
    assert all(ops.conv_x6p_supported(d, kind) for kind in (0, 1, 2))
    xp, dyp = ops.split_planes(x), ops.split_planes(dy)

    def close(got, want, bar):
        return float((got - want).abs().max() / want.abs().max()) < bar

    y = ops.conv_fwd_x6p(xp, ops.filter_tiles(w), d, b, relu=True)
    assert close(y, ops.conv_fwd(x, w, b, rate, relu=True), 2e-5)
    dx = ops.conv_dgrad_x6p(dyp, ops.filter_tiles(w, dgrad=True), d, gate=x, gate_scale=2.0)
    assert close(dx, ops.conv_dgrad(dy, w, rate, gate=x, gate_scale=2.0), 3e-5)
    assert torch.equal(dx, ops.conv_dgrad_x6p(dyp, ops.filter_tiles(w, dgrad=True), d, gate=x, gate_scale=2.0))
    dw = ops.conv_wgrad_x6p(xp, dyp, d)
    assert close(dw, ops.conv_wgrad(x, dy, k, k, rate), 2e-5)
    del xp, dyp, y, dx, dw
    torch.cuda.empty_cache()


def test_x6p_micro_step_matches_the_fp32_oracle_at_the_fp32_bars():
    """ADAPT(compute_dtype="bf16x6") at a width where conv4 .. fc7 run on the plane kernels, against the fp64 oracle of the fp32
    graph at the native path's bars (tests/test_gpu_model.py): logits 1e-4, loss 1e-4, gradients 2e-4, label maps identical.
    (Weights from seed 12: with some seeds a max-pool window holds two activations that differ by less than an fp32 rounding, or a
    pre-activation lies that close to zero, and ANY fp32 evaluation may route that gradient differently from the fp64 oracle --
    a discrete 1e-3 event in the layers below that says nothing about the kernels; measured gradient errors otherwise: 1e-6 at
    fc8 growing to 2e-5 at conv1_1.)"""
    import torch
    from em_adapt_tensorflow_amd.deeplab import ADAPT
    from oracle import deeplab_oracle as do
    from oracle import estep_oracle as eo
    B, width, size, C = 2, 32, 65, 5
    adapt = ADAPT({"input_size": (size, size), "category_num": C, "width": width, "accum_num": 1, "batch_size": B,
                   "compute_dtype": "bf16x6", "x6_planes": "always"})
    adapt.build()
    oracle = do.DeepLabOracle(category_num=C, width=width, dtype=torch.float64)
    params = do.init_params(oracle.specs, seed=12, he=True)
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
    engaged = adapt._bufs["xp"]
    assert engaged["fc7"]["fwd"] and engaged["fc7"]["dgrad"] and engaged["fc7"]["wgrad"]
    assert engaged["fc6"]["fwd"] and engaged["conv5_2"]["fwd"] and engaged["conv5_2"]["wgrad"]
    assert not engaged["conv1_1"]["fwd"] and not engaged["fc8"]["wgrad"]
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
    # the filter plane images follow the optimizer
    before = adapt._segments["fc7"]["wkp"].clone()
    adapt.apply_update()
    adapt.forward(images, keep_prob=1.0)
    torch.cuda.synchronize()
    seg = adapt._segments["fc7"]
    assert not torch.equal(before, seg["wkp"])
    import em_adapt_tensorflow_amd.ops as ops
    assert torch.equal(seg["wkp"], ops.filter_tiles(seg["w_pad"].contiguous()))
    assert torch.equal(seg["wp6"], ops.filter_tiles(seg["w_pad"].contiguous(), dgrad=True))


def test_full_size_single_image_bf16x6():
    """BASELINE config 1 (one 321x321 image, 21 classes, the network as written) with compute_dtype bf16x6: conv3_1 .. fc7 run on
    the plane kernels at their real channel counts (K up to 65536), logits within north_star's 1e-3 of the fp32 CPU oracle, label
    map identical (mirrors tests/test_gpu_model.py::test_full_size_single_image_config1)."""
    import torch
    from em_adapt_tensorflow_amd.deeplab import ADAPT
    from oracle import deeplab_oracle as do
    from oracle import estep_oracle as eo
    adapt = ADAPT({"input_size": (321, 321), "category_num": 21, "width": 64, "preset": "ref", "accum_num": 1, "batch_size": 1,
                   "compute_dtype": "bf16x6", "x6_planes": "always"})
    adapt.build()
    oracle = do.DeepLabOracle(category_num=21, preset="ref", width=64, dtype=torch.float32)
    params = do.init_params(oracle.specs, seed=11, he=True)
    for layer, (w, b) in params.items():
        adapt.weights[layer][0].copy_(torch.tensor(w).to(adapt.device))
        adapt.weights[layer][1].copy_(torch.tensor(b).to(adapt.device))
    images, labels = do.synthetic_batch(1, 321, 21, seed=0)
    images = (images / 50.0).astype(np.float32)
    logits = adapt.forward(images, 1.0).cpu().numpy()
    engaged = adapt._bufs["xp"]
    for layer in ("conv3_2", "conv4_2", "conv5_3", "fc6", "fc7"):
        assert engaged[layer]["fwd"] and engaged[layer]["dgrad"] and engaged[layer]["wgrad"], layer
    assert engaged["conv3_1"]["fwd"] and not engaged["conv3_1"]["wgrad"] and not engaged["conv2_2"]["fwd"]
    want = oracle.forward(oracle.tensors(params, requires_grad=False), images)["output"].numpy()
    assert logits.shape == (1, 41, 41, 21)
    assert _rel(logits, want) < 1e-3
    perms = eo.draw_perms(5, 21, seed=3)
    adapt.pre_train(1e-3, 1e-5, 0.9, 1)
    adapt.micro_step(images, labels, keep_prob=1.0, perms=perms)
    adapt.apply_update()
    torch.cuda.synchronize()
    want_labels, _ = oracle.estep_labels(logits, labels, perms)
    assert np.array_equal(adapt.net["e_argmax"].cpu().numpy(), want_labels)
    assert np.isfinite(float(adapt.loss["norm"].cpu()))


def test_full_size_config5_bf16x6():
    """BASELINE config 5 (513x513, 41 classes, fc6 3x3 rate 12 -> 1024) with compute_dtype bf16x6: the rate-12 gather over packed
    rows on 65x65 maps, logits within 1e-3 of the fp32 CPU oracle, label map identical (mirrors test_full_size_config5)."""
    import torch
    from em_adapt_tensorflow_amd.deeplab import ADAPT
    from oracle import deeplab_oracle as do
    from oracle import estep_oracle as eo
    C, size = 41, 513
    adapt = ADAPT({"input_size": (size, size), "category_num": C, "width": 64, "preset": "largefov", "accum_num": 1, "batch_size": 1,
                   "compute_dtype": "bf16x6", "x6_planes": "always"})
    adapt.build()
    oracle = do.DeepLabOracle(category_num=C, preset="largefov", width=64, dtype=torch.float32)
    params = do.init_params(oracle.specs, seed=11, he=True)
    for layer, (w, b) in params.items():
        adapt.weights[layer][0].copy_(torch.tensor(w).to(adapt.device))
        adapt.weights[layer][1].copy_(torch.tensor(b).to(adapt.device))
    images, labels = do.synthetic_batch(1, size, C, seed=4)
    images = (images / 50.0).astype(np.float32)
    logits = adapt.forward(images, 1.0).cpu().numpy()
    engaged = adapt._bufs["xp"]
    assert engaged["fc6"]["fwd"] and engaged["fc6"]["dgrad"] and engaged["fc6"]["wgrad"] and engaged["fc7"]["fwd"]
    want = oracle.forward(oracle.tensors(params, requires_grad=False), images)["output"].numpy()
    assert logits.shape == (1, 65, 65, C)
    assert _rel(logits, want) < 1e-3
    perms = eo.draw_perms(5, C, seed=6)
    adapt.pre_train(1e-3, 1e-5, 0.9, 1)
    adapt.micro_step(images, labels, keep_prob=1.0, perms=perms)
    adapt.apply_update()
    torch.cuda.synchronize()
    want_labels, _ = oracle.estep_labels(logits, labels, perms)
    got = adapt.net["e_argmax"].cpu().numpy()
    assert np.array_equal(got, want_labels)
    assert len(np.unique(got)) >= 2 and np.isfinite(float(adapt.loss["norm"].cpu()))


def test_differential_fuzz_against_the_native_kernels():
    """80 random shapes / filters / dilations / launch layouts / epilogues (scripts/x6p_fuzz.py, fixed seed): the plane kernels
    against the native fp32 kernels at 3e-5 of the tensor's maximum."""
    import importlib.util, os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "x6p_fuzz.py")
    spec = importlib.util.spec_from_file_location("x6p_fuzz", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.main(80, 7) == 0


def test_filter_plane_images_exist_only_for_the_layers_that_use_them():
    """The tiled filter images (2 x 6 bytes per weight) are allocated and refreshed for the layers whose forward / data gradient
    really runs on the plane kernels at the batch size in use, not for every layer the channel rule would allow (advisor finding,
    round 3: ~600 MB rebuilt per optimizer step at batch 1, where no plane kernel runs)."""
    import torch
    from em_adapt_tensorflow_amd.deeplab import ADAPT
    from oracle import deeplab_oracle as do
    images, labels = do.synthetic_batch(1, 65, 5, seed=1)
    for mode, expect_any in (("auto", False), ("always", True)):
        a = ADAPT({"input_size": (65, 65), "category_num": 5, "width": 32, "batch_size": 1, "accum_num": 1, "compute_dtype": "bf16x6",
                   "x6_planes": mode})
        a.build()
        a.pre_train(1e-3, 1e-5, 0.9, 1)
        a.micro_step((images / 50).astype(np.float32), labels, keep_prob=1.0)
        a.apply_update()
        a.forward((images / 50).astype(np.float32), 1.0)          # inference pass: refreshes the images in use, splits no wgrad-only input
        torch.cuda.synchronize()
        xp = a._bufs["xp"]
        for name, seg in a._segments.items():
            assert (seg["wkp"] is not None) == xp[name]["fwd"], (mode, name)
            assert (seg["wp6"] is not None) == xp[name]["dgrad"], (mode, name)
        assert any(e["fwd"] or e["dgrad"] for e in xp.values()) == expect_any, mode
        assert np.isfinite(float(a.loss["norm"].cpu()))
