"""GPU parity of the fp32 conv kernels AT THE BENCHMARK SHAPES (BASELINE configs[1]: batch 16, 41x41 maps) and of every
launch-plan variant on small awkward shapes.

Why a file of its own: the launch plan depends on the shape (csrc/conv_igemm.hip plan_launches).  At batch 16 the wide
layers -- fc6 / fc7 forward, fc7 / fc8 data gradient: deeplab.py:86-96, shapes deeplab.py:143-149, a quarter of the step
-- run `igemm_kernel<256,256,32,128>` on the rows that fill whole rounds of the 256 CUs plus a 128x128 split-K launch on
the remaining rows, and conv3..conv5 run split-K by 3; none of the small shapes in test_gpu_kernels.py selects those
plans by itself.  Here:

  * the true shapes run against the fp32 PyTorch-CPU oracle (oracle/deeplab_oracle.conv_same and its autograd; each
    layer is a few seconds on the GPU box's host cores) with a bar of 1e-4 of the tensor's max magnitude, with the
    epilogues the model uses (bias + ReLU + dropout mask forward; ReLU-gradient gate with the dropout scale backward),
    a bitwise repeatability check, and an assertion (emadapt_conv_plan) that the plan under test IS the big-tile one;
  * every tile variant (round-1 kernels 1-3, second-generation LDS-DMA kernel 4) is forced (emadapt_tune_set) onto small shapes with M / N / K tails, against the fp64 oracle at
    2e-5, with `plan_cus` shrunk so that the row split (m_begin / m_end), the partial last tile and the slab reuse of
    the tail launch are hit at awkward sizes.
"""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch
    assert torch.cuda.is_available()
    import em_adapt_tensorflow_amd.ops as ops
    from oracle import deeplab_oracle as do
    torch.set_num_threads(max(1, min(64, len(__import__("os").sched_getaffinity(0)))))
    return ops, do, torch


def _rel(got, want):
    want = np.asarray(want, dtype=np.float64)
    got = np.asarray(got, dtype=np.float64)
    return np.abs(got - want).max() / max(np.abs(want).max(), 1e-30)


def _t(torch, a, dtype=None):
    return torch.tensor(np.ascontiguousarray(a), dtype=dtype or torch.float32, device="cuda:0")


# name, (B, H, W, Cin, Cout, k, dil), expected forward plan variant, expected dgrad plan variant
FULL_SHAPES = [
    ("fc6", (16, 41, 41, 512, 4096, 4, 4), 5, 5),      # deeplab.py:92,145: 44.8 % of the MACs; stream-K over packed rows (the
                                                       # filter rows -6 / +6 lie outside the image for 6 of its 41 rows each)
    ("fc7", (16, 41, 41, 4096, 4096, 1, 1), 4, 4),     # deeplab.py:95,147
    ("fc8", (16, 41, 41, 4096, 24, 1, 1), 1, 1),       # deeplab.py:95,149 (21 classes padded to 24: dgrad K = 24, one K step)
    ("conv4_2", (16, 41, 41, 512, 512, 3, 1), 5, 5),   # 212 tiles of 256x256 on 256 CUs: stream-K pieces of 238.5 K steps
    ("conv5_2", (16, 41, 41, 512, 512, 3, 2), 5, 5),   # atrous rate 2 (deeplab.py:65): packed rows over 3 x 3 pixel classes
    ("conv3_2", (16, 81, 81, 256, 256, 3, 1), 5, 5),   # 411 tiles: 1.6 rounds -> stream-K
    ("conv3_1", (16, 81, 81, 128, 256, 3, 1), 5, 3),   # dgrad has N = 128
]


@pytest.mark.parametrize("name,shape,fwd_variant,dgrad_variant", FULL_SHAPES, ids=[s[0] for s in FULL_SHAPES])
def test_benchmark_shape_against_fp32_oracle(env, name, shape, fwd_variant, dgrad_variant):
    ops, do, torch = env
    B, H, W, Cin, Cout, k, dil = shape
    pf = ops.conv_plan(B, H, W, Cin, Cout, k, dil, dgrad=False)
    pd = ops.conv_plan(B, H, W, Cin, Cout, k, dil, dgrad=True)
    assert pf[0] == fwd_variant and pd[0] == dgrad_variant, (pf, pd)
    if fwd_variant in (2, 4):  # rows of whole rounds on 256x256 tiles + a split launch of smaller tiles over the rest
        assert 0 < pf[1] < B * H * W and pf[1] % 256 == 0 and pf[3] >= 1, pf
    if fwd_variant == 5:
        # one launch over all rows, >= 2 slab levels; dilated filters: over their pixel classes (fc6, 4x4 rate 4: 5 x 5)
        assert pf[1] == B * H * W and pf[2] >= 2 and pf[3] == {"fc6": 25, "conv5_2": 9}.get(name, 0), pf
    if name == "fc6":
        assert pd[1] == B * H * W and pd[2] >= 2, pd  # 212 big tiles on 256 CUs: pieces of the 65536-long reductions

    rng = np.random.default_rng(__import__("zlib").crc32(name.encode()))
    fan_in = k * k * Cin
    x = np.maximum(rng.normal(0, 1, (B, H, W, Cin)), 0).astype(np.float32)     # post-ReLU-like input (the dgrad gate)
    w = rng.normal(0, np.sqrt(2.0 / fan_in), (k, k, Cin, Cout)).astype(np.float32)
    b = rng.normal(0, 0.1, (Cout,)).astype(np.float32)
    dy = rng.normal(0, 1, (B, H, W, Cout)).astype(np.float32)
    mask = (rng.random((B, H, W, Cout)) < 0.5).astype(np.uint8)
    relu = name != "fc8"
    drop = name in ("fc6", "fc7")

    xt = torch.tensor(x, requires_grad=True)
    wt = torch.tensor(w, requires_grad=True)
    bt = torch.tensor(b)
    y_lin = do.conv_same(xt, wt, bt, dil)
    y_lin.backward(torch.tensor(dy))
    want_y = y_lin.detach().numpy()
    if relu:
        want_y = np.maximum(want_y, 0)
    if drop:
        want_y = want_y / np.float32(0.5) * mask
    want_dx = np.where(x > 0, xt.grad.numpy() * np.float32(2.0 if drop else 1.0), 0)

    xg, wg, bg, dyg = _t(torch, x), _t(torch, w), _t(torch, b), _t(torch, dy)
    mg = _t(torch, mask, torch.uint8) if drop else None
    kw = dict(relu=relu, keep_prob=0.5 if drop else 1.0, drop_mask=mg)
    y1 = ops.conv_fwd(xg, wg, bg, dil, **kw).cpu().numpy()
    assert _rel(y1, want_y) < 1e-4
    assert np.array_equal(y1, ops.conv_fwd(xg, wg, bg, dil, **kw).cpu().numpy()), "forward must repeat bitwise"
    dx1 = ops.conv_dgrad(dyg, wg, dil, gate=xg, gate_scale=2.0 if drop else 1.0).cpu().numpy()
    assert _rel(dx1, want_dx) < 1e-4
    assert np.array_equal(dx1, ops.conv_dgrad(dyg, wg, dil, gate=xg, gate_scale=2.0 if drop else 1.0).cpu().numpy())
    dw1 = ops.conv_wgrad(xg, dyg, k, k, dil).cpu().numpy()
    assert _rel(dw1, wt.grad.numpy()) < 1e-4
    assert np.array_equal(dw1, ops.conv_wgrad(xg, dyg, k, k, dil).cpu().numpy()), "filter gradient must repeat bitwise"
    assert _rel(ops.bias_grad(dyg).cpu(), dy.astype(np.float64).sum((0, 1, 2))) < 1e-5


@pytest.mark.parametrize("shape,cus", [((2, 29, 9, 256, 256, 3, 8), 3), ((2, 29, 9, 256, 256, 5, 12), 3),
                                        ((2, 29, 16, 256, 256, 4, 8), 3), ((3, 37, 12, 272, 288, 4, 6), 5),
                                        ((4, 45, 9, 256, 288, 3, 12), 7)])
def test_stream_k_over_row_bands_on_small_shapes(env, shape, cus):
    """Dilated filters on short maps: the automatic plan is a stream-K launch over packed rows (the GEMM's rows are the pixels
    class after class; a tile reduces over the union of the tap boxes of the classes it holds rows of).  Forward with bias + ReLU +
    dropout mask and data gradient with the gate against the fp64 oracle, bitwise repeatable, and equal to the dense plan
    (fp32_loader = 2) to rounding."""
    ops, do, torch = env
    B, H, W, Cin, Cout, k, dil = shape
    rng = np.random.default_rng(sum(shape) + cus)
    x = rng.normal(0, 1, (B, H, W, Cin)).astype(np.float32)
    w = rng.normal(0, 0.2, (k, k, Cin, Cout)).astype(np.float32)
    b = rng.normal(0, 0.5, (Cout,)).astype(np.float32)
    dy = rng.normal(0, 1, (B, H, W, Cout)).astype(np.float32)
    gate = rng.normal(0, 1, (B, H, W, Cin)).astype(np.float32)
    mask = (rng.random((B, H, W, Cout)) < 0.5).astype(np.uint8)
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    y = do.conv_same(xt, torch.tensor(w, dtype=torch.float64), torch.tensor(b, dtype=torch.float64), dil)
    y.backward(torch.tensor(dy, dtype=torch.float64))
    want_y = torch.relu(y).detach().numpy() / 0.5 * mask
    want_dx = np.where(gate > 0, xt.grad.numpy() * 2.0, 0.0)
    xg, wg, bg, dyg, gg, mg = (_t(torch, x), _t(torch, w), _t(torch, b), _t(torch, dy), _t(torch, gate), _t(torch, mask, torch.uint8))
    with ops.tuned(plan_cus=cus):
        pf = ops.conv_plan(B, H, W, Cin, Cout, k, dil)
        assert pf[0] == 5 and pf[3] >= 2, pf          # pixel classes in use
        got_y = ops.conv_fwd(xg, wg, bg, dil, relu=True, keep_prob=0.5, drop_mask=mg).cpu().numpy()
        again = ops.conv_fwd(xg, wg, bg, dil, relu=True, keep_prob=0.5, drop_mask=mg).cpu().numpy()
        pd = ops.conv_plan(B, H, W, Cin, Cout, k, dil, dgrad=True)
        assert pd[0] == 5 and pd[3] >= 2, pd
        got_dx = ops.conv_dgrad(dyg, wg, dil, gate=gg, gate_scale=2.0).cpu().numpy()
        with ops.tuned(fp32_loader=2):
            assert ops.conv_plan(B, H, W, Cin, Cout, k, dil)[3] == 0
            dense_y = ops.conv_fwd(xg, wg, bg, dil, relu=True, keep_prob=0.5, drop_mask=mg).cpu().numpy()
    assert _rel(got_y, want_y) < 2e-5 and np.array_equal(got_y, again)
    assert _rel(got_dx, want_dx) < 2e-5, pd
    assert _rel(got_y, dense_y) < 2e-6


@pytest.mark.parametrize("batch", [6, 1])
def test_fc6_packed_rows_at_the_reference_batch(env, batch):
    """fc6 (deeplab.py:92,145) at the reference's own batch of 6 (deeplab.py:288): 25 pixel classes of 6 x (2..29) x (2..29)
    pixels, most of them a fraction of a tile -- the rows are packed across class boundaries and tiles reduce over union tap
    boxes.  Forward (bias + ReLU + built-in dropout) and gated data gradient against the dense walk of the same kernels
    (fp32_loader = 2), which the batch-16 test holds to the CPU oracle; batch 1 (too few K steps per CU for stream-K) stays
    on the plain plan and must agree with itself."""
    ops, do, torch = env
    B, H, W, Cin, Cout, k, dil = batch, 41, 41, 512, 4096, 4, 4
    pf, pd = ops.conv_plan(B, H, W, Cin, Cout, k, dil), ops.conv_plan(B, H, W, Cin, Cout, k, dil, dgrad=True)
    assert (pf[3], pd[3]) == ((25, 25) if batch == 6 else (0, 0)), (pf, pd)
    g = torch.Generator(device="cuda:0").manual_seed(batch)
    x = torch.relu(torch.randn((B, H, W, Cin), device="cuda:0", generator=g))
    w = torch.randn((k, k, Cin, Cout), device="cuda:0", generator=g) * 0.02
    b = torch.randn((Cout,), device="cuda:0", generator=g) * 0.1
    dy = torch.randn((B, H, W, Cout), device="cuda:0", generator=g)
    y = ops.conv_fwd(x, w, b, dil, relu=True, keep_prob=0.5, drop_seed=5)
    dx = ops.conv_dgrad(dy, w, dil, gate=x, gate_scale=2.0)
    assert torch.equal(y, ops.conv_fwd(x, w, b, dil, relu=True, keep_prob=0.5, drop_seed=5))
    with ops.tuned(fp32_loader=2):
        p0 = ops.conv_plan(B, H, W, Cin, Cout, k, dil)
        assert not (p0[0] == 5 and p0[3] > 0), p0     # dense: no pixel classes (plan[3] of another variant = tail slab levels)
        y0 = ops.conv_fwd(x, w, b, dil, relu=True, keep_prob=0.5, drop_seed=5)
        dx0 = ops.conv_dgrad(dy, w, dil, gate=x, gate_scale=2.0)
    # the same elements are dropped whatever the row order: where one result is zero and the other is not, it is a
    # pre-activation at the ReLU threshold (the two summation orders round it to either side), never a kept / dropped mismatch
    flip = (y != 0) != (y0 != 0)
    assert float(flip.float().mean()) < 1e-4 and (not bool(flip.any()) or float(torch.maximum(y.abs(), y0.abs())[flip].max()) < 1e-4)
    # two fp32 summation orders of K = 8192 (forward) / 65536 (data gradient) products: 1e-5 of the tensor's maximum
    assert float((y - y0).abs().max() / y0.abs().max()) < 1e-5
    assert float((dx - dx0).abs().max() / dx0.abs().max()) < 1e-5


@pytest.mark.parametrize("shape", [(2, 29, 9, 256, 256, 3, 8), (3, 21, 13, 512, 264, 4, 4), (2, 33, 12, 256, 288, 3, 12)])
def test_filter_gradient_skips_padding_without_changing_a_bit(env, shape):
    """Cin % 256 == 0: a k tile of the 256x256 filter-gradient kernel is one tap, and its reduction walks only the pixels at
    which that tap reads inside the image.  The products it leaves out are exact zeros and the order of the others is kept,
    so with an unsplit reduction (plan laid out for 1 CU: no pixel splits) the result equals the dense walk bit for bit;
    with pixel splits (the split points move) it agrees to rounding.  Both against the fp64 oracle."""
    ops, do, torch = env
    B, H, W, Cin, Cout, k, dil = shape
    rng = np.random.default_rng(sum(shape))
    x = rng.normal(0, 1, (B, H, W, Cin)).astype(np.float32)
    dy = rng.normal(0, 1, (B, H, W, Cout)).astype(np.float32)
    wt = torch.zeros((k, k, Cin, Cout), dtype=torch.float64, requires_grad=True)
    do.conv_same(torch.tensor(x, dtype=torch.float64), wt, torch.zeros(Cout, dtype=torch.float64), dil).backward(
        torch.tensor(dy, dtype=torch.float64))
    xg, dyg = _t(torch, x), _t(torch, dy)
    for cus, exact in ((1, True), (0, False)):
        with ops.tuned(wgrad_gen=2, plan_cus=cus):
            skip = ops.conv_wgrad(xg, dyg, k, k, dil)
            with ops.tuned(fp32_loader=2):
                dense = ops.conv_wgrad(xg, dyg, k, k, dil)
        assert _rel(skip.cpu(), wt.grad) < 2e-5 and _rel(dense.cpu(), wt.grad) < 2e-5
        if exact:
            assert torch.equal(skip, dense), "skipping zero products must not change the sum"
        else:
            assert _rel(skip.cpu(), dense.cpu()) < 2e-6


def test_stream_k_tail_launch_on_a_small_shape(env):
    """The automatic plan of the N > 1024 layers (fc6 / fc7 at batch 16): rows of whole rounds on the 256x256 kernel + the
    remaining rows as a stream-K launch of the same kernel starting at m_begin > 0, here with the plan laid out for 4 CUs
    so that a small shape takes it (30 tiles = 7 rounds + 2 tiles; the tail's 5 tiles x 36 K steps over 4 workgroups)."""
    ops, do, torch = env
    B, H, W, Cin, Cout, k, dil = 2, 23, 29, 64, 1040, 3, 1
    rng = np.random.default_rng(8)
    x = rng.normal(0, 1, (B, H, W, Cin)).astype(np.float32)
    w = rng.normal(0, 0.1, (k, k, Cin, Cout)).astype(np.float32)
    b = rng.normal(0, 0.5, (Cout,)).astype(np.float32)
    mask = (rng.random((B, H, W, Cout)) < 0.5).astype(np.uint8)
    y = do.conv_same(torch.tensor(x, dtype=torch.float64), torch.tensor(w, dtype=torch.float64),
                     torch.tensor(b, dtype=torch.float64), dil)
    want = torch.relu(y).numpy() / 0.5 * mask
    with ops.tuned(plan_cus=4, pack_pct=90):   # (a 3x3 filter on so small a map would otherwise take the packed-rows plan)
        plan = ops.conv_plan(B, H, W, Cin, Cout, k, dil)
        assert plan[0] == 4 and plan[1] == 1280 and plan[3] >= 1, plan
        got = ops.conv_fwd(_t(torch, x), _t(torch, w), _t(torch, b), dil, relu=True, keep_prob=0.5,
                           drop_mask=_t(torch, mask, torch.uint8)).cpu().numpy()
        again = ops.conv_fwd(_t(torch, x), _t(torch, w), _t(torch, b), dil, relu=True, keep_prob=0.5,
                             drop_mask=_t(torch, mask, torch.uint8)).cpu().numpy()
    assert _rel(got, want) < 2e-5
    assert np.array_equal(got, again)


@pytest.mark.parametrize("layer,shape", [("conv4_2", (16, 41, 41, 512, 512, 3, 1)), ("conv5_2", (16, 41, 41, 512, 512, 3, 2)),
                                         ("conv3_2", (16, 81, 81, 256, 256, 3, 1))])
def test_stream_k_combine_inside_the_launch_sees_fresh_partial_tiles(env, layer, shape):
    """Tiles shared by two or three workgroups are combined INSIDE the stream-K launch (csrc/igemm_v2.h: write-through slab per
    workgroup, arrival word, sc1 loads by the workgroup that runs the tile's head).  The slabs sit at the same workspace addresses
    launch after launch, so a combine that read a STALE copy (an L1 / L2 line left by an earlier launch) would go unnoticed on
    repeated inputs: here every launch gets new operands, the workspace is read by another kernel in between (plain loads leave
    its lines in the caches), and each result is held against the split-free plan of the same kernel (variant 4, one workgroup per
    tile: other sums, 2e-5) and against itself launched again (bitwise).  Benchmark shapes: 212 / 212 / 411 tiles on 256 CUs,
    every tile shared."""
    ops, do, torch = env
    B, H, W, Cin, Cout, k, dil = shape
    plan = ops.conv_plan(B, H, W, Cin, Cout, k, dil), ops.conv_plan(B, H, W, Cin, Cout, k, dil, dgrad=True)
    assert plan[0][0] == 5 and plan[0][2] >= 2 and plan[1][0] == 5 and plan[1][2] >= 2, plan
    g = torch.Generator(device="cuda").manual_seed(11)
    w = torch.randn((k, k, Cin, Cout), device="cuda", generator=g) * float(np.sqrt(2.0 / (k * k * Cin)))
    b = torch.randn((Cout,), device="cuda", generator=g) * 0.1
    for it in range(6):
        x = torch.randn((B, H, W, Cin), device="cuda", generator=g).relu_() * (1.0 + it)   # (new values, new magnitudes)
        dy = torch.randn((B, H, W, Cout), device="cuda", generator=g) * (1.0 + it)
        y = ops.conv_fwd(x, w, b, dil, relu=True)
        dx = ops.conv_dgrad(dy, w, dil, gate=x)
        ws = ops._WS.get(1, x.device)
        float(ws[: 64 << 20].float().sum())     # another kernel reads the slabs' lines
        assert torch.equal(y, ops.conv_fwd(x, w, b, dil, relu=True)), (layer, it)
        assert torch.equal(dx, ops.conv_dgrad(dy, w, dil, gate=x)), (layer, it)
        with ops.tuned(fp32_tile=4, splitk=1):
            y4 = ops.conv_fwd(x, w, b, dil, relu=True)
            dx4 = ops.conv_dgrad(dy, w, dil, gate=x)
        assert float((y - y4).abs().max()) <= 2e-5 * float(y4.abs().max()), (layer, it)
        assert float((dx - dx4).abs().max()) <= 2e-5 * float(dx4.abs().max()), (layer, it)
        assert bool(torch.isfinite(y).all()) and bool(torch.isfinite(dx).all())


def test_stream_k_combine_with_the_slabs_left_by_another_shape(env):
    """Stale-line check of the in-launch combine ("memory-model contract"): launches of DIFFERENT shapes alternate, so every
    slab and arrival word a launch reads was last written by a launch with another tile count, another piece list and other values
    (conv4_2: 212 tiles x 288 K steps, conv3_2: 411 x 144, fc6's data gradient over packed rows: 3392 units per CU) -- a head that
    read a line cached from the previous launch, or that saw a flag of the previous launch before this launch's memset node, would
    combine a foreign partial tile.  Each result against the one-workgroup-per-tile plan (2e-5) and bitwise against its repeat."""
    ops, do, torch = env
    shapes = [(16, 41, 41, 512, 512, 3, 1), (16, 81, 81, 256, 256, 3, 1), (16, 41, 41, 512, 1024, 4, 4)]
    g = torch.Generator(device="cuda").manual_seed(31)
    ws, xs = [], []
    for B, H, W, Cin, Cout, k, dil in shapes:
        ws.append((torch.randn((k, k, Cin, Cout), device="cuda", generator=g) * float(np.sqrt(2.0 / (k * k * Cin))),
                   torch.randn((Cout,), device="cuda", generator=g) * 0.1))
    assert ops.status_poll(clear=True, raise_on_error=False) == 0
    for it in range(3):
        outs = []
        for (B, H, W, Cin, Cout, k, dil), (w, b) in zip(shapes, ws):
            assert ops.conv_plan(B, H, W, Cin, Cout, k, dil)[0] == 5 and ops.conv_plan(B, H, W, Cin, Cout, k, dil, dgrad=True)[0] == 5
            x = torch.randn((B, H, W, Cin), device="cuda", generator=g).relu_() * (1.0 + it)
            dy = torch.randn((B, H, W, Cout), device="cuda", generator=g) * (2.0 - 0.5 * it)
            outs.append((x, dy, ops.conv_fwd(x, w, b, dil, relu=True), ops.conv_dgrad(dy, w, dil, gate=x)))   # back to back, no sync
        for (B, H, W, Cin, Cout, k, dil), (w, b), (x, dy, y, dx) in zip(shapes, ws, outs):
            with ops.tuned(fp32_tile=4, splitk=1, fp32_loader=2):
                y4 = ops.conv_fwd(x, w, b, dil, relu=True)
                dx4 = ops.conv_dgrad(dy, w, dil, gate=x)
            assert float((y - y4).abs().max()) <= 2e-5 * float(y4.abs().max()), (it, Cout)
            assert float((dx - dx4).abs().max()) <= 2e-5 * float(dx4.abs().max()), (it, Cout)
            assert torch.equal(y, ops.conv_fwd(x, w, b, dil, relu=True)) and torch.equal(dx, ops.conv_dgrad(dy, w, dil, gate=x))
    assert ops.status_poll(clear=True, raise_on_error=False) == 0


@pytest.mark.parametrize("layer,shape", [("conv4_2", (16, 41, 41, 512, 512, 3, 1)), ("conv3_2", (16, 81, 81, 256, 256, 3, 1))])
def test_stream_k_combine_when_not_every_workgroup_is_resident(env, layer, shape):
    """A stream-K launch laid out for MORE workgroups than the chip holds (tune plan_cus = 384 on 256 CUs; the kernel takes a whole
    CU's LDS, so exactly one workgroup is resident per CU): workgroups 256 .. 383 are dispatched only as earlier ones retire, and
    the head of a shared tile really waits for a publisher that has not started yet (csrc/igemm_v2.h: publishers store their partial
    tile before they ever wait, heads wait only for the workgroups right behind them -- so the launch always drains).  Deterministic,
    one process (round 4 provoked the same condition with two processes racing for the GPU: advisor finding).  Results must be
    complete, bitwise repeatable, equal to the one-workgroup-per-tile plan to 2e-5, and the device status word must stay clean."""
    ops, do, torch = env
    B, H, W, Cin, Cout, k, dil = shape
    g = torch.Generator(device="cuda").manual_seed(21)
    w = torch.randn((k, k, Cin, Cout), device="cuda", generator=g) * float(np.sqrt(2.0 / (k * k * Cin)))
    b = torch.randn((Cout,), device="cuda", generator=g) * 0.1
    x = torch.randn((B, H, W, Cin), device="cuda", generator=g).relu_()
    dy = torch.randn((B, H, W, Cout), device="cuda", generator=g)
    assert ops.status_poll(clear=True, raise_on_error=False) == 0
    with ops.tuned(fp32_tile=4, splitk=1):
        y4 = ops.conv_fwd(x, w, b, dil, relu=True)
        dx4 = ops.conv_dgrad(dy, w, dil, gate=x)
    with ops.tuned(plan_cus=384):
        plan = ops.conv_plan(B, H, W, Cin, Cout, k, dil), ops.conv_plan(B, H, W, Cin, Cout, k, dil, dgrad=True)
        assert plan[0][0] == 5 and plan[0][2] >= 2 and plan[1][0] == 5 and plan[1][2] >= 2, plan
        y = ops.conv_fwd(x, w, b, dil, relu=True)
        dx = ops.conv_dgrad(dy, w, dil, gate=x)
        assert torch.equal(y, ops.conv_fwd(x, w, b, dil, relu=True)) and torch.equal(dx, ops.conv_dgrad(dy, w, dil, gate=x))
    assert bool(torch.isfinite(y).all()) and bool(torch.isfinite(dx).all())
    assert float((y - y4).abs().max()) <= 2e-5 * float(y4.abs().max()), layer
    assert float((dx - dx4).abs().max()) <= 2e-5 * float(dx4.abs().max()), layer
    assert ops.status_poll(clear=True, raise_on_error=False) == 0


@pytest.mark.timeout(120)
def test_stream_k_wait_for_a_publisher_is_bounded_and_its_failure_is_visible(env):
    """The head of a shared tile polls the arrival words of the workgroups behind it -- a bounded poll (csrc/igemm_v2.h): with the
    publishers' words held down (tune sk_drop_flag, tests only) the launch must END, and the failure must be VISIBLE three ways
    (round-4 verdict: the fused ReLU used to map the NaN poison to 0, and nothing told the host):
      * the shared tiles come out as NaNs THROUGH the fused bias + ReLU epilogue (tf.nn.relu propagates NaN, deeplab.py:71,100) and
        through the data gradient's ReLU gate;
      * the device status word carries EMADAPT_STATUS_STREAMK_TIMEOUT, emadapt_status_poll returns EMADAPT_ERR_DEVICE_STATUS;
      * while it is raised the optimiser kernels leave weights, momentum and accumulator untouched (no host sync needed).
    After the poll cleared the word the next launches are clean again.  fp32 kernel and the bf16x6 plane kernel."""
    ops, do, torch = env
    from em_adapt_tensorflow_amd import _lib
    B, H, W, Cin, Cout, k = 16, 41, 41, 512, 512, 3
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn((B, H, W, Cin), device="cuda", generator=g).relu_()
    dy = torch.randn((B, H, W, Cout), device="cuda", generator=g)
    w = torch.randn((k, k, Cin, Cout), device="cuda", generator=g) * 0.02
    b = torch.randn((Cout,), device="cuda", generator=g) * 0.1
    d = ops.conv_desc(B, H, W, Cin, Cout, k, k, 1)
    xp, wkp = ops.split_planes(x), ops.filter_tiles(w)
    assert ops.status_poll(clear=True, raise_on_error=False) == 0
    good = ops.conv_fwd(x, w, b, 1, relu=True)
    good_dx = ops.conv_dgrad(dy, w, 1, gate=x)
    good6 = ops.conv_fwd_x6p(xp, wkp, d, b, relu=True)
    for which in ("fwd", "dgrad", "x6p"):
        with ops.tuned(sk_drop_flag=1):
            if which == "fwd":
                bad = ops.conv_fwd(x, w, b, 1, relu=True)
            elif which == "dgrad":
                bad = ops.conv_dgrad(dy, w, 1, gate=x)
            else:
                bad = ops.conv_fwd_x6p(xp, wkp, d, b, relu=True)
            torch.cuda.synchronize()
        assert bool(torch.isnan(bad).any()), "%s: a tile whose publisher never arrived must not look valid behind the fused ReLU / gate" % which
        # the optimiser is a no-op on the device while the word is raised
        wt, vt, at = torch.ones(1024, device="cuda"), torch.zeros(1024, device="cuda"), torch.full((1024,), 0.5, device="cuda")
        ops.sgd_momentum(wt, vt, at, 0.1, 0.9)
        ops.sgd_accumulate_momentum(wt, vt, at.clone(), 512, 1e-3, 1.0, 0.1, 0.9)
        ops.sgd_momentum_payload(wt, vt, at, at.clone().bfloat16(), 0.1, 0.9)
        assert float(wt.min()) == 1.0 and float(vt.abs().max()) == 0.0 and float(at.min()) == 0.5, which
        assert ops.status_poll(clear=False, raise_on_error=False) == _lib.STATUS_STREAMK_TIMEOUT
        with pytest.raises(_lib.EmadaptDeviceError) as ei:
            ops.status_poll(clear=True)
        assert ei.value.flags == _lib.STATUS_STREAMK_TIMEOUT and ei.value.code == -6
        assert ops.status_poll(clear=True, raise_on_error=False) == 0
        ops.sgd_momentum(wt, vt, at, 0.1, 0.9)
        assert abs(float(wt.max()) - 0.95) < 1e-6 and float(at.abs().max()) == 0.0
    assert torch.equal(ops.conv_fwd(x, w, b, 1, relu=True), good) and torch.equal(ops.conv_fwd_x6p(xp, wkp, d, b, relu=True), good6)
    assert torch.equal(ops.conv_dgrad(dy, w, 1, gate=x), good_dx)
    assert ops.status_poll(clear=True, raise_on_error=False) == 0


def test_relu_and_gate_epilogues_propagate_nan(env):
    """tf.nn.relu(NaN) is NaN (deeplab.py:71,100): a NaN pre-activation (or gradient) must survive every fused epilogue -- the direct
    16-byte-store one, the LDS-staged one, the split-K second stage, the bf16 kernels' -- instead of reading as a clean zero."""
    ops, do, torch = env
    B, H, W, Cin, Cout, k = 2, 23, 29, 64, 288, 3
    g = torch.Generator(device="cuda").manual_seed(8)
    x = torch.randn((B, H, W, Cin), device="cuda", generator=g).relu_()
    w = torch.randn((k, k, Cin, Cout), device="cuda", generator=g) * 0.05
    b = torch.randn((Cout,), device="cuda", generator=g) * 0.1
    dy = torch.randn((B, H, W, Cout), device="cuda", generator=g)
    x[1, 11, 13, 7] = float("nan")
    dy[0, 5, 6, 17] = float("nan")
    for variant, cus, splitk in [(1, 0, 0), (3, 0, 0), (3, 2, 3), (4, 0, 0), (4, 3, 2), (5, 0, 0), (5, 7, 0)]:
        with ops.tuned(fp32_tile=variant, plan_cus=cus, splitk=splitk):
            y = ops.conv_fwd(x, w, b, 1, relu=True)
            dx = ops.conv_dgrad(dy, w, 1, gate=torch.ones_like(x))
        assert bool(torch.isnan(y[1, 11, 13]).all()) and bool(torch.isnan(y[1, 10:13, 12:15]).all()), variant
        assert int(torch.isnan(y).sum()) == 9 * Cout, variant
        assert bool(torch.isnan(dx[0, 5, 6]).all()) and int(torch.isnan(dx).sum()) == 9 * Cin, variant
        # (and where the gate is closed a NaN gradient still shows: the ReLU-gradient select lets it through)
        with ops.tuned(fp32_tile=variant, plan_cus=cus, splitk=splitk):
            dx0 = ops.conv_dgrad(dy, w, 1, gate=torch.zeros_like(x).nan_to_num_())
        assert int(torch.isnan(dx0).sum()) == 9 * Cin and float(dx0.nan_to_num(0.0).abs().max()) == 0.0, variant
    xb, dyb = x.bfloat16(), dy.bfloat16()
    wf, wd = ops.weights_to_bf16(w)
    yb = ops.conv_fwd_bf16(xb, wf, (k, k), b, 1, relu=True)
    dxb = ops.conv_dgrad_bf16(dyb, wd, (k, k), 1, gate=torch.ones_like(xb))
    assert int(torch.isnan(yb).sum()) == 9 * Cout and int(torch.isnan(dxb).sum()) == 9 * Cin


def test_builtin_dropout_draws_do_not_depend_on_the_launch_plan(env):
    """The counter-hash dropout generator is keyed by the GLOBAL output element, so every tile variant -- row splits,
    split-K second stages, stream-K pieces and their fix-up pass -- must drop exactly the same elements for a seed."""
    ops, do, torch = env
    B, H, W, Cin, Cout, k, dil = 2, 23, 29, 64, 288, 3, 1
    rng = np.random.default_rng(3)
    xg = _t(torch, np.abs(rng.normal(0, 1, (B, H, W, Cin))).astype(np.float32))
    wg = _t(torch, np.abs(rng.normal(0, 0.1, (k, k, Cin, Cout))).astype(np.float32))
    bg = _t(torch, np.full((Cout,), 0.5, np.float32))     # all pre-activations positive: zero <=> dropped
    ref = None
    for variant, cus, splitk in [(3, 0, 0), (1, 2, 2), (2, 3, 0), (3, 2, 3), (4, 0, 0), (4, 3, 2), (5, 0, 0), (5, 3, 0), (5, 7, 0)]:
        with ops.tuned(fp32_tile=variant, plan_cus=cus, splitk=splitk):
            y = ops.conv_fwd(xg, wg, bg, dil, relu=True, keep_prob=0.5, drop_seed=(9 << 32) + 1234)
        kept = y != 0
        if ref is None:
            ref = kept
            assert abs(float(kept.float().mean()) - 0.5) < 5e-3
        assert torch.equal(kept, ref), (variant, cus, splitk)


def test_builtin_dropout_on_the_big_tile_path(env):
    """fc7 at batch 16 with the counter-hash generator: the element index fed to the hash must be the global one in
    both launches of the plan (rows below and above main_rows): kept values equal relu(y)/keep exactly, the keep rate
    is right in both row ranges and the same seed repeats."""
    ops, do, torch = env
    B, H, W, C = 16, 41, 41, 4096
    plan = ops.conv_plan(B, H, W, C, C, 1, 1)
    assert plan[0] == 4 and plan[3] >= 1
    rng = np.random.default_rng(1)
    xg = _t(torch, np.abs(rng.normal(0, 1, (B, H, W, C))).astype(np.float32))
    wg = _t(torch, np.abs(rng.normal(0, 0.01, (1, 1, C, C))).astype(np.float32))
    bg = _t(torch, np.full((C,), 0.1, np.float32))
    base = ops.conv_fwd(xg, wg, bg, 1, relu=True)
    d1 = ops.conv_fwd(xg, wg, bg, 1, relu=True, keep_prob=0.5, drop_seed=77)
    d2 = ops.conv_fwd(xg, wg, bg, 1, relu=True, keep_prob=0.5, drop_seed=77)
    assert torch.equal(d1, d2)
    kept = d1 != 0
    assert torch.equal(d1[kept], (base / 0.5)[kept])
    rows = kept.reshape(-1, C)
    main = plan[1]
    assert abs(float(rows[:main].float().mean()) - 0.5) < 2e-3 and abs(float(rows[main:].float().mean()) - 0.5) < 2e-3
    # the two row ranges draw from one stream: no duplicated pattern between the first main and the first tail row
    assert not torch.equal(rows[0], rows[main])


# (B, H, W, Cin, Cout, k, dil): M tails (M % 256, M % 128 != 0), N tails, K % 32 != 0, N below / above the tile width
SMALL = [
    (2, 23, 29, 40, 72, 3, 1),
    (1, 37, 35, 64, 264, 3, 2),
    (3, 17, 19, 24, 520, 1, 1),
    (2, 21, 21, 32, 40, 4, 4),
    (1, 45, 43, 160, 24, 1, 1),
    (2, 23, 29, 64, 288, 3, 1),     # both directions eligible for the second-generation tiles (channels % 32 == 0)
    (1, 37, 35, 272, 48, 3, 2),     # channels % 16 == 0 but not % 32
    (3, 17, 19, 96, 544, 1, 1),     # 1x1 (one tap), N tail of the 256-wide tile
    (2, 21, 21, 32, 160, 4, 4),     # 16 taps, dilation 4
    (1, 19, 17, 256, 272, 3, 2),    # Cin % 256 == 0: the filter-gradient loaders' uniform-tap form, with out-of-image taps
    (2, 11, 13, 512, 264, 1, 1),    # the same for a 1x1 filter (two k tiles inside the one tap)
]


def _eligible(variant, Cx, k):
    """csrc/conv_igemm.hip v2_ok: what a forced second-generation variant needs (otherwise the automatic plan is used)."""
    bk = {4: 16, 5: 16}.get(variant)
    return bk is None or (Cx % bk == 0 and k * k <= 32 and k * k * Cx >= 2 * bk)


@pytest.mark.parametrize("shape", SMALL)
def test_every_plan_variant_on_small_awkward_shapes(env, shape):
    ops, do, torch = env
    B, H, W, Cin, Cout, k, dil = shape
    rng = np.random.default_rng(sum(shape))
    x = rng.normal(0, 1, (B, H, W, Cin)).astype(np.float32)
    w = rng.normal(0, 0.2, (k, k, Cin, Cout)).astype(np.float32)
    b = rng.normal(0, 0.5, (Cout,)).astype(np.float32)
    dy = rng.normal(0, 1, (B, H, W, Cout)).astype(np.float32)
    gate = rng.normal(0, 1, (B, H, W, Cin)).astype(np.float32)
    mask = (rng.random((B, H, W, Cout)) < 0.5).astype(np.uint8)
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    wt = torch.tensor(w, dtype=torch.float64, requires_grad=True)
    y = do.conv_same(xt, wt, torch.tensor(b, dtype=torch.float64), dil)
    y.backward(torch.tensor(dy, dtype=torch.float64))
    want_y = torch.relu(y).detach().numpy() / 0.5 * mask
    want_dx = np.where(gate > 0, xt.grad.numpy() * 2.0, 0.0)
    xg, wg, bg, dyg, gg, mg = (_t(torch, x), _t(torch, w), _t(torch, b), _t(torch, dy), _t(torch, gate),
                               _t(torch, mask, torch.uint8))
    M = B * H * W
    seen = set()
    # variant x CU count the plan is laid out for x forced split-K: reaches row splits at m_begin > 0, partial last
    # tiles in both launches, and the tail launch's slab at workspace offset 0
    for variant, cus, splitk, *rest in [(1, 0, 0), (1, 2, 2), (3, 0, 0), (3, 2, 3), (2, 0, 0), (2, 2, 0), (2, 3, 0), (2, 4, 2),
                                 (2, 5, 3), (4, 0, 0), (4, 2, 0), (4, 3, 2), (4, 5, 3),
                                 # stream-K: more workgroups than K steps per tile (pieces of 1-2 steps, many slab levels),
                                 # and few workgroups (whole tiles between a tail piece and a head piece); loader = 1: the
                                 # generic (any tap count) loaders of variants 4 / 5 and of the filter gradient
                                 (5, 0, 0), (5, 3, 0), (5, 7, 0), (5, 16, 0), (5, 5, 0, 1), (4, 3, 2, 1)]:
        loader = rest[0] if rest else 0
        with ops.tuned(fp32_tile=variant, plan_cus=cus, splitk=splitk, fp32_loader=loader):
            for dgrad in (False, True):
                plan = ops.conv_plan(B, H, W, Cin, Cout, k, dil, dgrad=dgrad)
                if _eligible(variant, Cout if dgrad else Cin, k):
                    assert plan[0] == variant, (plan, variant, dgrad)
                seen.add((plan[0], plan[1] < M, plan[2] > 1, plan[3] > 1))
            got_y = ops.conv_fwd(xg, wg, bg, dil, relu=True, keep_prob=0.5, drop_mask=mg).cpu().numpy()
            got_dx = ops.conv_dgrad(dyg, wg, dil, gate=gg, gate_scale=2.0).cpu().numpy()
        assert _rel(got_y, want_y) < 2e-5, (variant, cus, splitk, loader)
        assert _rel(got_dx, want_dx) < 2e-5, (variant, cus, splitk, loader)
    # both generations of the filter-gradient kernel (128x128 wave-specialised; 256x256 LDS-DMA), with the CU count the
    # split-M plan is laid out for varied: single pass, and several pixel splits with a ragged last one
    for gen, cus, loader in [(1, 0, 0), (1, 1, 0), (2, 0, 0), (2, 1, 0), (2, 3, 0), (2, 3, 1)]:
        with ops.tuned(wgrad_gen=gen, plan_cus=cus, fp32_loader=loader):
            got_dw = ops.conv_wgrad(xg, dyg, k, k, dil).cpu().numpy()
            again = ops.conv_wgrad(xg, dyg, k, k, dil).cpu().numpy()
        assert _rel(got_dw, wt.grad.numpy()) < 2e-5, (gen, cus, loader)
        assert np.array_equal(got_dw, again), "filter gradient must repeat bitwise"
    assert ops.tune_get("fp32_tile") == 0 and ops.tune_get("plan_cus") == 0 and ops.tune_get("wgrad_gen") == 0
    if M > 1024 and max(Cin, Cout) >= 40:
        assert any(v == 2 and rows_split for v, rows_split, _, _ in seen), "no forced plan split the rows: %s" % seen
    if M > 1024 and Cin % 16 == 0 and Cout % 16 == 0 and k * k <= 32:
        assert any(v == 4 and rows_split for v, rows_split, _, _ in seen), "variant 4 never split the rows: %s" % seen


def test_one_wide_filters_and_one_pixel_wide_maps(env):
    """KW == 1 with KH > 1 (and the reverse), W == 1 / H == 1 inputs: the magic-number divisions by 1 (common.h:
    div_magic) in the tap decomposition of the forward / data-gradient kernels and the pixel decomposition of wgrad."""
    ops, do, torch = env
    rng = np.random.default_rng(8)
    for (B, H, W, Cin, Cout, kh, kw, dil) in [(2, 9, 11, 16, 72, 3, 1, 1), (2, 9, 11, 16, 72, 1, 3, 2),
                                              (2, 1, 37, 16, 72, 3, 3, 1), (2, 37, 1, 16, 72, 3, 3, 2),
                                              (1, 1, 1, 8, 8, 3, 3, 1), (2, 9, 11, 32, 48, 3, 1, 2), (2, 9, 11, 32, 48, 1, 4, 1)]:
        x = rng.normal(0, 1, (B, H, W, Cin)).astype(np.float32)
        w = rng.normal(0, 0.2, (kh, kw, Cin, Cout)).astype(np.float32)
        dy = rng.normal(0, 1, (B, H, W, Cout)).astype(np.float32)
        xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
        wt = torch.tensor(w, dtype=torch.float64, requires_grad=True)
        y = do.conv_same(xt, wt, None, dil)
        y.backward(torch.tensor(dy, dtype=torch.float64))
        xg, wg, dyg = _t(torch, x), _t(torch, w), _t(torch, dy)
        case = (B, H, W, Cin, Cout, kh, kw, dil)
        for variant in (0, 4):     # automatic plan, and the second-generation kernel's tap masks on non-square filters
            with ops.tuned(fp32_tile=variant):
                assert _rel(ops.conv_fwd(xg, wg, None, dil).cpu(), y.detach()) < 2e-5, (case, variant)
                assert _rel(ops.conv_dgrad(dyg, wg, dil).cpu(), xt.grad) < 2e-5, (case, variant)
        for gen in (0, 2):
            with ops.tuned(wgrad_gen=gen):
                assert _rel(ops.conv_wgrad(xg, dyg, kh, kw, dil).cpu(), wt.grad) < 2e-5, (case, gen)


def test_tensors_of_2_gib_and_more_run_as_batch_slices(env):
    """513x513 inputs at batch 32 put conv1's activations (2.16 GB) past the 32-bit byte offsets the kernels address an operand
    with (config 5 "as memory allows" was capped at batch 31 by offsets, not by the 288 GB).  The fp32 entries now cut
    such a launch into slices of whole images (csrc/igemm_common.h slice_batch; the filter gradient adds the slices' results in
    slice order).  A synthetic layer whose OUTPUT is 2.16 GB (129 x 256 x 256 x 64 fp32: slices of 127 + 2 images): forward with
    bias + ReLU, gated data gradient and filter gradient against the fp32 CPU oracle over every image."""
    ops, do, torch = env
    B, H, W, Cin, Cout, k = 129, 256, 256, 32, 64, 3
    assert B * H * W * Cout * 4 >= 2 ** 31 > (B - 2) * H * W * Cout * 4
    g = torch.Generator(device="cuda:0").manual_seed(5)
    x = torch.randn((B, H, W, Cin), device="cuda:0", generator=g).clamp_(min=0)
    w = torch.randn((k, k, Cin, Cout), device="cuda:0", generator=g) * float(np.sqrt(2.0 / (k * k * Cin)))
    b = torch.randn((Cout,), device="cuda:0", generator=g) * 0.1
    dy = torch.randn((B, H, W, Cout), device="cuda:0", generator=g)
    y = ops.conv_fwd(x, w, b, 1, relu=True)
    dx = ops.conv_dgrad(dy, w, 1, gate=x)
    dw = ops.conv_wgrad(x, dy, k, k, 1)
    assert torch.equal(dw, ops.conv_wgrad(x, dy, k, k, 1)), "sliced filter gradient must repeat bitwise"
    # oracle, image by image (the CPU tensors of the whole batch would be 2 x 2 GB of autograd state)
    wt = w.cpu().requires_grad_(True)
    bt = b.cpu()
    worst_y = worst_dx = 0.0
    for b0 in range(0, B, 16):
        xs = x[b0:b0 + 16].cpu().requires_grad_(True)
        ys = do.conv_same(xs, wt, bt, 1)
        ys.backward(dy[b0:b0 + 16].cpu())
        worst_y = max(worst_y, _rel(y[b0:b0 + 16].cpu().numpy(), torch.relu(ys).detach().numpy()))
        want_dx = np.where(xs.detach().numpy() > 0, xs.grad.numpy(), 0)
        worst_dx = max(worst_dx, _rel(dx[b0:b0 + 16].cpu().numpy(), want_dx))
    assert worst_y < 1e-4 and worst_dx < 1e-4, (worst_y, worst_dx)
    assert _rel(dw.cpu().numpy(), wt.grad.numpy()) < 1e-4
